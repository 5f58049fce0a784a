"""Host-side overhead of one QuantizedLinear call in eager mode (tiny layer: the kernel itself takes ~3 us)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aqlm
from aqlm_amd.inference_kernels import hip_kernel

dev = torch.device("cuda:0")
def mk(K, nbits, g, fin=1024, fout=1024):
    m = aqlm.QuantizedLinear(fin, fout, g, 1, K, nbits, bias=False, device=dev, dtype=torch.float16)
    with torch.no_grad():
        m.codes.copy_(torch.randint(-2 ** (nbits - 1), 2 ** (nbits - 1), m.codes.shape, device=dev).to(m.codes.dtype))
        m.codebooks.normal_(); m.scales.fill_(1.0)
    return m

def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

x = torch.randn(1, 1024, dtype=torch.float16, device=dev)
lin = torch.nn.Linear(1024, 1024, bias=False, device=dev, dtype=torch.float16)
with torch.no_grad():
    print(f"nn.Linear                        {t(lambda: lin(x)):7.1f} us/call")
    for name, (K, nb, g) in {"1x16g8": (1, 16, 8), "2x8g8": (2, 8, 8), "8x8g32": (8, 8, 32)}.items():
        m = mk(K, nb, g)
        m(x)
        print(f"QuantizedLinear {name:8s}         {t(lambda: m(x)):7.1f} us/call")
    m = mk(1, 16, 8)
    m(x)
    print(f"torch.ops.aqlm.code1x16_matmat   {t(lambda: torch.ops.aqlm.code1x16_matmat(x, m.codes, m.codebooks, m.scales, None)):7.1f} us/call")
    print(f"hip_kernel.code1x16_matmat       {t(lambda: hip_kernel.code1x16_matmat(x, m.codes, m.codebooks, m.scales, None)):7.1f} us/call")
    big = mk(1, 16, 8, 4096, 8192)   # 4.2 M codes -> prepacked path
    xb = torch.randn(1, 4096, dtype=torch.float16, device=dev)
    big(xb)
    print(f"QuantizedLinear 4096->8192 (prepacked) {t(lambda: big(xb), 500):7.1f} us/call")

# shared-input group: host cost of one fused launch vs three separate module calls (tiny layers)
import aqlm
with torch.no_grad():
    holder = torch.nn.Module()
    for n in ("q_proj", "k_proj", "v_proj"):
        setattr(holder, n, mk(1, 16, 8))
    def three():
        holder.q_proj(x); holder.k_proj(x); holder.v_proj(x)
    three()
    print(f"3 separate QuantizedLinear calls      {t(three, 1000):7.1f} us")
    aqlm.fuse_shared_input_linears(holder)
    three()
    print(f"same through a shared-input group     {t(three, 1000):7.1f} us")
