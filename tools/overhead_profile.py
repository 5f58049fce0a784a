"""cProfile of the host side of one eager call of the prepacked op / the raw op / nn.Linear (where do the ~25 us go?)."""
import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqlm_amd.inference_kernels import hip_kernel as hk

dev = torch.device("cuda:0")
fin, fout = 4096, 11008
codes = torch.randint(-32768, 32767, (fout, fin // 8, 1), device=dev).to(torch.int16)
cb = torch.randn(1, 65536, 1, 8, device=dev, dtype=torch.float16)
sc = torch.ones(fout, 1, 1, 1, device=dev, dtype=torch.float16)
x = torch.randn(1, 1, fin, device=dev, dtype=torch.float16)
w = torch.randn(fout, fin, device=dev, dtype=torch.float16)
packed = hk.prepack_1x16(codes)

def t(fn, n=3000):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

print(f"F.linear dense                 {t(lambda: torch.nn.functional.linear(x, w)):6.1f} us/call")
print(f"code1x16_matmat_packed         {t(lambda: hk.code1x16_matmat_packed(x, packed, cb, sc, None)):6.1f} us/call")
print(f"code1x16_matmat (raw, cached)  {t(lambda: hk.code1x16_matmat(x, codes, cb, sc, None)):6.1f} us/call")
print(f"torch.ops.aqlm.code1x16_matmat {t(lambda: torch.ops.aqlm.code1x16_matmat(x, codes, cb, sc, None)):6.1f} us/call")
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    hk.code1x16_matmat(x, codes, cb, sc, None)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:4500])
