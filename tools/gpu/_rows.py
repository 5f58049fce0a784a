import sys, os
sys.path.insert(0, os.getcwd())
import torch
import aqlm_amd.inference_kernels.hip_kernel as hk
from tools.gemm_variants_benchmark import timeit, dev
def layers(fin, fout, K, n):
    gen = torch.Generator(device=dev).manual_seed(fin + fout)
    return [(torch.randint(-128, 128, (fout, fin // 8, K), generator=gen, device=dev, dtype=torch.int32).to(torch.int8),
             torch.randn((K, 256, 1, 8), generator=gen, device=dev).half()) for _ in range(n)]
for fin, fout in ((4096, 4096), (4096, 11008), (11008, 4096)):
    ls = layers(fin, fout, 2, 48)
    scales = torch.ones((fout, 1, 1, 1), device=dev, dtype=torch.float16)
    for B in (1, 2, 3, 4, 6, 8):
        x = torch.randn((B, fin), device=dev).half()
        t_v = timeit(lambda c, cb: hk.code2x8_matmat(x, c, cb, scales, None), ls)
        t_f = timeit(lambda c, cb: hk.code2x8_matmat_dequant(x, c, cb, scales, None), ls)
        print(f"2x8g8 {fin}->{fout} rows={B}: matvec kernel {t_v:.2f} us  fused MFMA kernel {t_f:.2f} us", flush=True)
