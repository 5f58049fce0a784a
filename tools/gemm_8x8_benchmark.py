"""8x8 g32 beyond one row: the fused dequant -> MFMA kernel (aqlm_hip_gemm_8x8_mfma, round 5) against the routes it replaces -- the
look-up-table matvec launched for 2..8 rows (one set of workgroups per row), dequantise + library GEMM above -- and a dense fp16
GEMM on rotating weights; hipGraph replay over 24 distinct layers, us per call, plus the parity of the two routes on one layer.

    python tools/gemm_8x8_benchmark.py > profiles/r05_gemm_8x8_mfma.log
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import aqlm_amd.inference_kernels.hip_kernel as hk
from tools.gemm_variants_benchmark import dev, timeit


def layers(fin, fout, n):
    gen = torch.Generator(device=dev).manual_seed(fin + fout)
    out = []
    for _ in range(n):
        codes = torch.randint(-128, 128, (fout, fin // 32, 8), generator=gen, device=dev, dtype=torch.int32).to(torch.int8)
        out.append((codes, (torch.randn((8, 256, 1, 32), generator=gen, device=dev) * 0.35).half()))
    return out


if len(sys.argv) > 1 and sys.argv[1] == "one":  # python tools/gemm_8x8_benchmark.py one <in> <out> <rows>: 200 calls (counter passes)
    fin, fout, B = (int(a) for a in sys.argv[2:5])
    ls = layers(fin, fout, 8)
    scales = torch.ones((fout, 1, 1, 1), device=dev, dtype=torch.float16)
    x = torch.randn((B, fin), device=dev).half()
    for i in range(200):
        hk._fused_8x8_mfma(x, ls[i % 8][0], ls[i % 8][1], scales, None, hk._dtype_id(x))
    torch.cuda.synchronize()
    sys.exit(0)

shapes = ((4096, 4096), (4096, 11008), (11008, 4096), (8192, 8192), (4096, 1024))
rows = (2, 3, 4, 6, 8, 16, 32, 64, 128)
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes, rows = shapes[:2], (2, 3, 4, 8, 16, 64)
for fin, fout in shapes:
    ls = layers(fin, fout, 24)
    scales = torch.ones((fout, 1, 1, 1), device=dev, dtype=torch.float16)
    Ws = [torch.randn((fout, fin), device=dev).half() for _ in range(24)]
    for B in rows:
        x = torch.randn((B, fin), device=dev).half()
        op = hk.codekx8_matmat if B <= 8 else hk.code2x8_matmat_dequant
        res = {}
        for fused in (True, False, True, False):
            if not fused and B > 32 and fin * fout > (1 << 26):
                continue
            hk.USE_FUSED_8X8_MFMA = fused
            hk.FUSED_8X8_MFMA_MIN_ROWS = 2
            t = timeit(lambda c, cb: op(x, c, cb, scales, None), ls)
            res[fused] = min(res.get(fused, 1e9), t)
        hk.USE_FUSED_8X8_MFMA = True
        ya = op(x, ls[0][0], ls[0][1], scales, None).float()
        hk.USE_FUSED_8X8_MFMA = False
        yb = op(x, ls[0][0], ls[0][1], scales, None).float()
        hk.USE_FUSED_8X8_MFMA = True
        rel = float((ya - yb).abs().mean() / yb.abs().mean())
        it = iter(range(10**9))
        t_d = timeit(lambda c, cb: torch.nn.functional.linear(x, Ws[next(it) % 24]), ls)
        other = "look-up-table rows" if B <= 8 else "dequant + GEMM"
        print(f"8x8g32 {fin}->{fout} B={B}: fused MFMA {res[True]:.2f} us  {other} {res.get(False, float('nan')):.2f} us  dense fp16 {t_d:.2f} us  "
              f"mean-rel(fused vs other) {rel:.2e}{'' if rel < 2e-3 else '   <-- MISMATCH'}", flush=True)
