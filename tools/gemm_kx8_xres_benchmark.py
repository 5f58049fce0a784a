"""Fused K x 8 MFMA op at <= 16 rows: the X-resident kernel (round 5, tuning key kx8_xres = 1) against the streaming 16-row kernel
(kx8_xres = 0) and a dense fp16 GEMM on rotating weights; hipGraph replay over 24 distinct layers, us per call.  Also the
parity of the two kernels on one layer (same exact products, another summation order).

    python tools/gemm_kx8_xres_benchmark.py > profiles/r05_gemm_kx8_xres.log
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import aqlm_amd.inference_kernels.hip_kernel as hk
from aqlm_amd import _native
from tools.gemm_variants_benchmark import dev, timeit


def layers(fin, fout, K, n):
    gen = torch.Generator(device=dev).manual_seed(fin + fout + K)
    out = []
    for _ in range(n):
        codes = torch.randint(-128, 128, (fout, fin // 8, K), generator=gen, device=dev, dtype=torch.int32).to(torch.int8)
        out.append((codes, torch.randn((K, 256, 1, 8), generator=gen, device=dev).half()))
    return out


if len(sys.argv) > 1 and sys.argv[1] == "one":  # python tools/gemm_kx8_xres_benchmark.py one <K> <in> <out> <rows>: 200 calls (counter passes)
    K, fin, fout, B = (int(a) for a in sys.argv[2:6])
    ls = layers(fin, fout, K, 8)
    scales = torch.ones((fout, 1, 1, 1), device=dev, dtype=torch.float16)
    x = torch.randn((B, fin), device=dev).half()
    op = hk.code2x8_matmat_dequant if K == 2 else hk.code1x8_matmat_dequant
    for i in range(200):
        op(x, ls[i % 8][0], ls[i % 8][1], scales, None)
    torch.cuda.synchronize()
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "multi":  # shared-input launches (q / k / v, gate / up): one launch over all layers vs one per layer
    groups = {"llama2-7b q/k/v": (4096, (4096, 4096, 4096)), "llama2-7b gate/up": (4096, (11008, 11008)),
              "llama3-8b q/k/v": (4096, (4096, 1024, 1024)), "llama2-13b gate/up": (5120, (13824, 13824))}
    rows_list = (1, 2, 4, 8, 16)
    if len(sys.argv) > 2 and sys.argv[2] == "b1":  # where does the MFMA form win at ONE row?  (small groups)
        groups = {"gqa 4096 q/k/v": (4096, (4096, 1024, 1024)), "two 4096": (4096, (4096, 4096)), "13b q/k/v": (5120, (5120, 5120, 5120)),
                  "70b q/k/v": (8192, (8192, 1024, 1024)), "two 2048": (4096, (2048, 2048)), "small trio": (2048, (2048, 512, 512)),
                  "mistral gate/up": (4096, (14336, 14336)), "k/v only": (4096, (1024, 1024))}
        rows_list = (1,)
    for name, (fin, fouts) in groups.items():
        sets = [[layers(fin, fo, 2, 1)[0] for fo in fouts] for _ in range(16)]
        scs = [torch.ones((fo, 1, 1, 1), device=dev, dtype=torch.float16) for fo in fouts]
        for B in rows_list:
            x = torch.randn((B, fin), device=dev).half()
            res = {}
            for rep in range(2):
                for mr in (0, 2, 1):
                    _native.set_tuning("kx8_multi_xres_min_rows", mr)
                    t = timeit(lambda *ls: hk.codekx8_matmat_multi(x, [l[0] for l in ls], [l[1] for l in ls], scs, [None] * len(ls)),
                               [tuple(st) for st in sets])
                    res[mr] = min(res.get(mr, 1e9), t)
            _native.set_tuning("kx8_multi_xres_min_rows", 2)
            print(f"2x8g8 {name} {fin}->{fouts} B={B}: one launch per layer {res[0]:.2f} us  shared-input default {res[2]:.2f} us  X-resident kernel also at 1 row {res[1]:.2f} us", flush=True)
    sys.exit(0)

for K in (2, 1):
    for fin, fout in ((4096, 4096), (4096, 11008), (11008, 4096), (8192, 8192), (4096, 1024)):
        if K == 1 and (fin, fout) != (4096, 4096):
            continue
        ls = layers(fin, fout, K, 24)
        scales = torch.ones((fout, 1, 1, 1), device=dev, dtype=torch.float16)
        Ws = [torch.randn((fout, fin), device=dev).half() for _ in range(24)]
        op = hk.code2x8_matmat_dequant if K == 2 else hk.code1x8_matmat_dequant
        for B in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            x = torch.randn((B, fin), device=dev).half()
            res = {}
            for xres in (1, 0, 1, 0):
                _native.set_tuning("kx8_xres", xres)
                t = timeit(lambda c, cb: op(x, c, cb, scales, None), ls)
                res[xres] = min(res.get(xres, 1e9), t)
            _native.set_tuning("kx8_xres", 1)
            ya = op(x, ls[0][0], ls[0][1], scales, None).float()
            _native.set_tuning("kx8_xres", 0)
            yb = op(x, ls[0][0], ls[0][1], scales, None).float()
            _native.set_tuning("kx8_xres", 1)
            rel = float((ya - yb).abs().mean() / yb.abs().mean())
            it = iter(range(10**9))
            t_d = timeit(lambda c, cb: torch.nn.functional.linear(x, Ws[next(it) % 24]), ls)
            print(f"{K}x8g8 {fin}->{fout} B={B}: X-resident {res[1]:.2f} us  streaming {res[0]:.2f} us  dense fp16 {t_d:.2f} us  "
                  f"mean-rel(resident vs streaming) {rel:.2e}{'' if rel < 1e-3 else '   <-- MISMATCH'}", flush=True)
