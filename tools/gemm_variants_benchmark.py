"""Large-batch 1x16 op: the K-split LDS-DMA pipeline (gemm_variant 3) against the 16-row no-split kernel (2) and a dense fp16
GEMM, over layer shapes, group sizes and batch rows (hipGraph replay over 24 distinct layers, HIP events).  The selection rule of
aqlm_hip_gemm_1x16_mfma (gemm_mfma.hip: R16_ROWS_ANY / R16_ROWS_SMALL / R16_SMALL_LAYER) was read off this table.

    python tools/gemm_variants_benchmark.py [--out profiles/r04_gemm_rows16_shapes.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import aqlm_amd.inference_kernels.hip_kernel as hk  # noqa: E402
from aqlm_amd import _native  # noqa: E402

dev = torch.device("cuda:0")


def layers(fin, fout, g, n):
    gen = torch.Generator(device=dev).manual_seed(fin + fout)
    out = []
    for _ in range(n):
        codes = torch.randint(-32768, 32768, (fout, fin // g, 1), generator=gen, device=dev, dtype=torch.int32).to(torch.int16)
        out.append((codes, torch.randn((1, 65536, 1, g), generator=gen, device=dev).half()))
    return out


def timeit(fn, ls, reps=5):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for item in ls:
            fn(*item)
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for item in ls:
            fn(*item)
    with torch.cuda.stream(st):
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            g.replay()
        e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(ls))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    res = {}
    for g in (8, 16):
        for fin, fout in ((4096, 4096), (4096, 11008), (11008, 4096), (8192, 8192)):
            ls = layers(fin, fout, g, 24)
            scales = torch.ones((fout, 1, 1, 1), device=dev, dtype=torch.float16)
            W = torch.randn((fout, fin), device=dev).half()  # ONE dense weight, re-used: cache-warm, i.e. flattering for dense
            for B in (8, 16, 32, 48, 64, 96, 128):
                x = torch.randn((B, fin), device=dev).half()
                row = {}
                for v in (3, 2):
                    _native.set_tuning("gemm_variant", v)
                    row[v] = timeit(lambda c, cb: hk.code1x16_matmat_dequant(x, c, cb, scales, None), ls)
                _native.set_tuning("gemm_variant", 0)
                dense = timeit(lambda c, cb: torch.nn.functional.linear(x, W), ls)
                print(f"g{g} {fin}->{fout} B={B}: K-split pipeline {row[3]:.2f} us  16-row kernel {row[2]:.2f} us  dense (warm) {dense:.2f} us", flush=True)
                res[f"g{g}_{fin}x{fout}_B{B}"] = {"glds_us": row[3], "rows16_us": row[2], "dense_fp16_us": dense}
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
