#!/usr/bin/env python3
"""1x16 g8 at 1 .. 128 rows: the slice-scan MFMA kernel (round 6) next to the prepacked matvec, round 5's L2-gather MFMA kernels and a
dense fp16 GEMM.  hipGraph replay over > 600 MB of distinct layers (cold), HIP events on the capture stream.

    python tools/scan_benchmark.py [--shapes 4096x4096,4096x11008,...] [--rows 1,2,4,8,16,32,64,128] [--json out.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096,8192x8192,4096x14336,14336x4096,4096x1024")
    ap.add_argument("--rows", default="1,2,3,4,6,8,12,16,24,32,64,128")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-old", action="store_true", help="skip round 5's L2-gather kernels")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from aqlm_amd import _native
    from aqlm_amd.inference_kernels import hip_kernel as hk
    from benchlib.layers import GraphedCalls, GraphedPass, Layer, algorithmic_bytes

    dev = torch.device("cuda:0")
    lib = _native.lib
    out = {}
    for shp in args.shapes.split(","):
        fi, fo = (int(v) for v in shp.split("x"))
        n = max(8, min(64, int(600e6 / algorithmic_bytes(fi, fo)) + 1))
        ls = [Layer(fi, fo, 1, 16, 8, 4242 + i, dev, batch=8) for i in range(n)]
        Ws = [torch.randn((fo, fi), device=dev, dtype=torch.float16) for _ in range(min(24, max(4, int(800e6 / (fi * fo * 2)))))]
        per = {}
        for B in (int(v) for v in args.rows.split(",")):
            xb = torch.randn((B, fi), device=dev, dtype=torch.float16)
            e = {}
            g = GraphedCalls([(lambda st, W=W: torch.nn.functional.linear(xb, W)) for W in Ws], dev)
            e["dense_fp16_us"] = g.us_per_pass(args.reps) / len(Ws)
            del g
            if B <= 8 and ls[0].packed is not None:
                gp = GraphedPass(ls, lib, batch=B)
                e["prepacked_matvec_us"] = gp.time_replays(args.reps) * 1e3 / gp.n
                del gp
            g = GraphedCalls([(lambda st, l=l: hk.code1x16_matmat_scan(xb, l.codes, l.codebooks, l.scales, None)) for l in ls], dev)
            e["scan_us"] = g.us_per_pass(args.reps) / len(ls)
            del g
            if not args.no_old and B <= 128:
                keep = _native.get_tuning("gemm_variant")
                _native.set_tuning("gemm_variant", 5)
                try:
                    g = GraphedCalls([(lambda st, l=l: hk.code1x16_matmat_dequant(xb, l.codes, l.codebooks, l.scales, None)) for l in ls], dev)
                    e["gather_mfma_us"] = g.us_per_pass(args.reps) / len(ls)
                    del g
                finally:
                    _native.set_tuning("gemm_variant", keep)
            e["scan_vs_dense"] = e["dense_fp16_us"] / e["scan_us"]
            per[f"B{B}"] = e
            print(f"{fi}->{fo} rows {B:4d}: " + "  ".join(f"{k} {v:7.2f}" for k, v in e.items()), flush=True)
        out[f"{fi}->{fo}"] = per
        del ls, Ws
        torch.cuda.empty_cache()
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
