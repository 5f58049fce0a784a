"""The reference's CPU matvec benchmark protocol (benchmark/matmul_benchmark_cpu.py of Vahe1994/AQLM: same shapes, inputs,
flags and output lines) on the native CPU kernels of this package (libaqlm_cpu.so) -- BASELINE.json config 1.  The
reference times a numba-compiled look-up-table gemv (numba is not installable here); its published figure for this path
is "up to ~4.0x over fp32 dense" (README.md:117).

Protocol (matmul_benchmark_cpu.py:3-4, 11-24, 27-38, 100-160): gate_proj shapes of Llama-2 7B / 13B / 70B, fp32,
x [1, in], uniform random codes in the CPU layout [in_groups, out, K], randn codebooks and scales, `nthreads` threads for
both sides (default 1), `warmup_iters` untimed + `benchmark_iters` timed calls; dense = F.linear on the dequantised fp32
weight.  K x 8-bit schemes run `aqlm_cpu_gemv_lut_kx8` (the reference's algorithm); one 16-bit codebook runs
`aqlm_cpu_gemv_1xn` (the reference script cannot time that scheme: it reinterprets the codes as bytes, :140).

    python tools/matmul_benchmark_cpu.py [--nbits_per_codebook 8 --num_codebooks 2 --in_group_size 8] [--nthreads 1] [--log_error]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

MODELS = {"Llama 2 7B": [(4096, 11008)], "Llama 2 13B": [(5120, 13824)], "Llama 2 70B": [(8192, 28672)]}


def timed(fn, warmup: int, iters: int) -> float:
    for _ in range(warmup):
        fn()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t0) / iters


def main():
    parser = argparse.ArgumentParser(add_help=True)
    parser.add_argument("--warmup_iters", type=int, default=10)
    parser.add_argument("--benchmark_iters", type=int, default=1000)
    parser.add_argument("--log_error", action="store_true")
    parser.add_argument("--nbits_per_codebook", type=int, default=8)
    parser.add_argument("--num_codebooks", type=int, default=2)
    parser.add_argument("--fp16_codebooks", action="store_true",
                        help="codebook values rounded to fp16 as in a checkpoint (the reference protocol draws fp32 randn): lets the 1 x n kernel use its fp16 table")
    parser.add_argument("--in_group_size", type=int, default=8)
    parser.add_argument("--nthreads", type=int, default=1)
    parser.add_argument("--max_seconds", type=float, default=20.0, help="cap of the timed loop per side and shape (0 = none)")
    parser.add_argument("--json", default=None)
    args = parser.parse_args()
    os.environ.setdefault("OMP_NUM_THREADS", str(args.nthreads))

    import torch
    import torch.nn.functional as F

    from aqlm_amd.inference_kernels import cpu_kernel as ck
    from aqlm_amd.utils import _dequantize_weight, pack_int_data, unpack_int_data

    torch.set_num_threads(args.nthreads)
    K, nbits, g = args.num_codebooks, args.nbits_per_codebook, args.in_group_size
    results = {}
    for model, layers in MODELS.items():
        dense = quant = 0.0
        rel = None
        for fin, fout in layers:
            gen = torch.Generator().manual_seed(fin + fout)
            x = torch.randn((1, fin), generator=gen, dtype=torch.float32)
            codes = pack_int_data(torch.randint(2 ** nbits, (fout, fin // g, K), generator=gen), nbits)   # canonical [out, in/g, K]
            codebooks = torch.randn((K, 2 ** nbits, 1, g), generator=gen, dtype=torch.float32)
            if args.fp16_codebooks:
                codebooks = codebooks.half().float()
            scales = torch.randn((fout, 1, 1, 1), generator=gen, dtype=torch.float32)
            weight = _dequantize_weight(unpack_int_data(codes, nbits), codebooks, scales).contiguous()
            y_ref = F.linear(x, weight)
            if nbits == 8:
                alt = ck.permute_codes_for_lut(codes)    # [in_groups, out, K] uint8: the reference's CPU layout
                fn = lambda: ck.cpu_gemm_lut(x, alt, codebooks, scales, None, nthreads=args.nthreads)  # noqa: E731
            elif K == 1:
                fn = lambda: ck.cpu_gemv_1xn(x, codes, codebooks, scales, None, nthreads=args.nthreads)  # noqa: E731
            else:
                raise SystemExit("the native CPU kernels cover K x 8-bit and 1 x n-bit schemes")
            y = fn()
            rel = float((y_ref - y).abs().mean() / y_ref.abs().mean())
            if args.log_error:
                print(f"Relative error: {rel:.2e}")

            def iters_for(f):
                if not args.max_seconds:
                    return args.benchmark_iters
                t1 = time.perf_counter()
                f()
                one = max(time.perf_counter() - t1, 1e-6)
                return max(3, min(args.benchmark_iters, int(args.max_seconds / one)))

            dense_fn = lambda: F.linear(x, weight, out=y_ref)  # noqa: E731
            dense += timed(dense_fn, args.warmup_iters, iters_for(dense_fn))
            quant += timed(fn, args.warmup_iters, iters_for(fn))
            del weight
        print(f"{model}: Dense forward = {dense * 1e3:.2f} ms")
        print(f"{model}: Quant forward = {quant * 1e3:.2f} ms")
        print(f"{model}: Speedup relative to dense = {(dense / quant):.3f}")
        results[model] = {"dense_ms": dense * 1e3, "quant_ms": quant * 1e3, "speedup": dense / quant, "relative_error": rel}
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"args": vars(args), "nproc": os.cpu_count(), "results": results}, f, indent=1)


if __name__ == "__main__":
    main()
