"""The reference's GPU matvec benchmark protocol on MI355X (benchmark/matmul_benchmark.py of Vahe1994/AQLM: same shapes,
inputs, flags and output lines), so that its only published figures for this path -- "speed-up relative to dense fp16"
(README.md:114-116: 1x16 up to ~1.3x, 2x8 up to ~3.0x) -- have a like-for-like counterpart.

Protocol (matmul_benchmark.py:11-20, 23-34, 83-109): gate_proj shapes of Llama-2 7B / 13B / 70B, x [1, 1, in] fp16 randn,
uniform random codes, randn codebooks, scales = 1, no bias; `warmup_iters` untimed + `benchmark_iters` timed eager calls
between two synchronisations; dense = F.linear on the dequantised fp16 weight.  The quantized call goes through the same
entry the reference script uses (`aqlm.inference_kernels.cuda_kernel.CUDA_KERNEL.code{1x16,2x8}_matmat`); `--module`
additionally times `aqlm.QuantizedLinear.forward` (the path Hugging Face takes, which uses the prepacked kernel for large
1x16 layers) and `--graph` replays both sides from a hipGraph (what a served model does).

    python tools/matmul_benchmark.py [--nbits_per_codebook 16 --num_codebooks 1 --in_group_size 8] [--log_error] [--module] [--graph]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

MODELS = {"Llama 2 7B": [(4096, 11008)], "Llama 2 13B": [(5120, 13824)], "Llama 2 70B": [(8192, 28672)]}


def timed(fn, warmup: int, iters: int) -> float:
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def graphed(fn, warmup: int, iters: int) -> float:
    """One hipGraph holding `iters` calls; replayed once untimed and once timed."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(max(warmup, 2)):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def run(args) -> dict:
    import aqlm
    from aqlm.inference_kernels.cuda_kernel import CUDA_KERNEL
    from aqlm.utils import _dequantize_weight, pack_int_data, unpack_int_data

    dev = torch.device("cuda")
    results = {}
    for model, layers in MODELS.items():
        dense = quant = module = 0.0
        for fin, fout in layers:
            g = torch.Generator(device="cpu").manual_seed(fin + fout)
            x = torch.randn((1, 1, fin), generator=g).to(dev, torch.half)
            codes = pack_int_data(torch.randint(2 ** args.nbits_per_codebook, (fout, fin // args.in_group_size, args.num_codebooks),
                                                generator=g), args.nbits_per_codebook).to(dev)
            codebooks = torch.randn((args.num_codebooks, 2 ** args.nbits_per_codebook, 1, args.in_group_size), generator=g).to(dev, torch.half)
            scales = torch.ones((fout, 1, 1, 1), dtype=torch.half, device=dev)
            weight = _dequantize_weight(unpack_int_data(codes, args.nbits_per_codebook), codebooks, scales).contiguous()
            y_ref = F.linear(x, weight)
            if args.nbits_per_codebook == 16:
                matmul = CUDA_KERNEL.code1x16_matmat
            elif args.num_codebooks == 2:
                matmul = CUDA_KERNEL.code2x8_matmat
            elif args.num_codebooks == 1:
                matmul = CUDA_KERNEL.code1x8_matmat
            else:  # e.g. 8x8 g32: the reference's script has no case for it (its selector sends such schemes to Triton)
                matmul = torch.ops.aqlm.codekx8_matmat
            y = matmul(x, codes, codebooks, scales, None)
            rel = float((y_ref.float() - y.float()).abs().mean() / y_ref.float().abs().mean())
            if args.log_error:
                print(f"Relative error: {rel:.2e}")
            bench = graphed if args.graph else timed
            d = bench(lambda: F.linear(x, weight, out=y_ref), args.warmup_iters, args.benchmark_iters)
            q = bench(lambda: matmul(x, codes, codebooks, scales, None), args.warmup_iters, args.benchmark_iters)
            dense += d
            quant += q
            if args.module:
                lin = aqlm.QuantizedLinear(fin, fout, args.in_group_size, 1, args.num_codebooks, args.nbits_per_codebook, bias=False,
                                           device=dev, dtype=torch.half)
                with torch.no_grad():
                    lin.codes.copy_(codes)
                    lin.codebooks.copy_(codebooks)
                    lin.scales.copy_(scales)
                    ym = lin(x)  # first call: kernel lookup (+ load-time prepack for large 1x16 layers)
                    relm = float((y_ref.float() - ym.float()).abs().mean() / y_ref.float().abs().mean())
                    assert relm < 2e-3, relm
                    module += bench(lambda: lin(x), args.warmup_iters, args.benchmark_iters)
                del lin
            del weight, codes
        print(f"{model}: Dense forward = {dense * 1e6:.0f} mus")
        print(f"{model}: Quant forward = {quant * 1e6:.0f} mus")
        print(f"{model}: Speedup relative to dense = {(dense / quant):.3f}")
        results[model] = {"dense_us": dense * 1e6, "quant_us": quant * 1e6, "speedup": dense / quant, "relative_error": rel}
        if args.module:
            print(f"{model}: QuantizedLinear forward = {module * 1e6:.0f} mus, speedup relative to dense = {(dense / module):.3f}")
            results[model].update({"module_us": module * 1e6, "module_speedup": dense / module})
    return results


def main():
    parser = argparse.ArgumentParser(add_help=True)
    parser.add_argument("--warmup_iters", type=int, default=10)
    parser.add_argument("--benchmark_iters", type=int, default=10)
    parser.add_argument("--log_error", action="store_true")
    parser.add_argument("--nbits_per_codebook", type=int, default=16)
    parser.add_argument("--num_codebooks", type=int, default=1)
    parser.add_argument("--in_group_size", type=int, default=8)
    parser.add_argument("--module", action="store_true", help="also time aqlm.QuantizedLinear.forward (prepacked path for large 1x16 layers)")
    parser.add_argument("--graph", action="store_true", help="time hipGraph replays instead of eager calls")
    parser.add_argument("--json", default=None, help="write the results to this file")
    args = parser.parse_args()
    assert torch.cuda.is_available(), "needs an MI355X"
    res = run(args)
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"args": vars(args), "results": res}, f, indent=1)


if __name__ == "__main__":
    main()
