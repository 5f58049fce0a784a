"""cpu_baseline (the oracle's C port of the reference CPU path, timed on the host cores) and gpu_reference_baseline (the reference's Triton gemv on this GPU): the checker-side legs of bench.py."""
import os
import sys
import time

import numpy as np
import torch

from .layers import ROOT, _time_calls, algorithmic_bytes


def oracle_parity(L, got):
    """mean relative error of one bench layer's output (`got`, fp32 [1, out]) against the C oracle's dequantise + gemv on the host
    (oracle/aqlm_oracle.c: the restated reference CPU path; checker only, outside every timed region)."""
    from oracle import c_oracle

    k = c_oracle.DequantGemv(L.codebooks.float().cpu().numpy(), L.codes.cpu().numpy(), L.scales.float().cpu().numpy(), None, 16,
                             nthreads=c_oracle.max_threads())
    y_or = torch.from_numpy(np.array(k(L.x[0].float().cpu().numpy()), copy=True))
    return float((got[0].cpu() - y_or).abs().mean() / y_or.abs().mean())


def cpu_baseline(sample_seconds=24.0):
    """The reference's CPU side, timed on this box's host cores (numba is not installable, so the
    kernels are the oracle's C restatements -- "kind": "port"):
      * `value`: what the reference EXECUTES on CPU for 1x16 (dequantize + F.linear, kernel_selector.py:99-102), all
        cores, on the two headline shapes -- algorithmic GB/s, comparable with the GPU `value`;
      * `protocol`: benchmark/matmul_benchmark_cpu.py's own protocol (10 warm-up + up to 1000 timed calls, :43-54; one
        thread as the script defaults, :77-87, and all cores) for its LUT gemv (:100-111 == numba_kernel.py:37-48) on
        the script's default scheme 2x8g8 and on 1x16g8 with u16 codes, both 4096 x 4096.  The sample is bounded
        (about sample_seconds in total): the iteration count actually run is reported."""
    from oracle import aqlm_oracle as orc
    from oracle import c_oracle

    threads = c_oracle.max_threads()
    total_bytes, total_time, per_shape = 0, 0.0, {}
    budget = sample_seconds / 8.0
    for fin, fout in ((4096, 4096), (4096, 11008)):
        L = orc.make_layer(0, fin, fout, 1, 16, 8, batch=1, bias=False, float_dtype=np.float32, edge_codes=False)
        k = c_oracle.DequantGemv(L["codebooks"], L["codes"], L["scales"], None, 16, nthreads=threads)
        x = L["x"][0]
        tw = time.perf_counter()
        while time.perf_counter() - tw < 1.0:  # let the OpenMP pool spin up (first parallel regions are 10x slow)
            k(x)
        st, n = _time_calls(lambda: k(x), budget, 1000, 10)
        b = algorithmic_bytes(fin, fout)
        dt = st["median"]  # all-core OpenMP calls on a shared host have heavy stragglers: the median is the repeatable figure
        per_shape[f"{fin}x{fout}"] = {"ms_median": dt * 1e3, "ms_mean": st["mean"] * 1e3, "ms_min": st["min"] * 1e3,
                                      "GBps": b / dt * 1e-9, "iters": n}
        total_bytes += b
        total_time += dt
    protocol = {}
    for name, (K, nbits) in (("2x8g8", (2, 8)), ("1x16g8_u16_codes", (1, 16))):
        L = orc.make_layer(1, 4096, 4096, K, nbits, 8, batch=1, bias=False, float_dtype=np.float32, edge_codes=False)
        x = L["x"][0]
        codes_alt = orc.permute_codes_for_lut(L["codes"])  # [in_groups, out, K], the script's layout (:114-119)
        b = algorithmic_bytes(4096, 4096, K, nbits, 8)
        for label, nt in (("1_thread", 1), (f"{threads}_threads", threads)):
            lk = c_oracle.LutGemv(L["codebooks"], codes_alt, L["scales"], nbits, nthreads=nt)
            st, n = _time_calls(lambda: lk(x), budget, 1000, 10 if nbits == 8 else 1)
            protocol[f"{name}_{label}"] = {"ms_mean": st["mean"] * 1e3, "ms_median": st["median"] * 1e3,
                                           "GBps_algorithmic": b / st["mean"] * 1e-9, "iters": n}
    # the product's own CPU kernels (libaqlm_cpu.so, what `QuantizedLinear` runs for CPU tensors: SURVEY.md 8(f) item 4),
    # same layers and protocol, fp32 torch tensors through aqlm_amd.inference_kernels.cpu_kernel
    native = {}
    try:
        from aqlm_amd.inference_kernels import cpu_kernel as ck

        for name, (K, nbits) in (("2x8g8", (2, 8)), ("1x16g8", (1, 16))):
            L = orc.make_layer(1, 4096, 4096, K, nbits, 8, batch=1, bias=False, float_dtype=np.float32, edge_codes=False)
            xt = torch.from_numpy(np.ascontiguousarray(L["x"][:1]))
            cbt = torch.from_numpy(np.ascontiguousarray(L["codebooks"]))
            sct = torch.from_numpy(np.ascontiguousarray(L["scales"]))
            signed = orc.pack_int_data(L["codes"], nbits)
            codes_t = torch.from_numpy(np.ascontiguousarray(signed))
            b = algorithmic_bytes(4096, 4096, K, nbits, 8)
            for label, nt in (("1_thread", 1), (f"{threads}_threads", threads)):
                if nbits == 8:
                    alt = ck.permute_codes_for_lut(codes_t)
                    fn = lambda: ck.cpu_gemm_lut(xt, alt, cbt, sct, None, nthreads=nt)  # noqa: E731
                else:
                    fn = lambda: ck.cpu_gemv_1xn(xt, codes_t, cbt, sct, None, nthreads=nt)  # noqa: E731
                st, n = _time_calls(fn, budget / 2, 1000, 10)
                native[f"{name}_{label}"] = {"ms_mean": st["mean"] * 1e3, "ms_median": st["median"] * 1e3,
                                             "GBps_algorithmic_median": b / st["median"] * 1e-9, "iters": n}
    except Exception as e:  # noqa: BLE001 - a reported extra, never fatal for the GPU benchmark
        native = {"error": f"{type(e).__name__}: {e}"}
    # what the reference itself EXECUTES on CPU for 1x16 (kernel_selector.py:99-102): the pure-torch dequantize_gemm
    # (dequantization.py:9-21 + utils.py:43-70: embedding_bag gather, reshape, F.linear).  /root/reference does not exist
    # on the GPU box, so this is aqlm_amd's module of the same name and semantics (checked against the reference's
    # outputs by tests/golden); fp32, batch 1, 4096 x 4096, one thread and all cores, a handful of calls each.
    ref_torch = {}
    try:
        from aqlm_amd.inference_kernels.dequantization import dequantize_gemm

        L = orc.make_layer(2, 4096, 4096, 1, 16, 8, batch=1, bias=False, float_dtype=np.float32, edge_codes=False)
        xt = torch.from_numpy(np.ascontiguousarray(L["x"][:1]))
        cbt = torch.from_numpy(np.ascontiguousarray(L["codebooks"]))
        sct = torch.from_numpy(np.ascontiguousarray(L["scales"]))
        codes_t = torch.from_numpy(np.ascontiguousarray(orc.pack_int_data(L["codes"], 16)))
        b = algorithmic_bytes(4096, 4096)
        keep = torch.get_num_threads()
        for label, nt in (("1_thread", 1), (f"{threads}_threads", threads)):
            torch.set_num_threads(nt)
            st, n = _time_calls(lambda: dequantize_gemm(xt, codes_t, cbt, sct, None), 2.5, 20, 2)
            ref_torch[f"1x16g8_4096x4096_{label}"] = {"ms_median": st["median"] * 1e3, "ms_min": st["min"] * 1e3,
                                                      "GBps_algorithmic_median": b / st["median"] * 1e-9, "iters": n}
        torch.set_num_threads(keep)
    except Exception as e:  # noqa: BLE001
        ref_torch = {"error": f"{type(e).__name__}: {e}"}
    return {
        "value": total_bytes / total_time * 1e-9,
        "unit": "GB/s",
        "cores": threads,
        "kind": "port",
        "reference_torch_path": ref_torch,
        "native_cpu_path": native,
        "sample": f"oracle C dequant-gemv (what the reference runs on CPU for 1x16), fp32, one 4096->4096 + one 4096->11008 "
                  f"layer, <= 1000 calls or {budget:.0f} s each on {threads} OpenMP threads; `protocol`: the reference "
                  f"benchmark's LUT gemv (matmul_benchmark_cpu.py) restated in C, 4096x4096, 10 warm-up + <= 1000 calls",
        "per_shape": per_shape,
        "protocol": protocol,
    }


def gpu_reference_baseline():
    """The reference's own GPU kernel on this GPU, beside `cpu_baseline`: its Triton gemv (triton_kernel.py:30-205 -- the only reference
    kernel that runs on ROCm; its CUDA extension carries inline PTX), staged unmodified under oracle/_ref/ by `make -C oracle ref`,
    timed with this file's protocol (hipGraph replay over > 600 MB of distinct layers) on 1x16g8 4096 -> 4096 next to the HIP operator.
    Checker side only: nothing under aqlm_amd/ imports it.  All four cases: profiles/r04_reference_triton.json."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import reference_triton as rt

        res = rt.run(quick=True)
        if not res.get("available"):
            return {"available": False, "why": res.get("why")}
        t, p = res["timing"][0], res["parity"][0]
        return {"available": True, "kind": "reference", "kernel": "aqlm.inference_kernels.triton_kernel.triton_matmul (Triton, autotuned)",
                "workload": "1x16g8 4096->4096, bs=1, cold (layers rotated through > 600 MB), hipGraph", "us": t["reference_triton_us"],
                "value": t["reference_triton_GBps"], "unit": "GB/s", "hip_operator_us": t["hip_us"], "hip_speedup": t["speedup"],
                "reference_autotune_s": t["reference_autotune_s"], "parity_mean_rel": p}
    except Exception as e:  # noqa: BLE001 - a reported extra, never fatal for the benchmark
        return {"available": False, "why": f"{type(e).__name__}: {e}"}

