"""The reference's own GPU kernel on this GPU: checker-side baseline, never part of the product path.

The only GPU kernel of Vahe1994/AQLM that can run on ROCm is its Triton gemv
(inference_lib/src/aqlm/inference_kernels/triton_kernel.py:30-205; the CUDA extension carries inline PTX).  `make -C oracle ref`
(run by `__graft_entry__.build()` in the authoring container) stages that file, unmodified, under the git-ignored
`oracle/_ref/`; it travels to the GPU box with the gpurun snapshot.  This module loads it from there and

  * checks `triton_matmul` against the fp64 oracle and against the HIP operator on the same seeded layer (parity with the
    reference RUNNING, not restated), and
  * times it with the protocol of bench.py (hipGraph replay over layers that together exceed the Infinity Cache, HIP
    events on the capture stream) next to the HIP operator.

Only tools/, bench.py's `gpu_reference_baseline` leg and tests/test_reference_triton.py import this module.

    python tools/reference_triton.py [--quick] [--out profiles/r04_reference_triton.json]
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REF_FILE = os.path.join(ROOT, "oracle", "_ref", "triton_kernel.py")


def load_reference():
    """The staged reference module, or None when it was not staged (no /root/reference at build time) or Triton is absent."""
    if not os.path.isfile(REF_FILE):
        return None
    try:
        import triton  # noqa: F401
    except Exception:
        return None
    spec = importlib.util.spec_from_file_location("aqlm_reference_triton_kernel", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _hip_op(K, nbits):
    import torch

    import aqlm_amd.inference_kernels.hip_kernel  # noqa: F401  (registers torch.ops.aqlm.*)

    if (K, nbits) == (1, 16):
        return torch.ops.aqlm.code1x16_matmat
    return torch.ops.aqlm.codekx8_matmat


def parity_case(ref, K, nbits, g, fin, fout, seed=11, bias=True):
    """max / mean relative error of the reference Triton kernel and of the HIP operator against the fp64 oracle, and of the two
    against each other, on one seeded layer."""
    import numpy as np
    import torch

    from oracle import aqlm_oracle as orc

    L = orc.make_layer(seed, fin, fout, K, nbits, g, batch=1, bias=bias)
    dev = torch.device("cuda:0")
    t = {k: (torch.from_numpy(v).to(dev) if v is not None and hasattr(v, "dtype") else None)
         for k, v in L.items() if k in ("codes", "codebooks", "scales", "x", "bias")}
    y_ref = ref.triton_matmul(t["x"].clone(), t["codes"], t["codebooks"], t["scales"], t["bias"]).float().cpu().numpy()
    y_hip = _hip_op(K, nbits)(t["x"], t["codes"], t["codebooks"], t["scales"], t["bias"]).float().cpu().numpy()
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    den = float(np.mean(np.abs(y64)))
    rel = lambda a, b: float(np.mean(np.abs(a - b)) / den)
    return {"scheme": f"{K}x{nbits}g{g}", "in": fin, "out": fout, "reference_triton_vs_oracle": rel(y_ref, y64),
            "hip_vs_oracle": rel(y_hip, y64), "hip_vs_reference_triton": rel(y_hip, y_ref)}


class _Layers:
    def __init__(self, K, nbits, g, fin, fout, n, dev):
        import torch

        gen = torch.Generator(device=dev).manual_seed(1234 + fin + fout + K)
        cdt = torch.int16 if nbits > 8 else torch.int8
        lo, hi = -(2 ** (nbits - 1)), 2 ** (nbits - 1)
        self.items = []
        for _ in range(n):
            codes = torch.randint(lo, hi, (fout, fin // g, K), generator=gen, device=dev, dtype=torch.int32).to(cdt)
            cb = torch.randn((K, 2**nbits, 1, g), generator=gen, device=dev, dtype=torch.float32).half()
            self.items.append((codes, cb))
        self.scales = torch.ones((fout, 1, 1, 1), device=dev, dtype=torch.float16)
        self.x = torch.randn((1, fin), generator=gen, device=dev, dtype=torch.float32).half()


def _time_graph(fn_per_layer, layers, reps):
    """us per call: `fn_per_layer(codes, codebooks)` over all layers captured in one hipGraph, replayed `reps` times."""
    import torch

    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for c, cb in layers.items:  # eager warm-up: autotune / kernel attributes outside capture; twice, so that the raw HIP
            fn_per_layer(c, cb)     # op's transparent prepack (first call packs, second call hits) is engaged for every layer
            fn_per_layer(c, cb)
    st.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=st):
        for c, cb in layers.items:
            fn_per_layer(c, cb)
    with torch.cuda.stream(st):
        graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            graph.replay()
        e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(layers.items))


def time_case(ref, K, nbits, g, fin, fout, rotate_bytes=600 << 20, reps=5, max_layers=96):
    """Cold-weights timing (layers rotate through more than the Infinity Cache) of the reference Triton gemv and of the HIP op."""
    import torch

    dev = torch.device("cuda:0")
    alg = fout * (fin // g) * K * (1 if nbits <= 8 else 2) + K * (2**nbits) * g * 2 + fin * 2 + fout * 4
    n = max(8, min(max_layers, rotate_bytes // alg + 1))
    layers = _Layers(K, nbits, g, fin, fout, int(n), dev)
    t0 = time.perf_counter()
    ref.triton_matmul(layers.x.clone(), layers.items[0][0], layers.items[0][1], layers.scales, None)  # autotune: 20 configs
    torch.cuda.synchronize()
    tune_s = time.perf_counter() - t0
    us_ref = _time_graph(lambda c, cb: ref.triton_matmul(layers.x, c, cb, layers.scales, None), layers, reps)
    op = _hip_op(K, nbits)
    import aqlm_amd.inference_kernels.hip_kernel as hk

    # the stateless op packs large 1x16 layers behind a cache capped at 1 GiB (least recently used out); this harness cycles
    # through up to 1.4 GB of packed layers to keep the weights cold, so it lifts the cap for its own run
    keep_cap, hk.RAW_OP_PREPACK_MAX_BYTES = hk.RAW_OP_PREPACK_MAX_BYTES, 8 << 30
    try:
        us_hip = _time_graph(lambda c, cb: op(layers.x, c, cb, layers.scales, None), layers, reps)
    finally:
        hk.RAW_OP_PREPACK_MAX_BYTES = keep_cap
        hk.clear_raw_op_prepack_cache()
    best = None
    try:
        best = str(ref._aqlm_gemv_simple.best_config)
    except Exception:
        pass
    return {"scheme": f"{K}x{nbits}g{g}", "in": fin, "out": fout, "layers_rotated": int(n), "algorithmic_bytes": int(alg),
            "reference_triton_us": us_ref, "hip_us": us_hip, "speedup": us_ref / us_hip,
            "reference_triton_GBps": alg / us_ref * 1e-3, "hip_GBps": alg / us_hip * 1e-3,
            "reference_autotune_s": tune_s, "reference_best_config": best}


# every distinct (shape, scheme) costs the reference's autotune 20 Triton compiles, so the list is short
CASES = [(1, 16, 8, 4096, 4096), (2, 8, 8, 4096, 4096), (8, 8, 32, 4096, 4096), (1, 16, 8, 4096, 11008)]


def run(quick=False):
    import torch

    ref = load_reference()
    if ref is None:
        return {"available": False, "why": "oracle/_ref/triton_kernel.py not staged (make -C oracle ref) or triton missing"}
    cases = CASES[:1] if quick else CASES
    out = {"available": True, "source": "inference_lib/src/aqlm/inference_kernels/triton_kernel.py (unmodified, staged by make -C oracle ref)",
           "device": torch.cuda.get_device_name(0), "protocol": "hipGraph replay over distinct layers (> 600 MB rotated), HIP events, batch 1, fp16",
           "parity": [], "timing": []}
    for (K, nbits, g, fin, fout) in cases:
        if fout <= 4096:  # same shape as the timing: the reference's autotune result is reused
            out["parity"].append(parity_case(ref, K, nbits, g, fin, fout))
        out["timing"].append(time_case(ref, K, nbits, g, fin, fout))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="1x16g8 4096x4096 only")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    res = run(a.quick)
    txt = json.dumps(res, indent=1)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
