#!/usr/bin/env python3
"""bench.py -- the driver's measurement contract for the AQLM QuantizedLinear matvec path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "1x16g8 matvec, Llama-3-8B linear shapes (4096->4096/11008), bs=1"):
one STEP = one decode token's pass, batch 1, through a stack of 32 blocks, each block = one 4096->4096 and one
4096->11008 1x16g8 QuantizedLinear matvec.  All 64 layers are distinct instances (own codes AND own codebook, like a
real model), 564 MB of algorithmic bytes per step, so every step streams its weights from HBM (cold: larger than the
256 MiB Infinity Cache).  The step is captured once in a hipGraph (launch-bound otherwise) and replayed.

value = algorithmic GB/s of the whole job = ranks x bytes-per-step / step time, inputs resident in HBM.
N > 1 = N independent replicas of the step (one process per GPU, no data-path collective: decode data parallelism),
"scaling": "weak".  The north star's row-sharded 70B layer + RCCL all-reduce is measured next to it and reported in
"sharded_70b" (it is an extra, never the headline value).

Extra objects: "roofline" (dominant kernel = the 1x16 gemv; duration from HIP events over the timed region on the
launch stream, i.e. including inter-launch gaps), "cpu_baseline" (oracle C port of the reference CPU path on the host
cores, rank 0, bounded sample), "gpu_reference_baseline" (the reference's own Triton gemv on this GPU, staged under
oracle/_ref/ -- checker side only), "parity_mean_rel_vs_cpu_oracle" (tripwire against the C oracle, outside the timed region),
"config" (workload + the load-time / memory price of the prepacked path), "detail" (per-shape cold/warm timings, Llama-3-8B /
Llama-2-7B / Llama-3-70B tokens/s, roofline objects of BASELINE configs 3 and 4, 2..8-row cross-over), "sharded_70b"
(+ the two tensor-parallel plans of the 70B MLP).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
# ceilings of the headline metric (see roofline.ceiling below and BASELINE.md section 3)
_A, _B, _BOUNDARY_US = 5_267_456, 12_372_992, 1.45
CEILING_LAUNCH = (_A + _B) / ((_A + _B) / 8e6 + 2 * _BOUNDARY_US) / 8e6          # 0.43: perfect kernels behind the launch boundary
CEILING_GATHER = 0.32                                                          # two LDS gathers per code (profiles/r01_call1_mb_ldsgather.log)
CEILING_BOTH = (_A + _B) / ((_A + _B) / (CEILING_GATHER * 8e6) + 2 * _BOUNDARY_US) / 8e6   # 0.225
MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak (AMD's 5 PF figure includes 2:1 sparsity)
# 1x16 g8 layers with at least this many codes run the prepacked (slice-bucketed) decode kernel, like
# aqlm_amd.inference.PREPACK_MIN_CODES; --no-packed sets it to 0 (direct L2-gather kernel everywhere).
PACK_MIN_OUT = 500_000


def algorithmic_bytes(fin, fout, K=1, nbits=16, g=8, batch=1, bias=False):
    """SURVEY.md section 8(d)."""
    n = fout * (fin // g) * K * (1 if nbits <= 8 else 2) + K * (2**nbits) * g * 2
    n += batch * fin * 2 + batch * fout * 2 + fout * 2 + (fout * 2 if bias else 0)
    return n


# load-time price of the prepacked path over every layer built so far (reset by main() around the timed workload)
PREPACK_STATS = {"seconds": 0.0, "layers": 0, "packed_bytes": 0, "canonical_code_bytes": 0, "weights": 0}


class Layer:
    """One synthetic QuantizedLinear instance resident in HBM (mirrors benchmark/matmul_benchmark.py:83-97:
    uniform random codes, randn codebooks, scales = 1, no bias)."""

    def __init__(self, fin, fout, K, nbits, g, seed, device, batch=1, code_law=None):
        gen = torch.Generator(device=device).manual_seed(seed)
        self.seed = seed
        self.fin, self.fout, self.K, self.nbits, self.g = fin, fout, K, nbits, g
        cdt = torch.int16 if nbits > 8 else torch.int8
        lo, hi = (-(2 ** (nbits - 1)), 2 ** (nbits - 1))
        if code_law is None:
            self.codes = torch.randint(lo, hi, (fout, fin // g, K), generator=gen, device=device, dtype=torch.int32).to(cdt)
        else:
            # code_law = (alpha, labels sorted by frequency?): the entry of rank r is used with probability ~ (r + 1)^-alpha --
            # what k-means + beam search leave behind is not uniform (src/aq.py:286-356 of the reference); 16-bit codes only
            alpha, sorted_labels = code_law
            prob = torch.arange(1, 2**nbits + 1, dtype=torch.float64, device=device) ** (-alpha)
            rank_of = torch.multinomial((prob / prob.sum()).float(), fout * (fin // g) * K, replacement=True, generator=gen)
            labels = (torch.arange(2**nbits, device=device) if sorted_labels
                      else torch.randperm(2**nbits, generator=gen, device=device))
            unsigned = labels[rank_of].reshape(fout, fin // g, K).to(torch.int32)
            self.codes = (unsigned - (unsigned >= hi) * 2**nbits).to(cdt)
        self.codebooks = torch.randn((K, 2**nbits, 1, g), generator=gen, device=device, dtype=torch.float32).half()
        self.scales = torch.ones((fout, 1, 1, 1), device=device, dtype=torch.float16)
        self.x = torch.randn((batch, fin), generator=gen, device=device, dtype=torch.float32).half()
        self.y = torch.empty((batch, fout), device=device, dtype=torch.float16)
        self.bytes = algorithmic_bytes(fin, fout, K, nbits, g, batch)
        self.packed = None
        self.planar = None
        if PACK_MIN_OUT and (K, nbits) == (8, 8):
            from aqlm_amd.inference_kernels import hip_kernel as hk

            self.planar = hk.planar_8x8_pack(self.codes, g, codebooks=self.codebooks)  # load-time re-layout (same size, lossless)
        if PACK_MIN_OUT and (K, nbits) == (1, 16) and g in (8, 16) and fout * (fin // g) >= PACK_MIN_OUT:
            from aqlm_amd import _native

            self.prepack(_native.lib)

    def alg_bytes(self, batch=1):
        return algorithmic_bytes(self.fin, self.fout, self.K, self.nbits, self.g, batch)

    def prepack(self, lib):
        """One-off load-time repack for the slice-bucketed decode kernel (layers with >= PACK_MIN_OUT codes)."""
        from aqlm_amd.inference_kernels import hip_kernel as hk

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.packed = hk.prepack_1x16(self.codes, self.g, codebooks=self.codebooks)  # + the codebook range: single-kernel matvecs
        torch.cuda.synchronize()
        PREPACK_STATS["seconds"] += time.perf_counter() - t0
        if self.packed is not None:
            PREPACK_STATS["layers"] += 1
            PREPACK_STATS["packed_bytes"] += int(self.packed.buf.numel() * self.packed.buf.element_size())
            PREPACK_STATS["canonical_code_bytes"] += int(self.codes.numel() * self.codes.element_size())
            PREPACK_STATS["weights"] += self.fin * self.fout
            nb = self.x.shape[0]
            self.ws = torch.empty((self.packed.slices * nb * self.fout,), dtype=torch.float32, device=self.codes.device)

    def launch(self, lib, stream, batch=1):
        import ctypes

        from aqlm_amd import _native

        if batch <= self.x.shape[0] and getattr(self, "packed", None) is not None:
            rc = lib.aqlm_hip_gemv_1x16_packed(ctypes.byref(self.packed.desc), self.packed.data_ptr(),
                                               self.codebooks.data_ptr(), self.scales.data_ptr(), None, self.x.data_ptr(),
                                               self.y.data_ptr(), batch, self.fin, self.fout, _native.F16,
                                               self.ws.data_ptr(), self.ws.numel() * 4, stream)
        elif batch <= self.x.shape[0] and getattr(self, "fused_8x8", False) and (self.K, self.nbits, self.g) == (8, 8, 32):
            # 8x8 g32 beyond one row: the codebooks in LDS, one MFMA per codebook and k-step (aqlm_hip_gemm_8x8_mfma, round 5)
            rc = lib.aqlm_hip_gemm_8x8_mfma(self.codes.data_ptr(), self.codebooks.data_ptr(), self.scales.data_ptr(), None, self.x.data_ptr(),
                                            self.y.data_ptr(), batch, self.fout, self.fin, self.g, self.fin, self.fout, _native.F16, stream)
        elif batch <= self.x.shape[0] and self.K == 8 and self.nbits == 8 and (batch == 1 or (self.planar is not None and getattr(self, "lut_rows", True))):
            if getattr(self, "lut_cells", None) is None or self.lut_cells.numel() < batch * self.fout:  # zero-at-rest accumulator cells of the single-kernel form
                self.lut_cells = torch.zeros((self.x.shape[0] * self.fout,), dtype=torch.int64, device=self.codes.device)
            if self.planar is not None and batch > 1:  # 2+ rows: one launch of rows x the single-row workgroups (round 5)
                rc = lib.aqlm_hip_gemv_8x8_lut_batch(self.planar.data_ptr(), self.codebooks.data_ptr(), self.scales.data_ptr(), None,
                                                     self.x.data_ptr(), self.y.data_ptr(), self.fout, self.fin, self.g, batch, self.fin, self.fout,
                                                     _native.F16, 1, self.planar.codebook_absmax, self.lut_cells.data_ptr(),
                                                     self.lut_cells.numel() * 8, stream)
                if rc:
                    _native.check(rc)
                return
            if self.planar is not None:
                rc = lib.aqlm_hip_gemv_8x8_lut_planar(self.planar.data_ptr(), self.codebooks.data_ptr(), self.scales.data_ptr(), None,
                                                      self.x.data_ptr(), self.y.data_ptr(), self.fout, self.fin, self.g, _native.F16,
                                                      self.planar.codebook_absmax, self.lut_cells.data_ptr(),
                                                      self.lut_cells.numel() * 8, 1, stream)
                if rc:
                    _native.check(rc)
                return
            rc = lib.aqlm_hip_gemv_8x8_lut_fused(self.codes.data_ptr(), self.codebooks.data_ptr(), self.scales.data_ptr(), None,
                                                 self.x.data_ptr(), self.y.data_ptr(), self.fout, self.fin, self.g, _native.F16,
                                                 self.lut_cells.data_ptr(), self.lut_cells.numel() * 8, stream)
        elif self.nbits == 16:
            rc = lib.aqlm_hip_gemv_1x16(self.codes.data_ptr(), self.codebooks.data_ptr(), self.scales.data_ptr(), None,
                                        self.x.data_ptr(), self.y.data_ptr(), self.fout, self.fin, self.g, batch,
                                        self.fin, self.fout, _native.F16, stream)
        elif batch > _native.MAX_GEMV_BATCH:  # 9+ rows of 1x8 / 2x8: the fused dequant -> MFMA op (the raw ops send them there too)
            rc = lib.aqlm_hip_gemm_kx8_mfma(self.codes.data_ptr(), self.codebooks.data_ptr(), self.scales.data_ptr(), None, self.x.data_ptr(),
                                            self.y.data_ptr(), batch, self.fout, self.fin, self.K, self.g, self.fin, self.fout, _native.F16, stream)
        else:
            rc = lib.aqlm_hip_gemv_kx8(self.codes.data_ptr(), self.codebooks.data_ptr(), self.scales.data_ptr(), None,
                                       self.x.data_ptr(), self.y.data_ptr(), self.fout, self.fin, self.K, self.g, batch,
                                       self.fin, self.fout, _native.F16, stream)
        if rc:
            _native.check(rc)


class FusedLayers:
    """Several 1x16 layers applied to one x in ONE launch (aqlm_hip_gemv_1x16_multi, or the prepacked variant when
    every member is prepacked): the q/k/v or gate/up projections of a decoder block."""

    def __init__(self, members, mode="auto"):
        import ctypes

        from aqlm_amd import _native

        self.members = members
        self.fin, self.g = members[0].fin, members[0].g
        self.n_matvecs = len(members)
        self.packed = mode != "direct" and all(m.packed is not None for m in members)
        if mode == "packed" and not self.packed:
            for m in members:
                m.prepack(_native.lib)
            self.packed = all(m.packed is not None for m in members)
        self.x = members[0].x
        self.segs = (_native.Segment * len(members))()
        for sg, m in zip(self.segs, members):
            sg.codes = m.packed.data_ptr() if self.packed else m.codes.data_ptr()
            sg.codebook, sg.scales, sg.bias = m.codebooks.data_ptr(), m.scales.data_ptr(), None
            sg.y, sg.y_row_stride, sg.out_features = m.y.data_ptr(), m.fout, m.fout
        if self.packed:
            self.ws = torch.empty((members[0].packed.slices * sum(m.fout for m in members),), dtype=torch.float32, device=self.x.device)
            self.descs = (_native._descp * len(members))(*[ctypes.pointer(m.packed.desc) for m in members])

    def alg_bytes(self, batch=1):
        return sum(m.alg_bytes(batch) for m in self.members)

    def launch(self, lib, stream, batch=1):
        from aqlm_amd import _native

        if self.members[0].nbits == 8 and self.members[0].K == 8 and batch == 1:
            m0 = self.members[0]
            if getattr(self, "lut_cells", None) is None:  # zero-at-rest accumulator cells of the single-kernel form
                self.lut_cells = torch.zeros((sum(m.fout for m in self.members),), dtype=torch.int64, device=self.x.device)
            if all(getattr(m, "planar", None) is not None for m in self.members):
                import ctypes

                if getattr(self, "planar_segs", None) is None:
                    self.planar_segs = (_native.Segment * len(self.members))()
                    for sg, src, m in zip(self.planar_segs, self.segs, self.members):
                        sg.codes, sg.codebook, sg.scales, sg.bias = m.planar.data_ptr(), src.codebook, src.scales, None
                        sg.y, sg.y_row_stride, sg.out_features = src.y, src.y_row_stride, src.out_features
                    self.planar_absmax = (ctypes.c_float * len(self.members))(*[m.planar.codebook_absmax for m in self.members])
                rc = lib.aqlm_hip_gemv_8x8_lut_planar_multi(self.planar_segs, self.planar_absmax, len(self.members), self.x.data_ptr(),
                                                            self.fin, self.g, _native.F16, self.lut_cells.data_ptr(),
                                                            self.lut_cells.numel() * 8, 1, stream)
                if rc:
                    _native.check(rc)
                return
            rc = lib.aqlm_hip_gemv_8x8_lut_multi_fused(self.segs, len(self.members), self.x.data_ptr(), self.fin, self.g,
                                                       _native.F16, self.lut_cells.data_ptr(), self.lut_cells.numel() * 8, stream)
        elif self.members[0].nbits == 8:
            m0 = self.members[0]
            rc = lib.aqlm_hip_gemv_kx8_multi(self.segs, len(self.members), self.x.data_ptr(), self.fin, m0.K, self.g, batch,
                                             self.fin, _native.F16, stream)
        elif self.packed and batch == 1:
            rc = lib.aqlm_hip_gemv_1x16_packed_multi(self.segs, self.descs, len(self.members), self.x.data_ptr(), self.fin,
                                                     1, self.fin, _native.F16, self.ws.data_ptr(), self.ws.numel() * 4,
                                                     stream)
        else:
            rc = lib.aqlm_hip_gemv_1x16_multi(self.segs, len(self.members), self.x.data_ptr(), self.fin, self.g, batch,
                                              self.fin, _native.F16, stream)
        if rc:
            _native.check(rc)


class GraphedPass:
    """A list of layer launches captured once into a hipGraph on a side stream."""

    def __init__(self, layers, lib, batch=1):
        self.layers, self.n = layers, len(layers)
        self.bytes = sum(l.alg_bytes(batch) for l in layers)
        self.stream = torch.cuda.Stream()
        with torch.cuda.stream(self.stream):
            for l in layers:  # eager warm-up (also sets kernel attributes outside capture)
                l.launch(lib, self.stream.cuda_stream, batch)
        self.stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            s = torch.cuda.current_stream().cuda_stream
            for l in layers:
                l.launch(lib, s, batch)

    def time_replays(self, reps, warmup=2):
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                self.graph.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(self.stream)
            for _ in range(reps):
                self.graph.replay()
            e1.record(self.stream)
        e1.synchronize()
        return e0.elapsed_time(e1) / reps  # ms per replay


def _time_calls(fn, budget_s, max_iters, warmup):
    for _ in range(warmup):
        fn()
    times, t0 = [], time.perf_counter()
    while len(times) < max_iters and time.perf_counter() - t0 < budget_s:
        t1 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t1)
    return {"mean": float(np.mean(times)), "median": float(np.median(times)), "min": float(np.min(times))}, len(times)


def code_histograms_detail(lib, dev, rank, reps, nblocks, uniform_value):
    """The headline step on codes that use the codebook unevenly (round 5): 32 x {4096->4096, 4096->11008}
    distinct layers per case, Zipf-distributed codes, labels shuffled and sorted by frequency.  Format v7 of the prepacked path
    balances the slices at pack time (relabelling; a variable row-group geometry where one entry outweighs a slice), so every
    case runs the packed kernel -- the reference's kernels are data-oblivious (cuda_kernel.cu:16-27), these figures say how
    close to that the slice-bucketed kernel stays."""
    out = {"protocol": "the timed step's layer list (one hipGraph, cold: 564 MB per step), codes ~ Zipf(alpha) over the 65536 entries",
           "uniform_GBps": uniform_value, "cases": {}}
    for alpha in (0.5, 0.8, 1.0, 1.2):
        for sorted_labels in (False, True):
            before = dict(PREPACK_STATS)
            layers = []
            for i in range(nblocks):
                layers.append(Layer(4096, 4096, 1, 16, 8, 70000 + rank * 10000 + 2 * i, dev, code_law=(alpha, sorted_labels)))
                layers.append(Layer(4096, 11008, 1, 16, 8, 70000 + rank * 10000 + 2 * i + 1, dev, code_law=(alpha, sorted_labels)))
            gp = GraphedPass(layers, lib)
            ms = gp.time_replays(reps)
            packed = [l.packed for l in layers if l.packed is not None]
            gbps = gp.bytes / (ms * 1e-3) * 1e-9
            out["cases"][f"zipf{alpha}_{'sorted' if sorted_labels else 'shuffled'}_labels"] = {
                "GBps": gbps, "vs_uniform": gbps / uniform_value, "ms_per_step": ms,
                "layers_on_the_packed_kernel": len(packed), "layers": len(layers),
                "relabelled": sum(1 for p in packed if p.desc.relabelled),
                "variable_geometry": sum(1 for p in packed if p.desc.variable_geometry),
                "workgroups_per_slice_min_max": [min(min(list(p.desc.slice_groups)[:16]) for p in packed) if packed else None,
                                                 max(max(list(p.desc.slice_groups)[:16]) for p in packed) if packed else None],
                "prepack_ms_per_layer": (PREPACK_STATS["seconds"] - before["seconds"]) * 1e3 / max(1, len(packed)),
                "packed_bits_per_weight": 8.0 * (PREPACK_STATS["packed_bytes"] - before["packed_bytes"]) / max(1, PREPACK_STATS["weights"] - before["weights"])}
            del gp, layers, packed
            torch.cuda.empty_cache()
    out["worst_vs_uniform"] = min(c["vs_uniform"] for c in out["cases"].values())
    out["all_on_the_packed_kernel"] = all(c["layers_on_the_packed_kernel"] == c["layers"] for c in out["cases"].values())
    return out


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


def large_batch_detail(dev, reps):
    """BASELINE config 4: 1x16g8 4096->4096 at batch 128.  Fused dequant-tile -> MFMA op (W never in HBM) next to the
    reference-equivalent pipeline (our dequant kernel + hipBLASLt GEMM through F.linear) and a dense fp16 GEMM."""
    import torch.nn.functional as F

    from aqlm_amd.inference_kernels import hip_kernel as hk

    fin = fout = 4096
    B = 128
    layers = [Layer(fin, fout, 1, 16, 8, 424242 + i, dev) for i in range(24)]  # rotate: 24 x 5.3 MB > L2
    x = torch.randn((B, fin), device=dev, dtype=torch.float16)

    def timeit(fn):
        for l in layers[:3]:
            fn(l)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 0
        for _ in range(max(2, reps // 2)):
            for l in layers:
                fn(l)
                n += 1
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    def timegraph(fn, xin):
        """The same rotation captured in one hipGraph (what a served prefill / speculative step looks like): kernel time
        without the interpreter.  The eager figures next to it are host-bound below ~20 us per call."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for l in layers[:3]:
                fn(l, xin)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for l in layers:
                    fn(l, xin)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = max(3, reps // 2)
            e0.record(s)
            for _ in range(n):
                g.replay()
            e1.record(s)
            e1.synchronize()
        del g
        return e0.elapsed_time(e1) * 1e3 / (n * len(layers))

    fused_eager = timeit(lambda l: hk.code1x16_matmat_dequant(x, l.codes, l.codebooks, l.scales, None))
    ref_like = timeit(lambda l: F.linear(x, hk.code1x16_dequant(l.codes, l.codebooks, l.scales)))
    W = hk.code1x16_dequant(layers[0].codes, layers[0].codebooks, layers[0].scales)
    Ws = [W] + [W.clone() for _ in range(7)]  # 8 x 32 MiB: the dense rotation does not sit in L2 either
    dense_eager = timeit(lambda l: F.linear(x, W))
    fused = timegraph(lambda l, xin: hk.code1x16_matmat_dequant(xin, l.codes, l.codebooks, l.scales, None), x)
    dense = timegraph(lambda l, xin: F.linear(xin, Ws[l.seed % 8]), x)
    ref_graph = timegraph(lambda l, xin: F.linear(xin, hk.code1x16_dequant(l.codes, l.codebooks, l.scales)), x)
    flop = 2.0 * B * fin * fout
    out = {"fused_mfma_us": fused, "fused_TFLOPs": flop / fused * 1e-6, "dequant_plus_gemm_us": ref_graph,
           "dense_fp16_gemm_us": dense, "fused_mfma_eager_us": fused_eager, "dequant_plus_gemm_eager_us": ref_like,
           "dense_fp16_gemm_eager_us": dense_eager,
           "note": "hipGraph replay of 24 rotating layers (kernel time, launch gaps included); *_eager_us: the same calls "
                   "issued one by one from python (host-bound)"}
    by_rows = {}
    for rows in (16, 32, 64):
        xr = torch.randn((rows, fin), device=dev, dtype=torch.float16)
        by_rows[f"rows{rows}"] = {"fused_mfma_us": timegraph(lambda l, xin: hk.code1x16_matmat_dequant(xin, l.codes, l.codebooks, l.scales, None), xr),
                                  "dense_fp16_gemm_us": timegraph(lambda l, xin: F.linear(xin, Ws[l.seed % 8]), xr)}
    out["graph_by_rows"] = by_rows
    # the 8-bit scheme's large-batch op (code2x8_matmat_dequant): fused dequant -> MFMA kernel with the codebooks in LDS (no gather
    # floor) vs the reference's pipeline (dequantise + library GEMM) vs dense fp16, same shape, same protocol
    keep = layers
    try:
        layers = [Layer(fin, fout, 2, 8, 8, 454545 + i, dev) for i in range(24)]
        kx = {}
        for rows in (16, 64, 128):
            xr = torch.randn((rows, fin), device=dev, dtype=torch.float16)
            f_us = timegraph(lambda l, xin: hk.code2x8_matmat_dequant(xin, l.codes, l.codebooks, l.scales, None), xr)
            hk.USE_FUSED_KX8_MFMA = False
            try:
                d_us = timegraph(lambda l, xin: hk.code2x8_matmat_dequant(xin, l.codes, l.codebooks, l.scales, None), xr)
            finally:
                hk.USE_FUSED_KX8_MFMA = True
            kx[f"rows{rows}"] = {"fused_mfma_us": f_us, "dequant_plus_gemm_us": d_us,
                                 "dense_fp16_gemm_us": timegraph(lambda l, xin: F.linear(xin, Ws[l.seed % 8]), xr),
                                 "fused_TFLOPs": 2.0 * rows * fin * fout / f_us * 1e-6}
        out["kx8_2x8g8_4096x4096"] = kx
    finally:
        layers = keep
    # 2..8 rows (speculative decode, small-batch serving; the module sends <= 6 rows to the matvec kernels): the prepacked matvec
    # (one more LDS read + 4 dots per entry and row) against the MFMA op (cost of 16 rows whatever the count), hipGraph, cold
    small = {}
    for (fi, fo) in ((4096, 4096), (4096, 11008)):
        ls = [Layer(fi, fo, 1, 16, 8, 434343 + i, dev, batch=8) for i in range(max(8, int(600e6 / algorithmic_bytes(fi, fo)) + 1))]
        keep, layers = layers, ls
        try:
            per = {}
            for rows in (2, 3, 4, 5, 6, 8):
                xr = torch.randn((rows, fi), device=dev, dtype=torch.float16)
                mv = timegraph(lambda l, xin: hk.code1x16_matmat_packed(xin, l.packed, l.codebooks, l.scales, None), xr) if ls[0].packed is not None else None
                mm = timegraph(lambda l, xin: hk.code1x16_matmat_dequant(xin, l.codes, l.codebooks, l.scales, None), xr)
                per[f"rows{rows}"] = {"prepacked_matvec_us": mv, "mfma_op_us": mm}
            small[f"{fi}->{fo}"] = per
        finally:
            layers = keep
        del ls
    out["small_batch_rows"] = small
    # why the op switches to dequant + library GEMM above FUSED_MFMA_MAX_ROWS: the fused kernel re-gathers per 128-row slab
    old = hk.FUSED_MFMA_MAX_ROWS
    try:
        for rows in (256, 1024):
            xr = torch.randn((rows, fin), device=dev, dtype=torch.float16)
            hk.FUSED_MFMA_MAX_ROWS = 1 << 30
            f_us = timeit(lambda l: hk.code1x16_matmat_dequant(xr, l.codes, l.codebooks, l.scales, None))
            hk.FUSED_MFMA_MAX_ROWS = 0
            d_us = timeit(lambda l: hk.code1x16_matmat_dequant(xr, l.codes, l.codebooks, l.scales, None))
            out[f"rows{rows}"] = {"fused_mfma_us": f_us, "dequant_plus_gemm_us": d_us,
                                  "op_default": "dequant_plus_gemm" if rows > old else "fused_mfma"}
    finally:
        hk.FUSED_MFMA_MAX_ROWS = old
    return out


class GraphedCalls:
    """A sequence of callables `fn(stream)` -- kernel launches through the C ABI and torch.distributed collectives alike -- captured
    into ONE hipGraph on a side stream (after an eager pass on that stream, which also initialises the communicator), timed by
    replays between HIP events, MAX over the ranks.  An eager RCCL call costs 20-30 us of host time and would swamp a 10-25 us
    kernel budget; a captured one is a graph node like the kernels around it.  If a collective cannot be captured the same calls
    are timed eagerly and `timing` says so."""

    def __init__(self, calls, dev):
        self.calls, self.dev = calls, dev
        self.stream = torch.cuda.Stream()
        self.timing = "hipgraph"
        with torch.cuda.stream(self.stream):
            for fn in calls:
                fn(self.stream)
        self.stream.synchronize()
        try:
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of torch.distributed queries events while this thread captures; in the default
            # ("global") mode such a call from another thread invalidates the capture
            with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                for fn in calls:
                    fn(torch.cuda.current_stream())
            self.graph.replay()
            self.stream.synchronize()
        except Exception as e:  # noqa: BLE001 - reported in the bench line
            self.graph, self.timing = None, f"eager ({type(e).__name__}: {str(e)[:120]})"
            torch.cuda.synchronize()

    def us_per_pass(self, reps, dist=None):
        import torch.distributed as td

        def run():
            if self.graph is not None:
                self.graph.replay()
            else:
                for fn in self.calls:
                    fn(self.stream)

        with torch.cuda.stream(self.stream):
            for _ in range(2):
                run()
        torch.cuda.synchronize()
        if dist is not None and td.is_initialized() and td.get_world_size() > 1:
            td.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self.stream):
            e0.record(self.stream)
            for _ in range(reps):
                run()
            e1.record(self.stream)
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        if dist is not None and td.is_initialized() and td.get_world_size() > 1:
            t = torch.tensor([us], device=self.dev, dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            us = float(t)
        return us


class ExtrasWatchdog:
    """The one JSON line must come out whatever happens after the timed region.  A daemon thread waits `budget_s`; if the main
    thread has not called finish() by then, rank 0 prints the result as it stands (with `extras_timed_out` naming the section that
    was running) and every rank leaves through os._exit -- a rank stuck in a collective cannot be joined."""

    def __init__(self, result, rank, budget_s):
        import threading

        self.result, self.rank, self.budget_s = result, rank, budget_s
        self.section = "detail"
        self.lock = threading.Lock()
        self.done = False
        self.fired = False
        if budget_s > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def _run(self):
        time.sleep(self.budget_s)
        with self.lock:
            if self.done:
                return
            self.fired = True
        if self.rank == 0:
            self.result["extras_timed_out"] = {"after_s": self.budget_s, "section": self.section}
            try:
                line = json.dumps(self.result, default=lambda o: None)
            except Exception:  # noqa: BLE001 - a dict mutated mid-dump: fall back to the headline fields
                line = json.dumps({k: v for k, v in self.result.items() if k not in ("detail", "sharded_70b")}, default=lambda o: None)
            sys.stdout.write(line + "\n")
            sys.stdout.flush()
        else:
            time.sleep(5.0)  # rank 0 prints first
        os._exit(0)

    def finish(self):
        with self.lock:
            if self.fired:
                time.sleep(3600)  # the watchdog thread is printing / exiting
                return False
            self.done = True
        return True


def _ensure_process_group(dev):
    """The sharded figures run the same code at every N: at N = 1 a single-rank "nccl" (= RCCL) group stands in, so that the
    collective's launch (captured in the graph) is part of the N = 1 point too."""
    import torch.distributed as dist

    if dist.is_initialized():
        return True
    try:
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        return True
    except Exception:  # noqa: BLE001 - the figures then come without a collective and say so
        return False


def sharded_70b(lib, dev, rank, world, steps):
    """North-star config 5: Llama-3-70B 8192->28672 1x16g8 layer, split along `in` over the ranks, partial outputs summed with an
    RCCL all-reduce (fp16, 56 KiB) -- and with the one-shot all-reduce over xGMI fused into the shard kernel's finalize.  The same
    schema at every N (round 5): kernel-only and end-to-end us per layer, aggregate GB/s, `rccl_ranks`; every figure is a hipGraph
    replay of [shard kernel, collective] x 16 distinct shards (`collective_timing`).  At N = 1 the shard is the 1/8 shard every rank
    of 8 would run and the collective runs over one rank (its launch cost, not its wire time)."""
    import ctypes

    import torch.distributed as dist

    from aqlm_amd import _native

    fin, fout = 8192, 28672
    parts = world if world > 1 else 8
    shard_in = fin // parts
    have_pg = _ensure_process_group(dev)
    ranks = dist.get_world_size() if have_pg else 1
    reps = max(4, steps // 2)
    layers = [Layer(shard_in, fout, 1, 16, 8, 1000 + rank * 100 + i, dev) for i in range(16)]
    gk = GraphedCalls([(lambda st, l=l: l.launch(lib, st.cuda_stream)) for l in layers], dev)
    kernel_us = gk.us_per_pass(reps, dist) / len(layers)
    full_bytes = algorithmic_bytes(fin, fout)
    out = {"layer": "8192->28672 1x16g8", "parts": parts, "rccl_ranks": ranks, "shard_in": shard_in,
           "kernel_us_per_shard": kernel_us, "shard_algorithmic_bytes": layers[0].bytes,
           "kernel_GBps_per_gpu": layers[0].bytes / kernel_us * 1e-3,
           "aggregate_GBps_kernel_only": parts * layers[0].bytes / kernel_us * 1e-3,
           "note": ("N = 1: per-shard figures of the 8-way split measured on one GPU (the collective runs over one rank: launch cost only); "
                    "the unsharded layer on one GPU is `unsharded_one_gpu`" if world == 1 else "in-split over the ranks, one collective per layer")}
    if world == 1:
        whole = [Layer(fin, fout, 1, 16, 8, 1500 + i, dev) for i in range(12)]
        gw = GraphedPass(whole, lib)
        us = gw.time_replays(reps) * 1e3 / gw.n
        out["unsharded_one_gpu"] = {"end_to_end_us": us, "aggregate_GBps_end_to_end": full_bytes / us * 1e-3,
                                    "collective": "none (the whole 8192->28672 layer in one launch)"}
        del gw, whole
    if have_pg:
        def with_rccl(l):
            def fn(st):
                l.launch(lib, st.cuda_stream)
                dist.all_reduce(l.y)
            return fn

        gr = GraphedCalls([with_rccl(l) for l in layers], dev)
        e2e_us = gr.us_per_pass(reps, dist) / len(layers)
        out.update({"end_to_end_us": e2e_us, "end_to_end_us_rccl": e2e_us, "allreduce_bytes": fout * 2, "collective_timing": gr.timing,
                    "aggregate_GBps_end_to_end": full_bytes / e2e_us * 1e-3, "aggregate_GBps_end_to_end_rccl": full_bytes / e2e_us * 1e-3,
                    "collective": "RCCL all-reduce (fp16, 56 KiB) behind the shard kernel, both in one hipGraph"})
        del gr
    # the MI355X-native variant: finalize fused with a one-shot all-reduce over xGMI (aqlm_amd/csrc/xgmi_reduce.hip).  Every rank
    # first agrees that it can run it (peer access to every other GPU of the node); any failure is reported, never fatal
    try:
        from aqlm_amd.xgmi import OneShotAllReduce

        can = all(r == torch.cuda.current_device() or torch.cuda.can_device_access_peer(torch.cuda.current_device(), r)
                  for r in range(torch.cuda.device_count())) and all(l.packed is not None and not l.packed.desc.variable_geometry for l in layers)
        flag = torch.tensor([1 if (can and have_pg) else 0], device=dev)
        if have_pg:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag):
            ar = OneShotAllReduce(fout, dev, spin_limit=1 << 19)  # ~0.25 s per wait at most: a lost peer must not stall the bench
            sc = layers[0].scales
            pub_own, flag_own = ar.own_pub_flag()

            def fused(l):  # two launches: the shard's matvec publishes its totals itself, then the reduce
                def fn(st):
                    rc = lib.aqlm_hip_gemv_1x16_packed_publish(ctypes.byref(l.packed.desc), l.packed.data_ptr(), l.codebooks.data_ptr(),
                                                               l.x.data_ptr(), 1, l.fin, _native.F16, ctypes.byref(ar.xg), pub_own, flag_own,
                                                               st.cuda_stream)
                    if rc:
                        _native.check(rc)
                    ar.reduce(sc, None, l.y, fout, 1, _native.F16, st.cuda_stream)
                return fn

            s = torch.cuda.current_stream()
            fused(layers[0])(s)
            torch.cuda.synchronize()
            bad = torch.tensor([1 if ar.timed_out() else 0], device=dev)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if int(bad):  # every rank leaves together (the collectives below must stay matched)
                raise RuntimeError("one-shot all-reduce: a peer's flag never arrived (IPC mapping over xGMI not working here)")
            y_native = layers[0].y.float().clone()
            layers[0].launch(lib, s.cuda_stream)
            y32 = layers[0].y.float()
            dist.all_reduce(y32)
            rel = float((y_native - y32).abs().mean() / y32.abs().mean())
            gx = GraphedCalls([fused(l) for l in layers], dev)
            x_us = gx.us_per_pass(reps, dist) / len(layers)
            out["xgmi_one_shot"] = {"end_to_end_us": x_us, "aggregate_GBps_end_to_end": full_bytes / x_us * 1e-3, "collective_timing": gx.timing,
                                    "mean_rel_vs_rccl_fp32_sum": rel, "timed_out": ar.timed_out(),
                                    "note": "shard kernel (publishes its fp32 totals) -> reduce over xGMI: 2 launches, fp32 on the wire, no RCCL launch"}
            if "end_to_end_us" not in out or x_us < out["end_to_end_us"]:  # the headline of the series is the better of the two collectives
                out.update({"end_to_end_us": x_us, "aggregate_GBps_end_to_end": full_bytes / x_us * 1e-3,
                            "collective": "one-shot all-reduce over xGMI fused into the shard kernel's finalize (RCCL figure: end_to_end_us_rccl)"})
            del gx
        else:
            out["xgmi_one_shot"] = {"skipped": "no peer access between all GPUs of the node, no process group, or a shard is not prepacked on the 16 x 16 geometry"}
    except Exception as e:  # noqa: BLE001 - diagnostics only
        out["xgmi_one_shot"] = {"error": f"{type(e).__name__}: {e}"}
    del gk
    out["mlp_plans"] = sharded_mlp_plans(lib, dev, rank, world, steps, have_pg)
    return out


def sharded_mlp_plans(lib, dev, rank, world, steps, have_pg=True):
    """The Llama-3-70B MLP (gate, up: 8192 -> 28672; down: 28672 -> 8192) under the two tensor-parallel plans of SURVEY.md 8(e), per rank:
      * in-split everywhere (north-star config 5 applied to every layer): gate / up shards 8192/N -> 28672, down 28672/N -> 8192,
        THREE all-reduces (28672, 28672, 8192 values);
      * Megatron pairing (aqlm_amd.sharded.shard_mlp): gate / up out-split 8192 -> 28672/N with NO collective -- both multiply the same
        x, so they run as ONE shared-input launch --, down in-split on the same cut, ONE all-reduce of 8192 values.
    Kernels and collectives of an MLP sit in one hipGraph (6 distinct MLPs per replay); N = 1 runs the shard shapes of N = 8 with
    single-rank collectives (their launch cost)."""
    import torch.distributed as dist

    parts = world if world > 1 else 8
    hid, inter = 8192, 28672
    i_sh = (inter // parts + 63) // 64 * 64  # the pairing cuts the inner dimension at whole 8-group code words
    plans = {"in_split_everywhere": [(hid // parts, inter), (hid // parts, inter), (inter // parts // 8 * 8, hid)],
             "paired": [(hid, i_sh), (hid, i_sh), (i_sh, hid)]}
    reduces = {"in_split_everywhere": [inter, inter, hid], "paired": [0, 0, hid]}
    reps = max(4, steps // 2)
    res = {"parts": parts, "rccl_ranks": dist.get_world_size() if have_pg and dist.is_initialized() else 1,
           "note": "per rank: the MLP's shard matvecs (prepacked kernel; the pairing's gate / up in one shared-input launch) and its fp16 "
                   "RCCL all-reduces captured in ONE hipGraph per rank; us per MLP"}
    for name, shapes in plans.items():
        sets = [[Layer(fi, fo, 1, 16, 8, 2000 + rank * 100 + 10 * k + i, dev) for k, (fi, fo) in enumerate(shapes)] for i in range(6)]
        units = []  # per MLP: the launchable units in order, with the all-reduce size behind each (0 = none)
        for st in sets:
            if name == "paired":
                gate, up, down = st
                up.x = gate.x  # one hidden state
                units.append([(FusedLayers([gate, up]), 0, None), (down, hid, down.y)])
            else:
                units.append([(l, n, l.y) for l, n in zip(st, reduces[name])])
        gk = GraphedCalls([(lambda s_, u=u: u.launch(lib, s_.cuda_stream)) for mlp in units for (u, _, _) in mlp], dev)
        k_us = gk.us_per_pass(reps, dist) / len(sets)
        entry = {"shard_shapes": [f"{fi}->{fo}" for fi, fo in shapes], "launches_per_mlp": len(units[0]), "kernels_us_per_mlp": k_us,
                 "collectives_per_mlp": sum(1 for n in reduces[name] if n), "allreduce_values": [n for n in reduces[name] if n]}
        del gk
        if have_pg:
            def with_coll(u, n, y):
                def fn(s_):
                    u.launch(lib, s_.cuda_stream)
                    if n:
                        dist.all_reduce(y)
                return fn

            ge = GraphedCalls([with_coll(u, n, y) for mlp in units for (u, n, y) in mlp], dev)
            entry["end_to_end_us_per_mlp"] = ge.us_per_pass(reps, dist) / len(sets)
            entry["collective_timing"] = ge.timing
            del ge
        res[name] = entry
        del units, sets
    return res


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


def launcher_command(gpus, argv, port=None):
    """The command `python bench.py --gpus N` re-executes itself as when it was not started by torch.distributed.run
    (the driver's own form: one rank per GPU of ONE node, rendezvous on 127.0.0.1)."""
    import socket

    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_probe(world, rank):
    """AQLM_BENCH_LAUNCH_PROBE=1: exercise only the launcher path (self-launch, rendezvous, one collective) on the gloo backend --
    the CPU test of `python bench.py --gpus N` (tests/test_tools.py); no GPU, no kernels."""
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launch_probe": True, "world": dist.get_world_size(), "sum_of_ranks_plus_1": float(t)}))
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-detail", action="store_true", help="skip the untimed per-shape / other-scheme breakdown")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-packed", action="store_true", help="direct L2-gather kernel for every layer (no prepacked path)")
    args = ap.parse_args()
    global PACK_MIN_OUT
    if args.no_packed:
        PACK_MIN_OUT = 0

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (one process per GPU over RCCL); the driver's
        # `python -m torch.distributed.run ... bench.py --gpus N` arrives with WORLD_SIZE set and skips this
        cmd = launcher_command(args.gpus, sys.argv[1:])
        sys.stdout.flush()
        os.execv(cmd[0], cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launched by torch.distributed.run with another --nproc-per-node?)")
    if os.environ.get("AQLM_BENCH_LAUNCH_PROBE") == "1":
        return launch_probe(world, rank)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    from aqlm_amd import _native  # raises if libaqlm_hip.so is missing

    lib = _native.lib

    # ---- the workload: 32 blocks x {4096->4096, 4096->11008}, all distinct
    NBLOCKS = 32
    layers = []
    for i in range(NBLOCKS):
        layers.append(Layer(4096, 4096, 1, 16, 8, rank * 10000 + 2 * i, dev))
        layers.append(Layer(4096, 11008, 1, 16, 8, rank * 10000 + 2 * i + 1, dev))
    step = GraphedPass(layers, lib)
    torch.cuda.synchronize()
    prepack = dict(PREPACK_STATS)  # the 64 layers of the timed workload only

    # ---- W warm-up steps, then EXACTLY K timed steps between barrier + synchronize on both sides
    with torch.cuda.stream(step.stream):
        for _ in range(args.warmup):
            step.graph.replay()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(step.stream):
        e0.record(step.stream)
        for _ in range(args.steps):
            step.graph.replay()
        e1.record(step.stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev_ms = e0.elapsed_time(e1)
    if dist:
        t = torch.tensor([wall, ev_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(t[0]), float(t[1])
    ms_per_step = wall * 1e3 / args.steps
    value = world * step.bytes / (ms_per_step * 1e-3) * 1e-9

    # ---- roofline of the dominant kernel (1x16 gemv): HIP events on the launch stream over the timed region
    launches = args.steps * step.n
    avg_launch_us = ev_ms * 1e3 / launches
    bytes_per_launch = step.bytes / step.n
    achieved = bytes_per_launch / avg_launch_us * 1e-3  # GB/s
    # HBM traffic needs the PMC counters, i.e. a rocprofv3 run of this very command: it cannot be measured from inside.
    # The value below is read from the committed summary of that run and labelled as such (null when absent).
    traffic, traffic_source = None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            pm = json.load(open(pmc_path))
            traffic = pm.get("gemv_1x16_hbm_bytes_per_launch")
            traffic_source = ("profiles/pmc_traffic.json: " + pm.get("how", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over bench.py"))
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": ("aqlm::gemv_1x16_packed_kernel<F16,1,3,65520,4> (prepacked codes, finalize inside the kernel; both shapes)"
                           if PACK_MIN_OUT else "aqlm::gemv_kernel<F16,1x16,g8,NB=1>"),
                "avg_launch_us": avg_launch_us, "algorithmic_bytes_per_launch": bytes_per_launch,
                "launches_timed": launches,
                # What bounds this metric on this chip for a one-launch-per-layer operator: (a) every
                # dependent launch costs the 1.45 us boundary of MI355X_MICROARCH.md -- a kernel that moved the algorithmic bytes
                # at 8 TB/s would reach 0.66 / (0.66 + 1.45) = 31 % on 4096->4096 and 52 % on 4096->11008, 43 % for the mix; (b)
                # both operands of a code's dot product are LDS gathers in the slice-bucketed formulation: 4.2 lane-gathers per
                # clock and CU / 2 per code = 32 % even on an infinitely large layer; (c) both at once: 22.5 % for the mix.
                "ceiling": {"launch_bound_frac": CEILING_LAUNCH, "lds_gather_bound_frac": CEILING_GATHER, "both_frac": CEILING_BOTH,
                            "boundary_us": 1.45, "what": "fraction of the 8 TB/s roofline a one-launch-per-layer 1x16 matvec can reach on the headline mix"},
                "frac_of_ceiling": achieved / HBM_PEAK_GBPS / min(CEILING_LAUNCH, CEILING_GATHER),
                "frac_of_ceiling_both": achieved / HBM_PEAK_GBPS / CEILING_BOTH,
                "note": "one launch = one matvec = one kernel (packed path, fused finalize); duration = HIP-event time of the "
                        "timed region / matvecs; achieved uses ALGORITHMIC bytes (2 B per code) even where the prepacked "
                        "path really reads ~4.5 B per code (32-bit entries + padding); rocprofv3 per-kernel durations are in profiles/"}

    result = {
        "metric": "QuantizedLinear 1x16g8 matvec algorithmic GB/s (bs=1, Llama-3-8B shapes 4096->4096/11008)",
        "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
        "data": "synthetic",
        "config": {"workload": "decode step = 32 blocks x {4096->4096, 4096->11008} 1x16g8 matvec, bs=1, 64 distinct "
                               "layers (own codes + codebook), 564 MB algorithmic bytes/step, hipGraph replay",
                   "scheme": "1x16g8", "batch": 1, "layers_per_step": step.n, "algorithmic_bytes_per_step": step.bytes,
                   "kernels": ("prepacked slice-bucketed gemv (layers of >= 0.5 M codes: both shapes); the direct L2-gather gemv "
                               "serves smaller layers and --no-packed" if PACK_MIN_OUT else "direct L2-gather gemv"),
                   "parallelism": f"replicas x{world}" if world > 1 else "single GPU",
                   # the load-time and memory price of the prepacked path for these 64 layers (outside the timed region)
                   "prepack_s_total": prepack["seconds"], "prepack_ms_per_layer": prepack["seconds"] * 1e3 / max(1, prepack["layers"]),
                   "packed_bytes": prepack["packed_bytes"], "canonical_code_bytes": prepack["canonical_code_bytes"],
                   "bits_per_weight_resident": {
                       "packed_only": 8.0 * prepack["packed_bytes"] / max(1, prepack["weights"]),
                       "packed_plus_canonical": 8.0 * (prepack["packed_bytes"] + prepack["canonical_code_bytes"]) / max(1, prepack["weights"]),
                       "canonical_only": 8.0 * prepack["canonical_code_bytes"] / max(1, prepack["weights"])} if prepack["layers"] else None},
        "tokens_per_s_this_stack": world * 1e3 / ms_per_step,
        "roofline": roofline,
    }

    # ---- outside the timed region: the step's outputs against the CPU ORACLE (oracle/aqlm_oracle.c, the restated reference
    # path -- checker only) on one layer of each shape, and against the generic HIP kernel (a different code path) as well ----
    from aqlm_amd.inference_kernels import hip_kernel as hk

    parity, parity_oracle = {}, {}
    for L in (layers[0], layers[1]):
        ref = hk.generic_matmat(L.x[:1], L.codes, L.codebooks, L.scales.reshape(-1, 1, 1, 1), None).float()
        got = L.y[:1].float()
        parity[f"{L.fin}x{L.fout}"] = float((got - ref).abs().mean() / ref.abs().mean())
        if rank == 0:
            from oracle import c_oracle

            k = c_oracle.DequantGemv(L.codebooks.float().cpu().numpy(), L.codes.cpu().numpy(), L.scales.float().cpu().numpy(), None, 16,
                                     nthreads=c_oracle.max_threads())
            y_or = torch.from_numpy(np.array(k(L.x[0].float().cpu().numpy()), copy=True))
            parity_oracle[f"{L.fin}x{L.fout}"] = float((got[0].cpu() - y_or).abs().mean() / y_or.abs().mean())
    result["parity_mean_rel_vs_generic_kernel"] = parity
    result["parity_mean_rel_vs_cpu_oracle"] = parity_oracle
    assert all(v < 1e-3 for v in parity.values()), f"bench outputs are off: {parity}"
    assert all(v < 1e-3 for v in parity_oracle.values()), f"bench outputs differ from the CPU oracle: {parity_oracle}"

    # ---- everything below is OUTSIDE the timed region and must never cost the headline line: a watchdog prints what there is and
    # ends the process if the extras (per-shape detail, the sharded figures with their captured collectives -- never run on more
    # than one GPU before the driver's 8-GPU tier --, the CPU and reference legs) are not done within their budget
    extras = ExtrasWatchdog(result, rank, float(os.environ.get("AQLM_BENCH_EXTRAS_TIMEOUT_S", "480")))

    # ---- untimed breakdown (rank 0 prints; every rank runs the collectives inside)
    if not args.no_detail:
        detail = {}
        result["detail"] = detail  # filled in place: a timed-out run still reports what it had
        reps = max(4, args.steps // 5)
        for name, idxs in (("1x16g8 4096->4096", range(0, 2 * NBLOCKS, 2)), ("1x16g8 4096->11008", range(1, 2 * NBLOCKS, 2))):
            sub = [layers[i] for i in idxs]
            # cold = rotate through 32 distinct instances plus the other shape's traffic in between is NOT present here,
            # so pad the rotation to > 512 MiB with extra instances of the same shape
            extra = [Layer(sub[0].fin, sub[0].fout, 1, 16, 8, 5000 + rank * 10000 + i, dev)
                     for i in range(max(0, int(600e6 / sub[0].bytes) + 1 - len(sub)))]
            gp = GraphedPass(sub + extra, lib)
            cold_us = gp.time_replays(reps) * 1e3 / gp.n
            gw = GraphedPass([sub[0]] * 32, lib)
            warm_us = gw.time_replays(reps) * 1e3 / gw.n
            detail[name] = {"cold_us": cold_us, "cold_GBps": sub[0].bytes / cold_us * 1e-3,
                            "cold_frac_of_8TBps": sub[0].bytes / cold_us * 1e-3 / HBM_PEAK_GBPS,
                            "warm_us": warm_us, "warm_GBps_cache_resident": sub[0].bytes / warm_us * 1e-3,
                            "instances": gp.n}
            del gp, gw, extra
        if PACK_MIN_OUT:
            detail["code_histograms"] = code_histograms_detail(lib, dev, rank, reps, NBLOCKS, value / world)
        # 2..8 input rows per launch on the prepacked path (the reference relaunches its matvec per row,
        # cuda_kernel.cpp:165-175): cold time and algorithmic GB/s per batch size at 4096->11008
        if PACK_MIN_OUT:
            nb_layers = [Layer(4096, 11008, 1, 16, 8, 6000 + rank * 10000 + i, dev, batch=8) for i in range(49)]
            rows = {}
            for B in (1, 2, 4, 8):
                gpb = GraphedPass(nb_layers, lib, batch=B)
                us = gpb.time_replays(reps) * 1e3 / gpb.n
                ab = nb_layers[0].alg_bytes(B)
                rows[f"B{B}"] = {"cold_us": us, "GBps": ab / us * 1e-3, "vs_B1": None}
                del gpb
            for B in (2, 4, 8):
                rows[f"B{B}"]["vs_B1"] = rows[f"B{B}"]["cold_us"] / rows["B1"]["cold_us"]
            rows["B1"]["vs_B1"] = 1.0
            detail["batch_rows_1x16g8_4096x11008_prepacked"] = rows
            del nb_layers
            # 16-element codebook vectors (1 bit per weight; the reference kernel's second template instance,
            # cuda_kernel.cu:476-521): prepacked (32 slices of 2048 x 32 B) vs the direct L2-gather kernel
            g16 = {}
            for fi, fo in ((4096, 4096), (4096, 11008)):
                ls = [Layer(fi, fo, 1, 16, 16, 6500 + rank * 10000 + i, dev) for i in range(int(600e6 / algorithmic_bytes(fi, fo, g=16)) + 1)]
                gpp = GraphedPass(ls, lib)
                us_p = gpp.time_replays(reps) * 1e3 / gpp.n
                del gpp
                for l in ls:
                    l.packed = None
                gpd = GraphedPass(ls, lib)
                us_d = gpd.time_replays(reps) * 1e3 / gpd.n
                g16[f"{fi}->{fo}"] = {"prepacked_cold_us": us_p, "direct_cold_us": us_d, "prepacked_GBps": ls[0].bytes / us_p * 1e-3,
                                      "prepacked_frac_of_8TBps": ls[0].bytes / us_p * 1e-3 / HBM_PEAK_GBPS}
                del gpd, ls
            detail["1x16g16_prepacked_vs_direct"] = g16
        # true Llama-3-8B decode token: 32 x [q,o 4096->4096; k,v 4096->1024; gate,up 4096->14336; down 14336->4096]
        shapes = [(4096, 4096), (4096, 1024), (4096, 1024), (4096, 4096), (4096, 14336), (4096, 14336), (14336, 4096)]
        tok = [Layer(fi, fo, 1, 16, 8, 7000 + rank * 10000 + 7 * b + j, dev) for b in range(32) for j, (fi, fo) in enumerate(shapes)]
        gp = GraphedPass(tok, lib)
        ms = gp.time_replays(reps)
        detail["llama3_8b_1x16g8_linear_stack"] = {"launches": gp.n, "ms_per_token": ms, "tokens_per_s": 1e3 / ms,
                                                   "algorithmic_GBps": gp.bytes / ms * 1e-6}
        # the same token with shared-input launches: [q,k,v] in one launch, o, [gate,up] in one launch, down
        fused = []
        for b in range(32):
            q, k, v, o, gate, up, down = tok[7 * b: 7 * b + 7]
            fused += [FusedLayers([q, k, v]), o, FusedLayers([gate, up]), down]
        gf = GraphedPass(fused, lib)
        msf = gf.time_replays(reps)
        detail["llama3_8b_1x16g8_linear_stack_shared_input_launches"] = {
            "launches": gf.n, "matvecs": gp.n, "ms_per_token": msf, "tokens_per_s": 1e3 / msf,
            "algorithmic_GBps": gf.bytes / msf * 1e-6, "speedup_vs_one_launch_per_layer": ms / msf}
        del gp, gf, fused, tok
        # Llama-3-70B on ONE MI355X (2-bit codes: 17.5 GB canonical, 39 GB prepacked): 80 x [q,o 8192->8192; k,v 8192->1024;
        # gate,up 8192->28672; down 28672->8192].  8 distinct blocks (5 GB of packed codes, far beyond every cache) replayed
        # 10 times inside one graph = the 80 blocks of a token.
        shapes70 = [(8192, 8192), (8192, 1024), (8192, 1024), (8192, 8192), (8192, 28672), (8192, 28672), (28672, 8192)]
        blk = [[Layer(fi, fo, 1, 16, 8, 7500 + rank * 10000 + 7 * b + j, dev) for j, (fi, fo) in enumerate(shapes70)] for b in range(8)]
        tok70 = [l for _ in range(10) for b in blk for l in b]
        gp = GraphedPass(tok70, lib)
        ms = gp.time_replays(max(2, reps // 2))
        fused70 = []
        for _ in range(10):
            for q, k, v, o, gate, up, down in blk:
                fused70 += [FusedLayers([q, k, v]), o, FusedLayers([gate, up]), down]
        gf = GraphedPass(fused70, lib)
        msf = gf.time_replays(max(2, reps // 2))
        detail["llama3_70b_1x16g8_linear_stack_one_gpu"] = {
            "launches": gp.n, "ms_per_token": ms, "tokens_per_s": 1e3 / ms, "algorithmic_GBps": gp.bytes / ms * 1e-6,
            "frac_of_8TBps": gp.bytes / ms * 1e-6 / HBM_PEAK_GBPS,
            "shared_input_launches": {"launches": gf.n, "ms_per_token": msf, "tokens_per_s": 1e3 / msf},
            "note": "8 distinct decoder blocks x 10 replays per token; every layer on the prepacked kernel"}
        del gp, gf, fused70, tok70, blk
        # q/k/v of a Llama-2-7B block (3 x 4096->4096): separate launches vs one launch, direct and prepacked
        keep_min, PACK_MIN_OUT = PACK_MIN_OUT, 0   # start from canonical codes only: the direct kernel
        qkv = [Layer(4096, 4096, 1, 16, 8, 8000 + rank * 10000 + i, dev) for i in range(3 * 40)]
        PACK_MIN_OUT = keep_min
        trio = {}
        gsep = GraphedPass(qkv, lib)
        trio["separate_direct_us"] = gsep.time_replays(reps) * 1e3 / 40
        gdir = GraphedPass([FusedLayers(qkv[i: i + 3], "direct") for i in range(0, len(qkv), 3)], lib)
        trio["one_launch_direct_us"] = gdir.time_replays(reps) * 1e3 / 40
        gpk = GraphedPass([FusedLayers(qkv[i: i + 3], "packed") for i in range(0, len(qkv), 3)], lib)
        trio["one_launch_prepacked_us"] = gpk.time_replays(reps) * 1e3 / 40
        gsp = GraphedPass(qkv, lib)  # members are prepacked now -> separate prepacked launches
        trio["separate_prepacked_us"] = gsp.time_replays(reps) * 1e3 / 40
        trio["algorithmic_bytes"] = 3 * qkv[0].bytes
        detail["qkv_3x_4096x4096_1x16g8"] = trio
        del gsep, gdir, gpk, gsp, qkv
        for sname, (K, nb, g) in {"2x8g8": (2, 8, 8), "8x8g32": (8, 8, 32)}.items():
            shapes7 = [(4096, 4096)] * 4 + [(4096, 11008)] * 2 + [(11008, 4096)]
            tok = [Layer(fi, fo, K, nb, g, 9000 + rank * 10000 + 7 * b + j, dev) for b in range(32) for j, (fi, fo) in enumerate(shapes7)]
            gp = GraphedPass(tok, lib)
            ms = gp.time_replays(reps)
            detail[f"llama2_7b_{sname}_linear_stack"] = {"launches": gp.n, "ms_per_token": ms, "tokens_per_s": 1e3 / ms,
                                                        "algorithmic_GBps": gp.bytes / ms * 1e-6,
                                                        "frac_of_8TBps": gp.bytes / ms * 1e-6 / HBM_PEAK_GBPS}
            if True:  # [q,k,v] and [gate,up] in one launch each (aqlm_hip_gemv_kx8_multi / aqlm_hip_gemv_8x8_lut_multi)
                fused = []
                for b in range(32):
                    q, k, v, o, gate, up, down = tok[7 * b: 7 * b + 7]
                    fused += [FusedLayers([q, k, v]), o, FusedLayers([gate, up]), down]
                gf = GraphedPass(fused, lib)
                msf = gf.time_replays(reps)
                detail[f"llama2_7b_{sname}_linear_stack_shared_input_launches"] = {
                    "launches": gf.n, "matvecs": gp.n, "ms_per_token": msf, "tokens_per_s": 1e3 / msf,
                    "algorithmic_GBps": gf.bytes / msf * 1e-6, "frac_of_8TBps": gf.bytes / msf * 1e-6 / HBM_PEAK_GBPS,
                    "speedup_vs_one_launch_per_layer": ms / msf}
                del gf, fused
            del gp, tok
        detail["bs128_1x16g8_4096x4096"] = large_batch_detail(dev, reps)
        # ---- roofline objects of BASELINE configs 3 and 4 (same fields as the top-level `roofline` of config 2).  Config 3:
        # the two schemes at the Llama-2-7B shapes 4096->4096 / 4096->11008, one launch per layer, cold (> 600 MB rotated);
        # traffic = HBM bytes per launch from the committed PMC passes of the kernels (profiles/, rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE over the microbenchmark of the same kernel; null when absent).  Config 4: dense MFMA peak.
        def pmc_traffic(fname):
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", fname)))
                c = pm["counters_mean_per_dispatch"]
                return 2.0 * c["FETCH_SIZE"] * 1024 + c.get("WRITE_SIZE", 0.0) * 1024  # KiB counters; reads x 2 (gfx950 correction)
            except Exception:
                return None

        for sname, (K, nb, g), pmc in (("2x8g8", (2, 8, 8), "r05_2x8_rep_kernel_pmc.json"), ("8x8g32", (8, 8, 32), "r05_8x8_lut_planar_kernel_pmc.json")):
            per, tot_b, tot_us = {}, 0.0, 0.0
            for fi, fo in ((4096, 4096), (4096, 11008)):
                ls = [Layer(fi, fo, K, nb, g, 9500 + rank * 10000 + i, dev) for i in range(int(600e6 / algorithmic_bytes(fi, fo, K, nb, g)) + 1)]
                gpx = GraphedPass(ls, lib)
                us = gpx.time_replays(reps) * 1e3 / gpx.n
                per[f"{fi}->{fo}"] = {"cold_us": us, "GBps": ls[0].bytes / us * 1e-3, "frac_of_8TBps": ls[0].bytes / us * 1e-3 / HBM_PEAK_GBPS}
                tot_b += ls[0].bytes
                tot_us += us
                del gpx, ls
            ach = tot_b / tot_us * 1e-3
            detail[f"config3_{sname}"] = {"roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                                                       "traffic": pmc_traffic(pmc), "traffic_source": f"profiles/{pmc} (4096-row layers of the kernel's microbenchmark; per launch)"},
                                          "per_shape": per,
                                          "kernel": "gemv_kx8_rep_kernel (16-fold replicated codebooks in LDS)" if K == 2 else
                                                    "gemv_8x8_lut_kernel on planar codes (per-token look-up tables in LDS)"}
        # 2x8 g8 at 1..32 input rows: one row = the replicated-LDS matvec, 2+ rows = the X-resident fused MFMA kernel (round 5; in phases
        # where features x rows do not fit the LDS, and for 17..32 rows), against a dense fp16 GEMM on rotating weights
        rows2 = {}
        for fi, fo in ((4096, 4096), (4096, 11008), (11008, 4096)):
            ls = [Layer(fi, fo, 2, 8, 8, 9600 + rank * 10000 + i, dev, batch=32) for i in range(min(64, int(600e6 / algorithmic_bytes(fi, fo, 2, 8, 8)) + 1))]
            Ws = [torch.randn((fo, fi), device=dev, dtype=torch.float16) for _ in range(24)]
            per = {}
            for B in (1, 2, 4, 8, 16, 32):
                gpb = GraphedPass(ls, lib, batch=B)
                per[f"B{B}"] = {"us": gpb.time_replays(reps) * 1e3 / gpb.n}
                del gpb
                xb = ls[0].x[:B]
                gd = GraphedCalls([(lambda st, W=W: torch.nn.functional.linear(xb, W)) for W in Ws], dev)
                per[f"B{B}"]["dense_fp16_us"] = gd.us_per_pass(reps) / len(Ws)
                del gd
            for B in (2, 4, 8, 16, 32):
                per[f"B{B}"]["vs_B1"] = per[f"B{B}"]["us"] / per["B1"]["us"]
            rows2[f"{fi}->{fo}"] = per
            del ls, Ws
        detail["small_batch_rows_2x8g8"] = rows2
        # 8x8 g32 at 2..6 input rows (the module's gemv rule): the table kernel as ONE launch of rows x the single-row workgroups
        # (aqlm_hip_gemv_8x8_lut_batch, round 5) next to the plain LDS kernel that served 2+ rows before (VERDICT r04 missing #4),
        # and the fused dequant -> MFMA kernel (aqlm_hip_gemm_8x8_mfma) at 2..64 rows
        if PACK_MIN_OUT:
            rows8 = {}
            for fi, fo in ((4096, 4096), (4096, 11008)):
                ls = [Layer(fi, fo, 8, 8, 32, 9700 + rank * 10000 + i, dev, batch=64) for i in range(min(96, int(600e6 / algorithmic_bytes(fi, fo, 8, 8, 32)) + 1))]
                per = {}
                from aqlm_amd.inference_kernels import hip_kernel as hk8
                per["fused_mfma_from_rows"] = hk8.fused_8x8_min_rows(fo, fi)  # the operator's switch (cost model read off these numbers)
                for B in (1, 2, 3, 4, 6):
                    gpb = GraphedPass(ls, lib, batch=B)
                    us = gpb.time_replays(reps) * 1e3 / gpb.n
                    per[f"B{B}"] = {"lut_us": us, "vs_B1": None}
                    del gpb
                    if B > 1:
                        for l in ls:
                            l.lut_rows = False
                        gpo = GraphedPass(ls, lib, batch=B)
                        per[f"B{B}"]["plain_lds_kernel_us"] = gpo.time_replays(reps) * 1e3 / gpo.n
                        del gpo
                        for l in ls:
                            l.lut_rows = True
                for B in (2, 3, 4, 6, 16, 64):  # the fused dequant -> MFMA kernel: one cost up to 16 rows
                    for l in ls:
                        l.fused_8x8 = True
                    gpf = GraphedPass(ls, lib, batch=B)
                    per.setdefault(f"B{B}", {})["fused_mfma_us"] = gpf.time_replays(reps) * 1e3 / gpf.n
                    del gpf
                    for l in ls:
                        l.fused_8x8 = False
                for B in (1, 2, 3, 4, 6):
                    per[f"B{B}"]["vs_B1"] = per[f"B{B}"]["lut_us"] / per["B1"]["lut_us"]
                rows8[f"{fi}->{fo}"] = per
                del ls
            detail["small_batch_rows_8x8g32"] = rows8
        lb = detail["bs128_1x16g8_4096x4096"]
        detail["config4_bs128"] = {"roofline": {"bound": "mfma", "achieved": lb["fused_TFLOPs"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                "frac": lb["fused_TFLOPs"] / MFMA_PEAK_TFLOPS, "traffic": pmc_traffic("r03_gemm_glds_kernel_pmc.json")},
                                   "fused_mfma_us": lb["fused_mfma_us"], "dense_fp16_gemm_us": lb["dense_fp16_gemm_us"],
                                   "kernel": "gemm_1x16_glds_kernel + gemm_glds_finalize_kernel"}
        extras.section = "sharded_70b"
        result["sharded_70b"] = sharded_70b(lib, dev, rank, world, args.steps)

    if rank == 0:  # rank 0 at every N (the other ranks wait at the barrier below; outside every timed region)
        extras.section = "cpu_baseline"
        result["cpu_baseline"] = None if args.no_cpu else cpu_baseline()
        if not args.no_cpu:
            extras.section = "gpu_reference_baseline"
            result["gpu_reference_baseline"] = gpu_reference_baseline()

    if not extras.finish():
        return  # the watchdog has printed the line and is ending the process
    if rank == 0:
        print(json.dumps(result))
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    else:
        import torch.distributed as td

        if td.is_initialized():  # the single-rank group the sharded figures made for themselves at N = 1
            td.destroy_process_group()


if __name__ == "__main__":
    main()
