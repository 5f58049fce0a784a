"""Synthetic QuantizedLinear layers, C-ABI launches and hipGraph timing shared by bench.py and its untimed extras."""
import os
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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
        elif code_law[0] == "rowblock":
            # ("rowblock", p): the rows of block b (16 equal blocks of rows) draw a fraction p of their codes from the labels
            # [4096 b, 4096 (b + 1)) and the rest uniformly -- label use correlated with the ROW while every entry is used equally
            # often over the layer (no global histogram sees it)
            p = float(code_law[1])
            n = fout * (fin // g) * K
            uni = torch.randint(0, 2**nbits, (n,), generator=gen, device=device, dtype=torch.int32)
            blk = (torch.arange(fout, device=device, dtype=torch.int32) * 16 // fout).repeat_interleave((fin // g) * K)
            own = blk * 4096 + torch.randint(0, 4096, (n,), generator=gen, device=device, dtype=torch.int32)
            take = torch.rand((n,), generator=gen, device=device) < p
            unsigned = torch.where(take, own, uni).reshape(fout, fin // g, K)
            self.codes = (unsigned - (unsigned >= hi) * 2**nbits).to(cdt)
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
