"""Untimed per-shape / per-scheme breakdown of bench.py (`detail` in the result line)."""
import json
import os
import time

import torch

from . import layers as LY
from .layers import (HBM_PEAK_GBPS, MFMA_PEAK_TFLOPS, PREPACK_STATS, ROOT, FusedLayers, GraphedCalls, GraphedPass, Layer,
                     algorithmic_bytes)


def code_histograms_detail(lib, dev, rank, reps, nblocks, uniform_value, full=True):
    """The headline step on codes that use the codebook unevenly (round 5): 32 x {4096->4096, 4096->11008}
    distinct layers per case, Zipf-distributed codes, labels shuffled and sorted by frequency.  Format v7 of the prepacked path
    balances the slices at pack time (relabelling; a variable row-group geometry where one entry outweighs a slice), so every
    case runs the packed kernel -- the reference's kernels are data-oblivious (cuda_kernel.cu:16-27), these figures say how
    close to that the slice-bucketed kernel stays."""
    out = {"protocol": "the timed step's layer list (one hipGraph, cold: 564 MB per step), codes ~ Zipf(alpha) over the 65536 entries",
           "uniform_GBps": uniform_value, "cases": {}}
    # the default (time-budgeted) pass takes two cases: the Zipf law that was worst in round 5 and the row-correlated layer
    # of VERDICT r05 weak #1 (rows of block b draw 90 % of their codes from labels [4096 b, 4096 (b + 1)): global usage uniform)
    laws = [(a, s_) for a in (0.5, 0.8, 1.0, 1.2) for s_ in (False, True)] if full else [(1.2, True)]
    laws.append(("rowblock", 0.9))
    for alpha, sorted_labels in laws:
        if True:
            before = dict(PREPACK_STATS)
            layers = []
            for i in range(nblocks):
                layers.append(Layer(4096, 4096, 1, 16, 8, 70000 + rank * 10000 + 2 * i, dev, code_law=(alpha, sorted_labels)))
                layers.append(Layer(4096, 11008, 1, 16, 8, 70000 + rank * 10000 + 2 * i + 1, dev, code_law=(alpha, sorted_labels)))
            gp = GraphedPass(layers, lib)
            ms = gp.time_replays(reps)
            packed = [l.packed for l in layers if l.packed is not None]
            gbps = gp.bytes / (ms * 1e-3) * 1e-9
            case = (f"rowblock_{sorted_labels}" if alpha == "rowblock" else f"zipf{alpha}_{'sorted' if sorted_labels else 'shuffled'}_labels")
            out["cases"][case] = {
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





def pmc_traffic(fname):
    """HBM bytes per launch from a committed PMC pass of the kernel's microbenchmark (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE; null
    when absent).  KiB counters; reads x 2 (MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide streaming read on gfx950)."""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", fname)))
        c = pm["counters_mean_per_dispatch"]
        return 2.0 * c["FETCH_SIZE"] * 1024 + c.get("WRITE_SIZE", 0.0) * 1024
    except Exception:  # noqa: BLE001
        return None


class Ctx:
    """What every section needs: the library, device, rank / world, repetitions, the timed step's layers and its value."""

    def __init__(self, lib, dev, rank, world, reps, layers, nblocks, value):
        self.lib, self.dev, self.rank, self.world, self.reps = lib, dev, rank, world, reps
        self.layers, self.nblocks, self.value = layers, nblocks, value


def sec_per_shape(c, detail):
    """cold (rotation > 512 MiB of distinct instances) and warm (cache-resident) time of the two headline shapes"""
    for name, idxs in (("1x16g8 4096->4096", range(0, 2 * c.nblocks, 2)), ("1x16g8 4096->11008", range(1, 2 * c.nblocks, 2))):
        sub = [c.layers[i] for i in idxs]
        extra = [Layer(sub[0].fin, sub[0].fout, 1, 16, 8, 5000 + c.rank * 10000 + i, c.dev)
                 for i in range(max(0, int(600e6 / sub[0].bytes) + 1 - len(sub)))]
        gp = GraphedPass(sub + extra, c.lib)
        cold_us = gp.time_replays(c.reps) * 1e3 / gp.n
        gw = GraphedPass([sub[0]] * 32, c.lib)
        warm_us = gw.time_replays(c.reps) * 1e3 / gw.n
        detail[name] = {"cold_us": cold_us, "cold_GBps": sub[0].bytes / cold_us * 1e-3,
                        "cold_frac_of_8TBps": sub[0].bytes / cold_us * 1e-3 / HBM_PEAK_GBPS,
                        "warm_us": warm_us, "warm_GBps_cache_resident": sub[0].bytes / warm_us * 1e-3, "instances": gp.n}
        del gp, gw, extra


def sec_code_histograms(c, detail, full):
    if LY.PACK_MIN_OUT:
        detail["code_histograms"] = code_histograms_detail(c.lib, c.dev, c.rank, c.reps, c.nblocks, c.value / c.world, full)


def sec_g16(c, detail):
    """16-element codebook vectors (1 bit per weight; the reference kernel's second template instance, cuda_kernel.cu:476-521):
    prepacked (32 slices of 2048 x 32 B) vs the direct L2-gather kernel"""
    if not LY.PACK_MIN_OUT:
        return
    g16 = {}
    for fi, fo in ((4096, 4096), (4096, 11008)):
        ls = [Layer(fi, fo, 1, 16, 16, 6500 + c.rank * 10000 + i, c.dev) for i in range(int(600e6 / algorithmic_bytes(fi, fo, g=16)) + 1)]
        gpp = GraphedPass(ls, c.lib)
        us_p = gpp.time_replays(c.reps) * 1e3 / gpp.n
        del gpp
        for l in ls:
            l.packed = None
        gpd = GraphedPass(ls, c.lib)
        us_d = gpd.time_replays(c.reps) * 1e3 / gpd.n
        g16[f"{fi}->{fo}"] = {"prepacked_cold_us": us_p, "direct_cold_us": us_d, "prepacked_GBps": ls[0].bytes / us_p * 1e-3,
                              "prepacked_frac_of_8TBps": ls[0].bytes / us_p * 1e-3 / HBM_PEAK_GBPS}
        del gpd, ls
    detail["1x16g16_prepacked_vs_direct"] = g16


def _stack(c, tok, fuse):
    gp = GraphedPass(tok, c.lib)
    ms = gp.time_replays(c.reps)
    out = {"launches": gp.n, "ms_per_token": ms, "tokens_per_s": 1e3 / ms, "algorithmic_GBps": gp.bytes / ms * 1e-6,
           "frac_of_8TBps": gp.bytes / ms * 1e-6 / HBM_PEAK_GBPS}
    fused = []
    for b in range(len(tok) // 7):
        q, k, v, o, gate, up, down = tok[7 * b: 7 * b + 7]
        fused += [FusedLayers([q, k, v]), o, FusedLayers([gate, up]), down]
    gf = GraphedPass(fused, c.lib)
    msf = gf.time_replays(c.reps)
    shared = {"launches": gf.n, "matvecs": gp.n, "ms_per_token": msf, "tokens_per_s": 1e3 / msf, "algorithmic_GBps": gf.bytes / msf * 1e-6,
              "frac_of_8TBps": gf.bytes / msf * 1e-6 / HBM_PEAK_GBPS, "speedup_vs_one_launch_per_layer": ms / msf}
    del gp, gf, fused
    return out, shared


def sec_llama3_8b(c, detail):
    """true Llama-3-8B decode token: 32 x [q,o 4096->4096; k,v 4096->1024; gate,up 4096->14336; down 14336->4096], one launch per
    layer and with shared-input launches ([q,k,v] and [gate,up] in one launch each)"""
    shapes = [(4096, 4096), (4096, 1024), (4096, 1024), (4096, 4096), (4096, 14336), (4096, 14336), (14336, 4096)]
    tok = [Layer(fi, fo, 1, 16, 8, 7000 + c.rank * 10000 + 7 * b + j, c.dev) for b in range(32) for j, (fi, fo) in enumerate(shapes)]
    detail["llama3_8b_1x16g8_linear_stack"], detail["llama3_8b_1x16g8_linear_stack_shared_input_launches"] = _stack(c, tok, True)


def sec_llama3_70b(c, detail):
    """Llama-3-70B on ONE MI355X (2-bit codes: 17.5 GB canonical, 39 GB prepacked): 80 x [q,o 8192->8192; k,v 8192->1024; gate,up
    8192->28672; down 28672->8192].  8 distinct blocks (5 GB of packed codes, far beyond every cache) replayed 10 times inside one
    graph = the 80 blocks of a token."""
    shapes70 = [(8192, 8192), (8192, 1024), (8192, 1024), (8192, 8192), (8192, 28672), (8192, 28672), (28672, 8192)]
    blk = [[Layer(fi, fo, 1, 16, 8, 7500 + c.rank * 10000 + 7 * b + j, c.dev) for j, (fi, fo) in enumerate(shapes70)] for b in range(8)]
    tok70 = [l for _ in range(10) for b in blk for l in b]
    gp = GraphedPass(tok70, c.lib)
    ms = gp.time_replays(max(2, c.reps // 2))
    fused70 = []
    for _ in range(10):
        for q, k, v, o, gate, up, down in blk:
            fused70 += [FusedLayers([q, k, v]), o, FusedLayers([gate, up]), down]
    gf = GraphedPass(fused70, c.lib)
    msf = gf.time_replays(max(2, c.reps // 2))
    detail["llama3_70b_1x16g8_linear_stack_one_gpu"] = {
        "launches": gp.n, "ms_per_token": ms, "tokens_per_s": 1e3 / ms, "algorithmic_GBps": gp.bytes / ms * 1e-6,
        "frac_of_8TBps": gp.bytes / ms * 1e-6 / HBM_PEAK_GBPS,
        "shared_input_launches": {"launches": gf.n, "ms_per_token": msf, "tokens_per_s": 1e3 / msf},
        "note": "8 distinct decoder blocks x 10 replays per token; every layer on the prepacked kernel"}


def sec_qkv(c, detail):
    """q/k/v of a Llama-2-7B block (3 x 4096->4096): separate launches vs one launch, direct and prepacked"""
    keep_min, LY.PACK_MIN_OUT = LY.PACK_MIN_OUT, 0   # start from canonical codes only: the direct kernel
    try:
        qkv = [Layer(4096, 4096, 1, 16, 8, 8000 + c.rank * 10000 + i, c.dev) for i in range(3 * 40)]
    finally:
        LY.PACK_MIN_OUT = keep_min
    trio = {}
    gsep = GraphedPass(qkv, c.lib)
    trio["separate_direct_us"] = gsep.time_replays(c.reps) * 1e3 / 40
    gdir = GraphedPass([FusedLayers(qkv[i: i + 3], "direct") for i in range(0, len(qkv), 3)], c.lib)
    trio["one_launch_direct_us"] = gdir.time_replays(c.reps) * 1e3 / 40
    gpk = GraphedPass([FusedLayers(qkv[i: i + 3], "packed") for i in range(0, len(qkv), 3)], c.lib)
    trio["one_launch_prepacked_us"] = gpk.time_replays(c.reps) * 1e3 / 40
    gsp = GraphedPass(qkv, c.lib)  # members are prepacked now -> separate prepacked launches
    trio["separate_prepacked_us"] = gsp.time_replays(c.reps) * 1e3 / 40
    trio["algorithmic_bytes"] = 3 * qkv[0].bytes
    detail["qkv_3x_4096x4096_1x16g8"] = trio


def sec_llama2_7b(c, detail):
    """Llama-2-7B decode token in the 2x8 g8 and 8x8 g32 schemes (BASELINE config 3's schemes as whole stacks)"""
    for sname, (K, nb, g) in {"2x8g8": (2, 8, 8), "8x8g32": (8, 8, 32)}.items():
        shapes7 = [(4096, 4096)] * 4 + [(4096, 11008)] * 2 + [(11008, 4096)]
        tok = [Layer(fi, fo, K, nb, g, 9000 + c.rank * 10000 + 7 * b + j, c.dev) for b in range(32) for j, (fi, fo) in enumerate(shapes7)]
        detail[f"llama2_7b_{sname}_linear_stack"], detail[f"llama2_7b_{sname}_linear_stack_shared_input_launches"] = _stack(c, tok, True)
        del tok


def sec_config4(c, detail):
    """BASELINE config 4 (1x16g8 4096->4096, 128 rows) + the large-batch neighbours; roofline object against the dense MFMA peak"""
    lb = large_batch_detail(c.dev, c.reps)
    detail["bs128_1x16g8_4096x4096"] = lb
    detail["config4_bs128"] = {"roofline": {"bound": "mfma", "achieved": lb["fused_TFLOPs"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                            "frac": lb["fused_TFLOPs"] / MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(lb.get("pmc_file", "r03_gemm_glds_kernel_pmc.json")),
                                            "algorithmic_bytes": algorithmic_bytes(4096, 4096, batch=128)},
                               "fused_mfma_us": lb["fused_mfma_us"], "dense_fp16_gemm_us": lb["dense_fp16_gemm_us"],
                               "kernel": lb.get("kernel", "gemm_1x16_glds_kernel + gemm_glds_finalize_kernel")}


def sec_config3(c, detail):
    """BASELINE config 3: 2x8 g8 and 8x8 g32 at the Llama-2-7B shapes 4096->4096 / 4096->11008, one launch per layer, cold (> 600 MB
    rotated); roofline objects with the same fields as the top-level one (traffic from the committed PMC passes of the kernels)"""
    for sname, (K, nb, g), pmc in (("2x8g8", (2, 8, 8), "r05_2x8_rep_kernel_pmc.json"), ("8x8g32", (8, 8, 32), "r05_8x8_lut_planar_kernel_pmc.json")):
        per, tot_b, tot_us = {}, 0.0, 0.0
        for fi, fo in ((4096, 4096), (4096, 11008)):
            ls = [Layer(fi, fo, K, nb, g, 9500 + c.rank * 10000 + i, c.dev) for i in range(int(600e6 / algorithmic_bytes(fi, fo, K, nb, g)) + 1)]
            gpx = GraphedPass(ls, c.lib)
            us = gpx.time_replays(c.reps) * 1e3 / gpx.n
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


def _dense_us(c, xb, Ws):
    gd = GraphedCalls([(lambda st, W=W: torch.nn.functional.linear(xb, W)) for W in Ws], c.dev)
    us = gd.us_per_pass(c.reps) / len(Ws)
    del gd
    return us


def sec_rows_2x8(c, detail):
    """2x8 g8 at 1..32 input rows: one row = the replicated-LDS matvec, 2+ rows = the X-resident fused MFMA kernel (in phases where
    features x rows do not fit the LDS, and for 17..32 rows), against a dense fp16 GEMM on rotating weights"""
    rows2 = {}
    for fi, fo in ((4096, 4096), (4096, 11008), (11008, 4096)):
        ls = [Layer(fi, fo, 2, 8, 8, 9600 + c.rank * 10000 + i, c.dev, batch=32) for i in range(min(64, int(600e6 / algorithmic_bytes(fi, fo, 2, 8, 8)) + 1))]
        Ws = [torch.randn((fo, fi), device=c.dev, dtype=torch.float16) for _ in range(24)]
        per = {}
        for B in (1, 2, 4, 8, 16, 32):
            gpb = GraphedPass(ls, c.lib, batch=B)
            per[f"B{B}"] = {"us": gpb.time_replays(c.reps) * 1e3 / gpb.n, "dense_fp16_us": _dense_us(c, ls[0].x[:B], Ws)}
            del gpb
        for B in (2, 4, 8, 16, 32):
            per[f"B{B}"]["vs_B1"] = per[f"B{B}"]["us"] / per["B1"]["us"]
        rows2[f"{fi}->{fo}"] = per
        del ls, Ws
    detail["small_batch_rows_2x8g8"] = rows2


def sec_rows_8x8(c, detail):
    """8x8 g32 at 2..64 rows: the table kernel as ONE launch of rows x the single-row workgroups, the plain LDS kernel that served
    2+ rows before, the fused dequant -> MFMA kernel (aqlm_hip_gemm_8x8_mfma), a dense fp16 GEMM, and the operator's switch"""
    if not LY.PACK_MIN_OUT:
        return
    from aqlm_amd.inference_kernels import hip_kernel as hk8

    rows8 = {}
    for fi, fo in ((4096, 4096), (4096, 11008)):
        ls = [Layer(fi, fo, 8, 8, 32, 9700 + c.rank * 10000 + i, c.dev, batch=64) for i in range(min(96, int(600e6 / algorithmic_bytes(fi, fo, 8, 8, 32)) + 1))]
        Ws = [torch.randn((fo, fi), device=c.dev, dtype=torch.float16) for _ in range(24)]
        per = {"fused_mfma_from_rows": hk8.fused_8x8_min_rows(fo, fi)}  # the operator's switch (cost model read off these numbers)
        for B in (1, 2, 3, 4, 6):
            gpb = GraphedPass(ls, c.lib, batch=B)
            per[f"B{B}"] = {"lut_us": gpb.time_replays(c.reps) * 1e3 / gpb.n, "vs_B1": None}
            del gpb
            if B > 1:
                for l in ls:
                    l.lut_rows = False
                gpo = GraphedPass(ls, c.lib, batch=B)
                per[f"B{B}"]["plain_lds_kernel_us"] = gpo.time_replays(c.reps) * 1e3 / gpo.n
                del gpo
                for l in ls:
                    l.lut_rows = True
        for B in (2, 3, 4, 6, 8, 16, 64):  # the fused dequant -> MFMA kernel: one cost up to 16 rows
            for l in ls:
                l.fused_8x8 = True
            gpf = GraphedPass(ls, c.lib, batch=B)
            per.setdefault(f"B{B}", {})["fused_mfma_us"] = gpf.time_replays(c.reps) * 1e3 / gpf.n
            del gpf
            for l in ls:
                l.fused_8x8 = False
            if B >= 8:
                per[f"B{B}"]["dense_fp16_us"] = _dense_us(c, ls[0].x[:B], Ws)
        for B in (1, 2, 3, 4, 6):
            per[f"B{B}"]["vs_B1"] = per[f"B{B}"]["lut_us"] / per["B1"]["lut_us"]
        rows8[f"{fi}->{fo}"] = per
        del ls, Ws
    detail["small_batch_rows_8x8g32"] = rows8


def sec_rows_1x16(c, detail):
    """1x16 g8 at 1..32 input rows on both headline shapes: the prepacked matvec (1..8 rows: one more LDS read
    + 4 dots per entry and row), the large-batch op `code1x16_matmat_dequant` (L2-gather MFMA kernels: one cost up to 16 rows), the
    slice-scan MFMA kernel built in round 6 (`scan_us`; not on a default route) and a dense fp16 GEMM on rotating weights.
    hipGraph replay over > 600 MB of distinct layers; `operator_us` = what QuantizedLinear.forward runs for that count (gemv rule:
    <= 6 rows the matvec kernels, above the large-batch op)."""
    if not LY.PACK_MIN_OUT:
        return
    from aqlm_amd import inference as inf
    from aqlm_amd.inference_kernels import hip_kernel as hk

    out = {}
    for fi, fo in ((4096, 4096), (4096, 11008)):
        ls = [Layer(fi, fo, 1, 16, 8, 6000 + c.rank * 10000 + i, c.dev, batch=8) for i in range(int(600e6 / algorithmic_bytes(fi, fo)) + 1)]
        Ws = [torch.randn((fo, fi), device=c.dev, dtype=torch.float16) for _ in range(24)]
        per = {}
        for B in (1, 2, 4, 8, 16, 32):
            xb = torch.randn((B, fi), device=c.dev, dtype=torch.float16)
            e = {"dense_fp16_us": _dense_us(c, xb, Ws)}
            if B <= 8:
                gpb = GraphedPass(ls, c.lib, batch=B)
                e["prepacked_matvec_us"] = gpb.time_replays(c.reps) * 1e3 / gpb.n
                del gpb
            if B >= 2:
                g = GraphedCalls([(lambda st, l=l: hk.code1x16_matmat_dequant(xb, l.codes, l.codebooks, l.scales, None)) for l in ls], c.dev)
                e["mfma_op_us"] = g.us_per_pass(c.reps) / len(ls)
                del g
            if B in (8, 16):
                g = GraphedCalls([(lambda st, l=l: hk.code1x16_matmat_scan(xb, l.codes, l.codebooks, l.scales, None)) for l in ls], c.dev)
                e["scan_us"] = g.us_per_pass(c.reps) / len(ls)
                del g
            e["operator_us"] = e["prepacked_matvec_us"] if B <= inf.GEMV_MAX_ROWS else e["mfma_op_us"]
            e["operator_vs_dense"] = e["dense_fp16_us"] / e["operator_us"]
            per[f"B{B}"] = e
        out[f"{fi}->{fo}"] = per
        del ls, Ws
    detail["batch_rows_1x16g8"] = out


def sec_unpack(c, detail):
    """What dropping the canonical codes costs the calls that need them: `aqlm_hip_unpack_1x16` of one layer
    into a transient buffer (one pass over the packed bytes), hipGraph replay over distinct layers, next to the 16-row large-batch op
    that would follow it"""
    if not LY.PACK_MIN_OUT:
        return
    from aqlm_amd.inference_kernels import hip_kernel as hk

    out = {}
    for fi, fo in ((4096, 4096), (4096, 11008)):
        ls = [Layer(fi, fo, 1, 16, 8, 6100 + c.rank * 10000 + i, c.dev) for i in range(min(48, int(600e6 / algorithmic_bytes(fi, fo)) + 1))]
        g = GraphedCalls([(lambda st, l=l: hk.unpack_1x16(l.packed)) for l in ls], c.dev)
        us = g.us_per_pass(c.reps) / len(ls)
        del g
        xb = torch.randn((16, fi), device=c.dev, dtype=torch.float16)
        g = GraphedCalls([(lambda st, l=l: hk.code1x16_matmat_dequant(xb, l.codes, l.codebooks, l.scales, None)) for l in ls], c.dev)
        op = g.us_per_pass(c.reps) / len(ls)
        del g
        out[f"{fi}->{fo}"] = {"unpack_us": us, "packed_bytes": int(ls[0].packed.buf.numel() * ls[0].packed.buf.element_size()),
                              "large_batch_op_16_rows_us": op, "unpack_over_op": us / op}
        del ls
    detail["unpack_1x16_us"] = out


# (name, function, runs in the default (budgeted) pass?)  Order = priority under the default time budget.
SECTIONS = [
    ("per_shape", sec_per_shape),
    ("batch_rows_1x16", None),       # (forward references: resolved in run_detail)
    ("code_histograms", None),
    ("config4", sec_config4),
    ("unpack", sec_unpack),
    ("rows_8x8", sec_rows_8x8),
    ("config3", sec_config3),
    ("rows_2x8", sec_rows_2x8),
    ("llama3_8b", sec_llama3_8b),
    ("llama2_7b", sec_llama2_7b),
    ("qkv", sec_qkv),
    ("g16", sec_g16),
    ("llama3_70b", sec_llama3_70b),
]


def run_detail(c, detail, t_start, budget_s, full, watchdog=None):
    """Run the sections in priority order; a section is only STARTED while the process is younger than `budget_s` (0 = no limit),
    so the default `python bench.py` stays within about a minute of driver time and `--full-detail` measures everything.  Sections
    that did not run are listed in detail["skipped"]; seconds per section in detail["section_seconds"]."""
    detail["section_seconds"], detail["skipped"] = {}, []
    for name, fn in SECTIONS:
        if fn is None:
            fn = {"batch_rows_1x16": sec_rows_1x16, "code_histograms": lambda c_, d_: sec_code_histograms(c_, d_, full)}[name]
        if budget_s and time.perf_counter() - t_start > budget_s:
            detail["skipped"].append(name)
            continue
        if watchdog is not None:
            watchdog.section = f"detail.{name}"
        t0 = time.perf_counter()
        try:
            fn(c, detail)
        except Exception as e:  # noqa: BLE001 - an extra never costs the headline line
            detail[f"{name}_error"] = f"{type(e).__name__}: {e}"
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        detail["section_seconds"][name] = round(time.perf_counter() - t0, 2)
    if detail["skipped"]:
        detail["skipped_note"] = f"sections not started after {budget_s:.0f} s of process time (default budget); `python bench.py --full-detail` runs all"
