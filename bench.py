#!/usr/bin/env python3
"""bench.py -- the driver's measurement contract for the AQLM QuantizedLinear matvec path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "1x16g8 matvec, Llama-3-8B linear shapes (4096->4096/11008), bs=1"): one STEP = one decode
token's pass, batch 1, through 32 blocks, each = one 4096->4096 and one 4096->11008 1x16g8 QuantizedLinear matvec.  All 64 layers
are distinct instances (own codes AND own codebook), 564 MB of algorithmic bytes per step, so every step streams its weights from
HBM (larger than the 256 MiB Infinity Cache).  The step is captured once in a hipGraph and replayed.

value = algorithmic GB/s of the whole job = ranks x bytes-per-step / step time, inputs resident in HBM.  N > 1 = N independent
replicas of the step (one process per GPU, no data-path collective), "scaling": "weak"; the north star's row-sharded 70B layer +
all-reduce is an extra ("sharded_70b"), never the headline value.

Output protocol (benchlib/emit.py): the process prints ONE line to stdout -- the JSON result, written by rank 0 as the last
thing it ever writes there (fd 1 is pointed at stderr for everything else, C libraries and exit handlers included).  The same
document is written to bench_result.json.  This file is the headline path; the untimed extras live in benchlib/ (`detail`:
benchlib/detail.py, budgeted so the default run stays near a minute -- `--full-detail` runs every section; `sharded_70b`:
benchlib/sharded.py; `cpu_baseline` / `gpu_reference_baseline`: benchlib/cpu.py, the only importers of oracle/).
"""
import argparse
import json
import os
import sys
import time

T_PROCESS = time.perf_counter()
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.emit import ExtrasWatchdog, StdoutGuard, emit_final  # noqa: E402  (no torch: the guard goes in first)


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
    """AQLM_BENCH_LAUNCH_PROBE=1: exercise only the launcher path (self-launch, rendezvous, one collective, the emission protocol) on
    the gloo backend -- the CPU test of `python bench.py --gpus N` (tests/test_tools.py); no GPU, no kernels."""
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)
    print(f"rank {rank}: a stray print after the guard is installed goes to stderr")
    dist.barrier()
    dist.destroy_process_group()
    emit_final({"launch_probe": True, "world": world, "sum_of_ranks_plus_1": float(t)}, rank)


def emit_probe():
    """AQLM_BENCH_EMIT_PROBE=1 (tests/test_tools.py): the emission path with everything that broke round 5 in the way -- a C-level
    printf pending in the stdio buffer of fd 1 (what librccl's banner does), Python prints, an atexit handler that prints -- and a
    result carrying `roofline` and `cpu_baseline`.  stdout must be exactly the one JSON line."""
    import atexit
    import ctypes

    libc = ctypes.CDLL(None)
    libc.printf(b"RCCL version : pretend banner, pending in the C buffer without a flush\n")
    print("python chatter before the line")
    atexit.register(lambda: print("atexit chatter after the line"))
    emit_final({"metric": "probe", "value": 1.0, "roofline": {"frac": 0.5}, "cpu_baseline": {"value": 2.0}, "detail": {"x": [1, 2]}},
               0, os.environ.get("AQLM_BENCH_RESULT_FILE"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-detail", action="store_true", help="skip the untimed per-shape / other-scheme breakdown and the sharded figures")
    ap.add_argument("--full-detail", action="store_true", help="run every section of the breakdown (minutes) instead of the time-budgeted default")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline and gpu_reference_baseline legs")
    ap.add_argument("--no-packed", action="store_true", help="direct L2-gather kernel for every layer (no prepacked path)")
    ap.add_argument("--soft-exit", action="store_true", help="exit through the interpreter after the line (profilers write in exit handlers)")
    args = ap.parse_args()

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
    if args.soft_exit:
        os.environ["AQLM_BENCH_SOFT_EXIT"] = "1"
    StdoutGuard.install()  # from here on fd 1 is stderr for everybody; the result line goes out through emit_final
    os.environ.setdefault("NCCL_DEBUG", "WARN")  # (the librccl banner is harmless now; this only keeps stderr short)
    if os.environ.get("AQLM_BENCH_EMIT_PROBE") == "1":
        return emit_probe()
    if os.environ.get("AQLM_BENCH_LAUNCH_PROBE") == "1":
        return launch_probe(world, rank)

    import numpy as np
    import torch

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
    from benchlib import layers as LY
    from benchlib.layers import CEILING_BOTH, CEILING_GATHER, CEILING_LAUNCH, HBM_PEAK_GBPS, GraphedPass, Layer

    lib = _native.lib
    if args.no_packed:
        LY.PACK_MIN_OUT = 0

    # ---- the workload: 32 blocks x {4096->4096, 4096->11008}, all distinct
    NBLOCKS = 32
    layers = []
    for i in range(NBLOCKS):
        layers.append(Layer(4096, 4096, 1, 16, 8, rank * 10000 + 2 * i, dev))
        layers.append(Layer(4096, 11008, 1, 16, 8, rank * 10000 + 2 * i + 1, dev))
    step = GraphedPass(layers, lib)
    torch.cuda.synchronize()
    prepack = dict(LY.PREPACK_STATS)  # the 64 layers of the timed workload only

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

    # ---- roofline of the dominant kernel (1x16 gemv): HIP events on the launch stream over the timed region.  HBM traffic needs
    # the PMC counters, i.e. a rocprofv3 run of this very command: it cannot be measured from inside; the value is read from the
    # committed summary of that run (tools/gpu/final_evidence.sh -> profiles/pmc_traffic.json) and labelled as such
    launches = args.steps * step.n
    avg_launch_us = ev_ms * 1e3 / launches
    bytes_per_launch = step.bytes / step.n
    achieved = bytes_per_launch / avg_launch_us * 1e-3  # GB/s
    traffic, traffic_source = None, None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        traffic = pm.get("gemv_1x16_hbm_bytes_per_launch")
        traffic_source = "profiles/pmc_traffic.json: " + pm.get("how", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over bench.py")
    except Exception:  # noqa: BLE001
        pass
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_source": traffic_source,
                "kernel": ("aqlm::gemv_1x16_packed_kernel<F16,1,3,65520,4> (prepacked codes, finalize inside the kernel; both shapes)"
                           if LY.PACK_MIN_OUT else "aqlm::gemv_kernel<F16,1x16,g8,NB=1>"),
                "avg_launch_us": avg_launch_us, "algorithmic_bytes_per_launch": bytes_per_launch, "launches_timed": launches,
                # what bounds this metric for a one-launch-per-layer operator: the 1.45 us launch boundary
                # (43 % for the mix), two LDS gathers per code (32 %), both at once (22.5 %)
                "ceiling": {"launch_bound_frac": CEILING_LAUNCH, "lds_gather_bound_frac": CEILING_GATHER, "both_frac": CEILING_BOTH,
                            "boundary_us": 1.45, "what": "fraction of the 8 TB/s roofline a one-launch-per-layer 1x16 matvec can reach on the headline mix"},
                "frac_of_ceiling": achieved / HBM_PEAK_GBPS / min(CEILING_LAUNCH, CEILING_GATHER),
                "frac_of_ceiling_both": achieved / HBM_PEAK_GBPS / CEILING_BOTH,
                "note": "one launch = one matvec = one kernel; duration = HIP-event time of the timed region / matvecs; achieved uses "
                        "ALGORITHMIC bytes (2 B per code) even where the prepacked path reads ~4.8 B per code; rocprofv3 durations: profiles/"}
    bits = None
    if prepack["layers"]:
        w = max(1, prepack["weights"])
        bits = {"packed_only": 8.0 * prepack["packed_bytes"] / w, "canonical_only": 8.0 * prepack["canonical_code_bytes"] / w,
                "packed_plus_canonical": 8.0 * (prepack["packed_bytes"] + prepack["canonical_code_bytes"]) / w}
    side_file = os.environ.get("AQLM_BENCH_RESULT_FILE", os.path.join(ROOT, "bench_result.json")) if rank == 0 else None
    result = {
        "metric": "QuantizedLinear 1x16g8 matvec algorithmic GB/s (bs=1, Llama-3-8B shapes 4096->4096/11008)",
        "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "decode step = 32 blocks x {4096->4096, 4096->11008} 1x16g8 matvec, bs=1, 64 distinct layers (own codes + "
                               "codebook), 564 MB algorithmic bytes/step, hipGraph replay",
                   "scheme": "1x16g8", "batch": 1, "layers_per_step": step.n, "algorithmic_bytes_per_step": step.bytes,
                   "kernels": ("prepacked slice-bucketed gemv (layers of >= 0.5 M codes: both shapes)" if LY.PACK_MIN_OUT else "direct L2-gather gemv"),
                   "parallelism": f"replicas x{world}" if world > 1 else "single GPU",
                   # load-time and memory price of the prepacked path for these 64 layers (outside the timed region)
                   "prepack_s_total": prepack["seconds"], "prepack_ms_per_layer": prepack["seconds"] * 1e3 / max(1, prepack["layers"]),
                   "packed_bytes": prepack["packed_bytes"], "canonical_code_bytes": prepack["canonical_code_bytes"],
                   "bits_per_weight_resident": bits,
                   # share of the packed entry slots that hold no code (buckets padded to lane-steps of 4 entries), per shape
                   "packed_padding_fraction": {f"{L.fin}->{L.fout}": round(L.packed.padding_fraction(), 4) for L in layers[:2] if L.packed is not None}},
        "tokens_per_s_this_stack": world * 1e3 / ms_per_step,
        "roofline": roofline,
        "cpu_baseline": None,
    }

    # ---- outside the timed region: the step's outputs against the CPU ORACLE (oracle/aqlm_oracle.c, the restated reference path --
    # checker only, rank 0) on one layer of each shape, and against the generic HIP kernel (a different code path) on every rank
    from aqlm_amd.inference_kernels import hip_kernel as hk
    from benchlib import cpu as CPU

    parity, parity_oracle = {}, {}
    for L in (layers[0], layers[1]):
        ref = hk.generic_matmat(L.x[:1], L.codes, L.codebooks, L.scales.reshape(-1, 1, 1, 1), None).float()
        got = L.y[:1].float()
        parity[f"{L.fin}x{L.fout}"] = float((got - ref).abs().mean() / ref.abs().mean())
        if rank == 0:
            parity_oracle[f"{L.fin}x{L.fout}"] = CPU.oracle_parity(L, got)
    result["parity_mean_rel_vs_generic_kernel"] = parity
    result["parity_mean_rel_vs_cpu_oracle"] = parity_oracle
    assert all(v < 1e-3 for v in parity.values()), f"bench outputs are off: {parity}"
    assert all(v < 1e-3 for v in parity_oracle.values()), f"bench outputs differ from the CPU oracle: {parity_oracle}"

    # ---- everything below is OUTSIDE the timed region and must never cost the headline line: a watchdog emits what there is and
    # ends the process if the extras are not done within their budget
    extras = ExtrasWatchdog(result, rank, float(os.environ.get("AQLM_BENCH_EXTRAS_TIMEOUT_S", "480")), side_file)
    try:
        if rank == 0 and not args.no_cpu:  # the CPU leg first: the contract's `cpu_baseline` must not depend on the budgeted extras
            extras.section = "cpu_baseline"
            try:  # (guarded on its own: rank 0 must reach the collectives of `sharded_70b` like every other rank)
                result["cpu_baseline"] = CPU.cpu_baseline(float(os.environ.get("AQLM_BENCH_CPU_SAMPLE_S", "12")))
            except Exception as e:  # noqa: BLE001
                result["cpu_baseline_error"] = f"{type(e).__name__}: {e}"
        if not args.no_detail:
            from benchlib import detail as DT
            from benchlib import sharded as SH

            extras.section = "sharded_70b"
            try:
                result["sharded_70b"] = SH.sharded_70b(lib, dev, rank, world, args.steps)
            except Exception as e:  # noqa: BLE001 - an extra never costs the headline line
                result["sharded_70b"] = {"error": f"{type(e).__name__}: {e}"}
            result["detail"] = {}  # filled in place: a timed-out run still reports what it had
            budget = 0.0 if args.full_detail else float(os.environ.get("AQLM_BENCH_DETAIL_BUDGET_S", "42"))
            ctx = DT.Ctx(lib, dev, rank, world, max(4, args.steps // 5), layers, NBLOCKS, value)
            DT.run_detail(ctx, result["detail"], T_PROCESS, budget, args.full_detail, extras)
        if rank == 0 and not args.no_cpu:
            extras.section = "gpu_reference_baseline"
            result["gpu_reference_baseline"] = CPU.gpu_reference_baseline()
    except Exception as e:  # noqa: BLE001 - whatever an extra does (an import error included), the line goes out
        import traceback

        result["extras_error"] = {"section": extras.section, "error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()[-1500:]}
    result["bench_wall_s"] = round(time.perf_counter() - T_PROCESS, 1)

    if not extras.finish():
        return  # the watchdog is emitting the line and ending the process

    def teardown():  # ranks meet and the process groups go BEFORE the line (librccl may talk on the way out) -- but never cost it
        try:
            import torch.distributed as td

            if td.is_initialized():
                td.barrier()
                td.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass

    import threading

    th = threading.Thread(target=teardown, daemon=True)
    th.start()
    th.join(timeout=120.0 if dist else 20.0)
    emit_final(result, rank, side_file)


if __name__ == "__main__":
    main()
