"""North-star config 5 (row-sharded 70B layer + all-reduce) and the two tensor-parallel plans of the 70B MLP: untimed extras of bench.py."""
import torch

from .layers import FusedLayers, GraphedCalls, GraphedPass, Layer, algorithmic_bytes


def _process_group():
    """(initialised?, ranks).  At N = 1 there is NO process group: a single-rank all-reduce is a no-op -- it measures neither launch
    cost nor wire time -- so the N = 1 point carries kernel figures only and says `collective: none at 1 rank`."""
    import torch.distributed as dist

    if dist.is_initialized() and dist.get_world_size() > 1:
        return True, dist.get_world_size()
    return False, 1


def sharded_70b(lib, dev, rank, world, steps):
    """North-star config 5: the Llama-3-70B 8192->28672 1x16g8 layer over the ranks, both partitions of SURVEY.md 8(e):
      * in-split (the north star's row sharding: codes [:, j0:j1], x slice): shard 8192/N -> 28672, partial outputs summed with an
        RCCL all-reduce (fp16, 56 KiB) -- or with the one-shot all-reduce over xGMI fused into the shard kernel's finalize;
      * out-split (`out_split`): shard 8192 -> 28672/N, every rank reads all of x, outputs all-gathered (7 KiB per rank) or left
        sharded for a following in-split layer (the Megatron pairing of `mlp_plans`).
    The same schema at every N: kernel-only and end-to-end us per layer, aggregate GB/s, `rccl_ranks`; every figure is a hipGraph
    replay of [shard kernel, collective] x 16 distinct shards (`collective_timing`).  At N = 1 the shards are those of an 8-way
    split, measured on one GPU, and there is no collective (`collective: "none at 1 rank"`, end_to_end_us == kernel_us)."""
    import ctypes

    import torch.distributed as dist

    from aqlm_amd import _native

    fin, fout = 8192, 28672
    parts = world if world > 1 else 8
    shard_in = fin // parts
    have_pg, ranks = _process_group()
    reps = max(4, steps // 2)
    layers = [Layer(shard_in, fout, 1, 16, 8, 1000 + rank * 100 + i, dev) for i in range(16)]
    gk = GraphedCalls([(lambda st, l=l: l.launch(lib, st.cuda_stream)) for l in layers], dev)
    kernel_us = gk.us_per_pass(reps, dist) / len(layers)
    full_bytes = algorithmic_bytes(fin, fout)
    pk = layers[0].packed
    out = {"layer": "8192->28672 1x16g8", "parts": parts, "rccl_ranks": ranks, "partition": "in-split (row-sharded codes, all-reduce)", "shard_in": shard_in,
           "kernel_us_per_shard": kernel_us, "shard_algorithmic_bytes": layers[0].bytes,
           "kernel_GBps_per_gpu": layers[0].bytes / kernel_us * 1e-3,
           "aggregate_GBps_kernel_only": parts * layers[0].bytes / kernel_us * 1e-3,
           "packed_padding_fraction": (pk.padding_fraction() if pk is not None and hasattr(pk, "padding_fraction") else None),
           "note": ("N = 1: per-shard figures of the 8-way split measured on one GPU, no collective; the unsharded layer on one GPU is "
                    "`unsharded_one_gpu`" if world == 1 else "in-split over the ranks, one collective per layer")}
    if world == 1:
        whole = [Layer(fin, fout, 1, 16, 8, 1500 + i, dev) for i in range(12)]
        gw = GraphedPass(whole, lib)
        us = gw.time_replays(reps) * 1e3 / gw.n
        out["unsharded_one_gpu"] = {"end_to_end_us": us, "aggregate_GBps_end_to_end": full_bytes / us * 1e-3,
                                    "collective": "none (the whole 8192->28672 layer in one launch)"}
        del gw, whole
        out.update({"end_to_end_us": kernel_us, "aggregate_GBps_end_to_end": full_bytes / kernel_us * 1e-3, "collective": "none at 1 rank",
                    "collective_timing": "none"})
    # ---- the other partition: out-split shard 8192 -> 28672 / parts, all of x on every rank, outputs all-gathered
    try:
        fo_sh = fout // parts
        ol = [Layer(fin, fo_sh, 1, 16, 8, 1700 + rank * 100 + i, dev) for i in range(16)]
        go = GraphedCalls([(lambda st, l=l: l.launch(lib, st.cuda_stream)) for l in ol], dev)
        o_us = go.us_per_pass(reps, dist) / len(ol)
        del go
        osp = {"shard_shape": f"{fin}->{fo_sh}", "kernel_us_per_shard": o_us, "kernel_GBps_per_gpu": ol[0].bytes / o_us * 1e-3,
               "aggregate_GBps_kernel_only": parts * ol[0].bytes / o_us * 1e-3, "allgather_bytes_per_rank": fo_sh * 2,
               "packed_padding_fraction": (ol[0].packed.padding_fraction() if ol[0].packed is not None and hasattr(ol[0].packed, "padding_fraction") else None)}
        if have_pg:
            gathered = [torch.empty((1, fout), device=dev, dtype=torch.float16) for _ in ol]

            def with_gather(l, g):
                def fn(st):
                    l.launch(lib, st.cuda_stream)
                    dist.all_gather_into_tensor(g.view(-1), l.y.view(-1))
                return fn

            gg = GraphedCalls([with_gather(l, g) for l, g in zip(ol, gathered)], dev)
            e_us = gg.us_per_pass(reps, dist) / len(ol)
            osp.update({"end_to_end_us": e_us, "aggregate_GBps_end_to_end": full_bytes / e_us * 1e-3, "collective_timing": gg.timing,
                        "collective": "RCCL all-gather (fp16, 7 KiB per rank) behind the shard kernel, both in one hipGraph"})
            del gg, gathered
        else:
            osp.update({"end_to_end_us": o_us, "collective": "none at 1 rank"})
        out["out_split"] = osp
        del ol
    except Exception as e:  # noqa: BLE001 - diagnostics only
        out["out_split"] = {"error": f"{type(e).__name__}: {e}"}
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
    # first agrees that it can run it (peer access to every other GPU of the node); any failure is reported, never fatal.  At one
    # rank it is the publish-matvec + reduce pair over the rank's own buffer: 2 launches, its launch cost (labelled as such)
    try:
        from aqlm_amd.xgmi import OneShotAllReduce

        can = all(r == torch.cuda.current_device() or torch.cuda.can_device_access_peer(torch.cuda.current_device(), r)
                  for r in range(torch.cuda.device_count())) and all(l.packed is not None and not l.packed.desc.variable_geometry for l in layers)
        flag = torch.tensor([1 if can else 0], device=dev)
        if have_pg:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag) and (have_pg or world == 1):
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
            if have_pg:
                dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if int(bad):  # every rank leaves together (the collectives below must stay matched)
                raise RuntimeError("one-shot all-reduce: a peer's flag never arrived (IPC mapping over xGMI not working here)")
            y_native = layers[0].y.float().clone()
            layers[0].launch(lib, s.cuda_stream)
            y32 = layers[0].y.float()
            if have_pg:
                dist.all_reduce(y32)
            rel = float((y_native - y32).abs().mean() / y32.abs().mean())
            gx = GraphedCalls([fused(l) for l in layers], dev)
            x_us = gx.us_per_pass(reps, dist) / len(layers)
            out["xgmi_one_shot"] = {"end_to_end_us": x_us, "aggregate_GBps_end_to_end": full_bytes / x_us * 1e-3, "collective_timing": gx.timing,
                                    "mean_rel_vs_rccl_fp32_sum": rel, "timed_out": ar.timed_out(), "ranks": ranks,
                                    "note": "shard kernel (publishes its fp32 totals) -> reduce over xGMI: 2 launches, fp32 on the wire, no RCCL launch"
                                            + ("; at 1 rank: the pair's launch cost over the rank's own buffer, no wire" if not have_pg else "")}
            if have_pg and x_us < out["end_to_end_us"]:  # the headline of the series is the better of the two collectives
                out.update({"end_to_end_us": x_us, "aggregate_GBps_end_to_end": full_bytes / x_us * 1e-3,
                            "collective": "one-shot all-reduce over xGMI fused into the shard kernel's finalize (RCCL figure: end_to_end_us_rccl)"})
            del gx
        else:
            out["xgmi_one_shot"] = {"skipped": "no peer access between all GPUs of the node, or a shard is not prepacked on the 16 x 16 geometry"}
    except Exception as e:  # noqa: BLE001 - diagnostics only
        out["xgmi_one_shot"] = {"error": f"{type(e).__name__}: {e}"}
    del gk
    # which partition a single layer should take (ShardedQuantizedLinear's cost line uses the same two terms): kernel time of the
    # shard + the collective that follows it
    if "end_to_end_us" in out.get("out_split", {}):
        out["preferred_partition"] = "out-split" if out["out_split"]["end_to_end_us"] < out["end_to_end_us"] else "in-split"
    out["mlp_plans"] = sharded_mlp_plans(lib, dev, rank, world, steps, have_pg)
    return out


def sharded_mlp_plans(lib, dev, rank, world, steps, have_pg=True):
    """The Llama-3-70B MLP (gate, up: 8192 -> 28672; down: 28672 -> 8192) under the two tensor-parallel plans of SURVEY.md 8(e), per rank:
      * in-split everywhere (north-star config 5 applied to every layer): gate / up shards 8192/N -> 28672, down 28672/N -> 8192,
        THREE all-reduces (28672, 28672, 8192 values);
      * Megatron pairing (aqlm_amd.sharded.shard_mlp): gate / up out-split 8192 -> 28672/N with NO collective -- both multiply the same
        x, so they run as ONE shared-input launch --, down in-split on the same cut, ONE all-reduce of 8192 values.
    Kernels and collectives of an MLP sit in one hipGraph (6 distinct MLPs per replay); N = 1 runs the shard shapes of N = 8 with
    no collective (kernel figures only)."""
    import torch.distributed as dist

    parts = world if world > 1 else 8
    hid, inter = 8192, 28672
    i_sh = (inter // parts + 63) // 64 * 64  # the pairing cuts the inner dimension at whole 8-group code words
    plans = {"in_split_everywhere": [(hid // parts, inter), (hid // parts, inter), (inter // parts // 8 * 8, hid)],
             "paired": [(hid, i_sh), (hid, i_sh), (i_sh, hid)]}
    reduces = {"in_split_everywhere": [inter, inter, hid], "paired": [0, 0, hid]}
    reps = max(4, steps // 2)
    res = {"parts": parts, "rccl_ranks": dist.get_world_size() if have_pg and dist.is_initialized() else 1,
           "collective": "RCCL all-reduce (fp16)" if have_pg else "none at 1 rank (kernel figures only)",
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

