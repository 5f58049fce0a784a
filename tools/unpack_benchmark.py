#!/usr/bin/env python3
"""aqlm_hip_unpack_1x16 per layer (what a call that needs the canonical codes of a module that dropped them pays), hipGraph replay over
distinct layers; checks the round trip bit for bit.    python tools/unpack_benchmark.py [shapes]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from aqlm_amd.inference_kernels import hip_kernel as hk  # noqa: E402
from benchlib.layers import GraphedCalls, Layer, algorithmic_bytes  # noqa: E402

dev = torch.device("cuda:0")
for shp in (sys.argv[1] if len(sys.argv) > 1 else "4096x4096,4096x11008,11008x4096,8192x28672").split(","):
    fi, fo = (int(v) for v in shp.split("x"))
    ls = [Layer(fi, fo, 1, 16, 8, 900 + i, dev) for i in range(max(4, min(32, int(400e6 / algorithmic_bytes(fi, fo)))))]
    for l in ls[:2]:
        assert torch.equal(hk.unpack_1x16(l.packed), l.codes), "unpack is not lossless"
    g = GraphedCalls([(lambda st, l=l: hk.unpack_1x16(l.packed)) for l in ls], dev)
    us = g.us_per_pass(10) / len(ls)
    pb = ls[0].packed.buf.numel()
    print(f"{fi}->{fo}: unpack {us:7.2f} us  ({pb / 1e6:.1f} MB packed -> {ls[0].codes.numel() * 2 / 1e6:.1f} MB of codes: {(pb + ls[0].codes.numel() * 2) / us * 1e-3:.0f} GB/s)", flush=True)
