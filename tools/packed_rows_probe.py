#!/usr/bin/env python3
"""The prepacked 1x16 matvec at B rows on one shape, a few launches: the command the counter passes of tools/gpu/gpu_pmc_cmd.sh wrap.
    python tools/packed_rows_probe.py <in> <out> <rows> [layers]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from aqlm_amd import _native  # noqa: E402
from benchlib.layers import Layer  # noqa: E402

fi, fo, rows = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 12
dev = torch.device("cuda:0")
ls = [Layer(fi, fo, 1, 16, 8, 77 + i, dev, batch=8) for i in range(n)]
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    for l in ls:
        l.launch(_native.lib, s, rows)
torch.cuda.synchronize()
print("done", fi, fo, rows)
