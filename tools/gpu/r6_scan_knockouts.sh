#!/bin/bash
# Where does the slice-scan kernel's time go?  Knock-out switches (gemm_debug; results are wrong) and prefetch depths, 8 rows.
OUT=gpurun_out/${1:-r6c}; mkdir -p $OUT
python - <<'PY' 2>&1 | tee $OUT/knockouts.log
import sys, torch
sys.path.insert(0, '.')
from aqlm_amd import _native
from aqlm_amd.inference_kernels import hip_kernel as hk
from benchlib.layers import GraphedCalls, Layer, algorithmic_bytes
from benchlib import layers as LY
LY.PACK_MIN_OUT = 0
dev = torch.device('cuda:0')
for fi, fo in ((4096, 4096), (4096, 11008)):
    n = int(600e6 / algorithmic_bytes(fi, fo)) + 1
    ls = [Layer(fi, fo, 1, 16, 8, 4242 + i, dev) for i in range(n)]
    for B in (8,):
        xb = torch.randn((B, fi), device=dev, dtype=torch.float16)
        def t(lst):
            g = GraphedCalls([(lambda st, l=l: hk.code1x16_matmat_scan(xb, l.codes, l.codebooks, l.scales, None)) for l in lst], dev)
            us = g.us_per_pass(10) / len(lst)
            del g
            return us
        for depth in (4,):
            _native.set_tuning('scan_prefetch', depth)
            print(f"{fi}->{fo} B{B} prefetch {depth}: cold {t(ls):.2f} us   warm (one layer) {t([ls[0]] * 32):.2f} us", flush=True)
        _native.set_tuning('scan_prefetch', 4)
        for dbg, name in ((4, 'no reduction'), (15, 'skeleton'), (16 + 15, 'skeleton, no finalize launch'), (32 + 15, 'skeleton, no fill'), (64 + 15, 'skeleton, no x loads'), (96 + 15, 'skeleton, no fill, no x loads'), (16 + 96 + 15, 'skeleton, no fill / x / finalize')):
            _native.set_tuning('gemm_debug', dbg)
            print(f"{fi}->{fo} B{B} dbg {dbg:2d} ({name}): cold {t(ls):.2f} us", flush=True)
        _native.set_tuning('gemm_debug', 0)
PY
