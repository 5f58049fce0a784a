#!/bin/bash
# Round 5, development call 4: fused 8x8 g32 MFMA kernel -- parity tests, then its benchmark against the routes it replaces.
set +e
TAG=${1:-r5c4}
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x --timeout=300 -k "fused_8x8 or 8x8 or lut_rows" > $OUT/pytest_sel.log 2>&1; echo "pytest(sel) rc=$?" | tee -a $OUT/pytest_sel.log
tail -30 $OUT/pytest_sel.log
timeout 600 python tools/gemm_8x8_benchmark.py ${2:-} > $OUT/gemm_8x8_mfma.log 2> $OUT/gemm_8x8_mfma.err; echo "8x8 bench rc=$?"; cat $OUT/gemm_8x8_mfma.log; tail -5 $OUT/gemm_8x8_mfma.err
