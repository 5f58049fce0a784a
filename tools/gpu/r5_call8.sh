#!/bin/bash
# Round 5, development call 8: X-resident K x 8 kernel with the strength-reduced X image fill -- tests, then its benchmark.
set +e
TAG=${1:-r5c8}
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 -k "x_resident or kx8 or fused_8x8 or sweep" > $OUT/pytest_sel.log 2>&1; echo "pytest(sel) rc=$?" | tee -a $OUT/pytest_sel.log
tail -5 $OUT/pytest_sel.log
timeout 900 python tools/gemm_kx8_xres_benchmark.py > $OUT/gemm_kx8_xres.log 2> $OUT/gemm_kx8_xres.err; echo "xres bench rc=$?"; cat $OUT/gemm_kx8_xres.log
