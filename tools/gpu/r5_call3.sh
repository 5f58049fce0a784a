#!/bin/bash
# Round 5, development call 3: X-resident K x 8 MFMA kernel (tests + A/B timing), the fixed tests, full suite.
set +e
TAG=${1:-r5c3}
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 -k "x_resident or kx8 or lut_rows or full_size" > $OUT/pytest_sel.log 2>&1; echo "pytest(sel) rc=$?" | tee -a $OUT/pytest_sel.log
tail -30 $OUT/pytest_sel.log
timeout 900 python tools/gemm_kx8_xres_benchmark.py > $OUT/gemm_kx8_xres.log 2> $OUT/gemm_kx8_xres.err; echo "xres bench rc=$?"; cat $OUT/gemm_kx8_xres.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -30
