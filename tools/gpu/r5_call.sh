#!/bin/bash
# One development call of round 5:  bash tools/gpu/r5_call.sh <tag> [pytest -k expression]
# GPU suite (or the selected tests first), then the skewed-histogram microbenchmark.  Outputs under gpurun_out/<tag>/.
set +e
TAG=${1:-r5}
SEL=${2:-}
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench/mb
if [ -n "$SEL" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 -k "$SEL" > $OUT/pytest_sel.log 2>&1; echo "pytest(sel) rc=$?" | tee -a $OUT/pytest_sel.log
  tail -25 $OUT/pytest_sel.log
fi
timeout 600 $MB skew 4096 > $OUT/mb_skew_4096.log 2>&1; echo "mb skew 4096 rc=$?"
timeout 600 $MB skew 11008 > $OUT/mb_skew_11008.log 2>&1; echo "mb skew 11008 rc=$?"
grep -v "^# check\|^# packed" $OUT/mb_skew_4096.log $OUT/mb_skew_11008.log | cut -c1-200
grep -c MISMATCH $OUT/mb_skew_4096.log $OUT/mb_skew_11008.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -30
