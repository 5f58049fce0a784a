#!/bin/bash
# phase trace + tuning-knob variants of the v5 packed kernel on the headline shapes
set +e
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -q -x --timeout=600 -k "prepack or packed or drop_canonical" --durations=5 -s > $OUT/pytest_packed.log 2>&1; echo "pytest rc=$?"
grep "LDS cycles" $OUT/pytest_packed.log; tail -12 $OUT/pytest_packed.log
for shape in "4096 4096" "4096 11008"; do
  timeout 300 tools/microbench/mb_trace trace $shape > $OUT/trace_${shape// /_}.log 2>&1; echo "trace rc=$?"
  tail -8 $OUT/trace_${shape// /_}.log
done
timeout 900 tools/microbench/mb gemv quick 1x16g8P > $OUT/mb_quick.log 2>&1; echo "mb rc=$?"
grep -v "^# check.*worst-abs [0-9.e-]*$" $OUT/mb_quick.log
