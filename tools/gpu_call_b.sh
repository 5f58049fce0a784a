#!/bin/bash
set +e
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error|Error" $OUT/pytest_gpu.log | tail -8
for o in 4096 11008; do
  timeout 900 tools/microbench/mb gemv full 1x16g8P $o > $OUT/mb_var_${o}.log 2>&1; echo "rc=$?"
  grep " default\|two-kernel\|MISMATCH" $OUT/mb_var_${o}.log
done
