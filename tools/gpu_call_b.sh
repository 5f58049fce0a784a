#!/bin/bash
set +e
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|rror" $OUT/pytest_gpu.log | tail -8
timeout 900 tools/microbench/mb gemv full 8x8g32LUT > $OUT/mb_lut.log 2>&1; echo "rc=$?"
grep "default\|two-kernel" $OUT/mb_lut.log
