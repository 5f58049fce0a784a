#!/bin/bash
set +e
OUT=gpurun_out/call3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -k "packed or prepack" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
timeout 600 tools/microbench/mb gemv quick > $OUT/mb_gemv_quick.log 2>&1
tail -30 $OUT/pytest_gpu.log
head -40 $OUT/mb_gemv_quick.log
