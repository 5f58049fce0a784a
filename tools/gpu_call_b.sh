#!/bin/bash
set +e
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "packed or prepack or fused or headline" 2>&1 | tail -2
for rep in 1 2; do
timeout 900 tools/microbench/mb gemv quick 1x16g8P > $OUT/mb_quick.log 2>&1; echo "rc=$?"
grep "default" $OUT/mb_quick.log
done
