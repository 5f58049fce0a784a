#!/bin/bash
# K x 8 replicated kernel with the first round's code words and the epilogue's scale / bias requested before the LDS fill
set +e
OUT=gpurun_out/r3c12
mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "kx8 or 2x8 or 1x8 or multi or fusion or scheme" > $OUT/pytest_kx8.log 2>&1; echo "pytest kx8 rc=$?"; tail -6 $OUT/pytest_kx8.log
for sch in 2x8g8 1x8g8; do
  timeout 300 $MB/mb gemv full $sch > $OUT/mb_gemv_$sch.log 2>&1; echo "mb gemv $sch rc=$?"; cat $OUT/mb_gemv_$sch.log | head -40
done
