#!/bin/bash
set +e
OUT=gpurun_out/r3c10
mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench
for o in 4096 11008 1024 28672; do
  timeout 300 $MB/mb gemv quick 1x16g8P $o > $OUT/mb_gemv_$o.log 2>&1; echo "mb gemv $o rc=$?"; grep -v "^# repacked" $OUT/mb_gemv_$o.log | grep -v "check packed.*mean-rel [0-9.e-]*  *worst-abs [0-9.e-]*$" | head -60
done
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "packed or prepack or fast_lane or pipelined or shared_input or two_streams" > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_sel.log
