#!/bin/bash
set +e
OUT=gpurun_out/r3c14
mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench
for o in 4096 11008 1024 28672; do
  timeout 300 $MB/mb gemv quick 1x16g8P $o > $OUT/mb_gemv_$o.log 2>&1; echo "mb gemv $o rc=$?"; grep -v "^# repacked\|^# check" $OUT/mb_gemv_$o.log | grep "default\|MISMATCH" | head -30
done
grep -h MISMATCH $OUT/*.log | head
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "packed or prepack or fast_lane or pipelined or shared_input or two_streams or xgmi or sharded" > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_sel.log
