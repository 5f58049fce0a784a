#!/bin/bash
set +e
OUT=gpurun_out/r3c9
mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench
timeout 300 $MB/mb multi > $OUT/mb_multi.log 2>&1; echo "mb multi rc=$?"; grep -v "^# packed\|^# check" $OUT/mb_multi.log
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "pipelined or shared_input or fused_" > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_sel.log
