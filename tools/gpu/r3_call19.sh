#!/bin/bash
set +e
OUT=gpurun_out/r3c19
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
AQLM_TEST_PIPE_CASES=150 timeout 300 python -m pytest tests/test_hip_parity.py -x -q --timeout=250 -k "pipelined or g16 or shared_input or multi or compiled_group" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout 150 stdbuf -oL tools/microbench/mb multi > $OUT/multi.log 2>&1; echo "mb rc=$?"
grep "pipelined\|differ\|separate" $OUT/multi.log | grep -v "^# packed"
