#!/bin/bash
# local-search pass of the prepack (pk_improve_kernel): parity, conflict figures of the device layout, same-box A/B vs the greedy deal
set +e
OUT=gpurun_out/r3c20
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_hip_parity.py -x -q --timeout=300 -k "packed or pipelined or g16" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout 200 python tools/conflict_report.py > $OUT/conflicts.log 2>&1; echo "conflict rc=$?"; cat $OUT/conflicts.log
for o in 4096 11008 14336 28672; do
  MB_ARR=1 timeout 150 stdbuf -oL tools/microbench/mb gemv full 1x16g8P $o > $OUT/mb_$o.log 2>&1; echo "mb $o rc=$?"
  grep -v "^# check" $OUT/mb_$o.log | grep "default\|arrange\|repacked"
done
MB_ARR=1 timeout 100 stdbuf -oL tools/microbench/mb gemv full 1x16g16P 11008 > $OUT/mb_g16.log 2>&1; echo "mb g16 rc=$?"
grep -v "^# check" $OUT/mb_g16.log | grep "default\|arrange\|repacked"
