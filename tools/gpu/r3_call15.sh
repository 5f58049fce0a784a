#!/bin/bash
set +e
OUT=gpurun_out/r3c15
mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench
for o in 4096 11008 28672; do
  timeout 600 $MB/mb gemv full 1x16g8P $o > $OUT/mb_gemv_full_$o.log 2>&1; echo "mb gemv $o rc=$?"; grep -v "^# repacked\|^# check" $OUT/mb_gemv_full_$o.log | grep " 1 default\|entry=3B\|two-kernel\|^# packed" | head -30
done
