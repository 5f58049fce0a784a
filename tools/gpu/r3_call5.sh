#!/bin/bash
set +e
OUT=gpurun_out/r3c5
mkdir -p $OUT
MB=$PWD/tools/microbench
for o in 4096 11008; do
  timeout 300 $MB/mb gemv quick 1x16g8P $o > $OUT/mb_gemv_$o.log 2>&1; echo "mb gemv $o rc=$?"; grep -v "^# check\|^# repacked" $OUT/mb_gemv_$o.log
done
MB_PREFETCH=8 timeout 200 $MB/mb_trace trace 4096 4096 > $OUT/trace_pd8_4096x4096.log 2>&1
grep -A7 "run 1 \|run 6 \|run 7 " $OUT/trace_pd8_4096x4096.log | grep -v "by block\|by wave"
