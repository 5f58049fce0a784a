#!/bin/bash
set +e
OUT=gpurun_out/r3c6
mkdir -p $OUT
MB=$PWD/tools/microbench
timeout 200 $MB/mb_trace trace 4096 4096 > $OUT/trace_4096x4096.log 2>&1
grep -A8 "run 1 \|run 6 " $OUT/trace_4096x4096.log | grep -v "by block\|by wave"
timeout 200 $MB/mb_trace trace 8192 28672 > $OUT/trace_8192x28672.log 2>&1
grep -A8 "run 1 " $OUT/trace_8192x28672.log | grep -v "by block\|by wave"
