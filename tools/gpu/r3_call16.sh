#!/bin/bash
# steady-state phase trace of the shipped packed kernel at the 70B shapes (trace build)
set +e
OUT=gpurun_out/r3c16
mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench
for s in "8192 28672" "1024 28672" "8192 8192"; do
  set -- $s
  timeout 200 $MB/mb_trace trace $1 $2 > $OUT/trace_$1x$2.log 2>&1; echo "trace $1 $2 rc=$?"
  grep -A12 "run 1 \|run 3 " $OUT/trace_$1x$2.log | head -40
done
