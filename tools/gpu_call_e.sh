#!/bin/bash
set +e
OUT=gpurun_out/r2e
mkdir -p $OUT
export TMPDIR=/tmp
for shape in "4096 4096" "4096 11008"; do
  timeout 300 tools/microbench/mb_trace trace $shape > $OUT/trace_${shape// /_}.log 2>&1; echo "trace rc=$?"
  grep -v "^# check" $OUT/trace_${shape// /_}.log | tail -24
done
