#!/bin/bash
set +e
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 tools/microbench/mb rates > $OUT/mb_rates.log 2>&1; echo "rates rc=$?"
grep -v "^lds" $OUT/mb_rates.log
for o in 4096 11008; do
timeout 900 tools/microbench/mb gemv full 1x16g8P $o > $OUT/mb_var_$o.log 2>&1; echo "mb rc=$?"
grep -v "^# check" $OUT/mb_var_$o.log | grep -v "device\|empty-kernel\|^scheme\|waves=\|xcopies\|arrange\|repacked\|prefetch\|entry="
done
