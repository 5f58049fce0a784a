#!/bin/bash
set +e
OUT=gpurun_out/r3c18
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench/mb
for o in 14336 8192; do
  timeout 300 $MB gemv full 1x16g8P $o > $OUT/mb_full_$o.log 2>&1; echo "rc=$?"
  grep " 1 default  \|waves=1[2346]\|^# packed" $OUT/mb_full_$o.log | grep -v "^# repacked" | cut -c1-100
done
