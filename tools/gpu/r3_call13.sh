#!/bin/bash
set +e
OUT=gpurun_out/r3c13
mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench
for o in 11008 4096; do
timeout 600 $MB/mb gemv full 1x16g16P $o > $OUT/mb_g16_full_$o.log 2>&1; echo "rc=$?"; grep -v "^# check\|^# repacked" $OUT/mb_g16_full_$o.log | head -70
done
