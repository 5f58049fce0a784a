#!/bin/bash
set +e
OUT=gpurun_out/r3c7
mkdir -p $OUT
MB=$PWD/tools/microbench
for v in s6n4x0 s5n4x0 s5n5x0 s4n5x0 s6n3x0; do
 for o in 4096 11008 1024; do
  timeout 300 $MB/mb_$v gemv quick 1x16g8P $o > $OUT/mb_${v}_gemv_$o.log 2>&1; echo "== $v out $o rc=$?"; grep "^1x16g8P.*default \|^# packed\|MISMATCH" $OUT/mb_${v}_gemv_$o.log | grep -v "default again"
 done
done
