#!/bin/bash
set +e
OUT=gpurun_out/r3ab2
mkdir -p $OUT
export TMPDIR=/tmp
AB=$PWD/tools/microbench/ab
for pass in 1 2 3; do
  for o in 4096 11008; do
    MB_CHAIN=1 timeout 200 $AB/mb_new gemv quick 1x16g8P $o > $OUT/mb_new_${o}_$pass.log 2>&1
    grep -v "^#" $OUT/mb_new_${o}_$pass.log | grep -v "14336\|chain\|scheme" | sed "s/^/pass $pass: /"
    timeout 200 $AB/mb_0b9166b gemv quick 1x16g8P $o 2>&1 | grep " 1 default  " | grep -v 14336 | sed "s/^/pass $pass r02: /"
  done
done
