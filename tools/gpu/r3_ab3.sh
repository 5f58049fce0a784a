#!/bin/bash
set +e
OUT=gpurun_out/r3ab3
mkdir -p $OUT
export TMPDIR=/tmp
AB=$PWD/tools/microbench/ab
for pass in 1 2 3; do
  for c in 6059df0 new2 new3; do
    timeout 200 $AB/mb_$c gemv quick 2x8g8 > $OUT/mb_${c}_$pass.log 2>&1
    grep " 1 default  \| 1 replicas" $OUT/mb_${c}_$pass.log | sed "s/^/$c pass $pass: /"
  done
done
