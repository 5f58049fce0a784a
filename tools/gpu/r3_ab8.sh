#!/bin/bash
# pipelined shared-input kernel: entry ring depth 3 (shipped) vs 6 -- same-box A/B
set +e
OUT=gpurun_out/r3ab8
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
AB=$PWD/tools/microbench/ab
for pass in 1 2; do
  for c in pd3 pd6; do
    timeout 120 stdbuf -oL $AB/mb_$c multi > $OUT/multi_${c}_$pass.log 2>&1; echo "multi $c rc=$?"
    grep "pipelined segments\|differ" $OUT/multi_${c}_$pass.log | sed "s/^/$c pass $pass: /"
  done
done
