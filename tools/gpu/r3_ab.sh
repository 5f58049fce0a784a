#!/bin/bash
# A/B of the prepacked matvec across commits of the round on ONE box: tools/microbench/ab/{libaqlm_hip,mb}_<commit> (built by hand
# from git worktrees), two interleaved passes
set +e
OUT=gpurun_out/r3ab
mkdir -p $OUT
export TMPDIR=/tmp
AB=$PWD/tools/microbench/ab
for pass in 1 2 3; do
for c in ${AB_COMMITS:-0b9166b c79269f 9f0f72c new}; do
  for o in 4096 11008; do
    timeout 200 $AB/mb_$c gemv quick 1x16g8P $o > $OUT/mb_${c}_${o}_$pass.log 2>&1
    echo "$c pass $pass: $(grep ' 1 default  ' $OUT/mb_${c}_${o}_$pass.log | grep -v 14336 | head -1)"
  done
done
done
