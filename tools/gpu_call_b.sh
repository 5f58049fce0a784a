#!/bin/bash
set +e
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
export MB_AB12=1
for o in 11008 14336 4096 8192; do
timeout 900 tools/microbench/mb gemv full 1x16g8P $o > $OUT/mb_ab12_$o.log 2>&1; echo "rc=$?"
grep " waves=" $OUT/mb_ab12_$o.log
done
