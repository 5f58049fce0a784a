#!/bin/bash
set +e
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
for o in 11008 14336 8192 28672; do
timeout 900 tools/microbench/mb gemv full 1x16g8P $o > $OUT/mb_var_$o.log 2>&1; echo "rc=$?"
grep " 1 default\| waves=" $OUT/mb_var_$o.log
done
