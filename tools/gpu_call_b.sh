#!/bin/bash
set +e
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
export MB_PD=1
for o in 4096 11008; do
timeout 900 tools/microbench/mb gemv full 1x16g8P $o > $OUT/mb_pd_$o.log 2>&1; echo "rc=$?"
grep " 1 default\| prefetch=" $OUT/mb_pd_$o.log
done
