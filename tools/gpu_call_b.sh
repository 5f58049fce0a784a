#!/bin/bash
set +e
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
export MB_W67=1
for o in 4096 1024; do
timeout 900 tools/microbench/mb gemv full 1x16g8P $o > $OUT/mb_w67_$o.log 2>&1; echo "rc=$?"
grep " 1 default\| waves=\|repacked" $OUT/mb_w67_$o.log | grep -v "14336"
done
