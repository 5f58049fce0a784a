#!/bin/bash
set +e
OUT=gpurun_out/r3c17
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench/mb
for w in 0 8 10 12; do
  if [ $w = 0 ]; then timeout 120 stdbuf -oL $MB multi > $OUT/multi_default.log 2>&1; else MB_WAVES=$w timeout 120 stdbuf -oL $MB multi > $OUT/multi_w$w.log 2>&1; fi
  echo "waves $w rc=$?"
done
for f in $OUT/multi_*.log; do echo "== $f"; grep "3 x 4096\|8B q/k/v" $f | grep -v "^# \|workgroup per\|no DMA"; done
