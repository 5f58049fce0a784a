#!/bin/bash
set +e
OUT=gpurun_out/r3c17
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench/mb
for w in 0 14 12; do
  if [ $w = 0 ]; then timeout 120 stdbuf -oL $MB multi > $OUT/multi_default.log 2>&1; else MB_WAVES=$w timeout 120 stdbuf -oL $MB multi > $OUT/multi_w$w.log 2>&1; fi
  echo "waves $w rc=$?"
done
for f in $OUT/multi_*.log; do echo "== $f"; grep "gate/up\|70B" $f | grep -v "^# packed\|workgroup per"; done
