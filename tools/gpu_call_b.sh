#!/bin/bash
# A/B: library built with / without kernel-argument preload (tools/microbench/alt/libaqlm_hip.so = PRELOAD=0 build)
set +e
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for alt in 0 1; do
  if [ $alt = 1 ]; then export LD_LIBRARY_PATH=$PWD/tools/microbench/alt; else unset LD_LIBRARY_PATH; fi
  timeout 900 tools/microbench/mb gemv quick > $OUT/mb_quick_all_alt$alt.log 2>&1; echo "alt=$alt (1 = no preload) rc=$?"
  grep "default" $OUT/mb_quick_all_alt$alt.log | grep -v "1x16g8P"
done
done
