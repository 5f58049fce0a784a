#!/bin/bash
# experiments: launch waves (mb variants) and read-batch size for 3-4 rows (alt library)
set +e
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
for alt in 0 1; do
  if [ $alt = 1 ]; then export LD_LIBRARY_PATH=$PWD/tools/microbench/alt; else unset LD_LIBRARY_PATH; fi
  for o in 4096 11008; do
  timeout 900 tools/microbench/mb gemv full 1x16g8P $o > $OUT/mb_var_${o}_alt$alt.log 2>&1; echo "alt=$alt (1 = EG4 build) rc=$?"
  grep " default\|launch waves" $OUT/mb_var_${o}_alt$alt.log
  done
done
