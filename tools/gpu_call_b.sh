#!/bin/bash
# A/B: default library vs tools/microbench/alt/libaqlm_hip.so (an alternative build)
set +e
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "packed or prepack or sharded or fus" > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
for rep in 1 2; do
for alt in 0 1; do
  if [ $alt = 1 ]; then export LD_LIBRARY_PATH=$PWD/tools/microbench/alt; else unset LD_LIBRARY_PATH; fi
  for o in 4096 11008; do
  timeout 900 tools/microbench/mb gemv full 1x16g8P $o > $OUT/mb_var_${o}_alt$alt.log 2>&1; echo "alt=$alt rc=$?"
  grep " default" $OUT/mb_var_${o}_alt$alt.log
  done
done
done
