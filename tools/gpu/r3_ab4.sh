#!/bin/bash
set +e
OUT=gpurun_out/r3ab4
mkdir -p $OUT
export TMPDIR=/tmp
AB=$PWD/tools/microbench/ab
for pass in 1 2 3; do
  for c in new4 new5; do
    timeout 200 $AB/mb_$c gemv quick 8x8g32LUT > $OUT/mb_${c}_$pass.log 2>&1
    grep " 1 default  \| 1 two" $OUT/mb_${c}_$pass.log | sed "s/^/$c pass $pass: /"
  done
done
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -k "lut or 8x8" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
