#!/bin/bash
set +e
OUT=gpurun_out/r3ab5
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
AB=$PWD/tools/microbench/ab
timeout 150 python -m pytest tests/test_hip_parity.py -x -q --timeout=100 -k "pipelined or shared_input or multi or fusion" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
for pass in 1 2; do
  for c in new8 new9; do
    timeout 120 stdbuf -oL $AB/mb_$c multi > $OUT/mb_${c}_$pass.log 2>&1; echo "mb $c rc=$?"
    grep -v "^#" $OUT/mb_${c}_$pass.log | grep "launch\|DMA" | grep -v "workgroup per" | sed "s/^/$c pass $pass: /"
  done
done
grep -h "differ" $OUT/mb_new9_1.log
