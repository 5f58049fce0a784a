#!/bin/bash
# 14-wave packing of mid-size layers: tests, single-layer A/B, shared-input A/B (new9 = 16 waves for q >= 48, new10 = new rule)
set +e
OUT=gpurun_out/r3ab6
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
AB=$PWD/tools/microbench/ab
timeout 400 python -m pytest tests/test_hip_parity.py -x -q --timeout=200 -k "packed or prepack or pipelined or shared_input or multi or fusion or fast_lane" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
for pass in 1 2; do
  for c in new9 new10; do
    timeout 200 stdbuf -oL $AB/mb_$c gemv quick 1x16g8P > $OUT/gemv_${c}_$pass.log 2>&1; echo "gemv $c rc=$?"
    grep " 1 default  " $OUT/gemv_${c}_$pass.log | sed "s/^/$c pass $pass: /"
    timeout 120 stdbuf -oL $AB/mb_$c multi > $OUT/multi_${c}_$pass.log 2>&1; echo "multi $c rc=$?"
    grep "pipelined segments\|separate" $OUT/multi_${c}_$pass.log | sed "s/^/$c pass $pass: /"
  done
done
grep -h "differ\|MISMATCH" $OUT/multi_new10_1.log $OUT/gemv_new10_1.log | grep -v "mean-rel [0-9.e-]*  *worst-abs [0-9.e-]*$"
