#!/bin/bash
# round 4, call 5: four-step butterfly reduction in the row walk: parity, mb, traces
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r4c5
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "lut or 8x8" --timeout=600 > $OUT/pytest_lut.log 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest_lut.log
MB=$PWD/tools/microbench/mb
timeout 300 $MB gemv quick 8x8g32LUT > $OUT/mb_8x8.log 2>&1; echo "mb rc=$?"; cat $OUT/mb_8x8.log
MBT=$PWD/tools/microbench/mb_trace
MB_LUT_WAVES=16 timeout 120 $MBT lut_trace 4096 4096 32 > $OUT/lut_trace_4096x4096_w16.log 2>&1; echo "== canonical waves 16"; tail -11 $OUT/lut_trace_4096x4096_w16.log
for W in 16 8; do
  MB_LUT_WAVES=$W timeout 120 $MBT lut_trace 4096 4096 32 planar > $OUT/lut_trace_planar_4096x4096_w$W.log 2>&1; echo "== planar waves $W"; tail -11 $OUT/lut_trace_planar_4096x4096_w$W.log
done
