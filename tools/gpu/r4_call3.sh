#!/bin/bash
# round 4, call 3: look-up-table kernel, second cut (staged hand-in, early max|x|, 8 / 16 waves) + planar code layout: parity, mb, traces, counters
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r4c3
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "lut or 8x8 or kx8 or copy_and_pickle or raw_op" --timeout=600 > $OUT/pytest_lut.log 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest_lut.log
MB=$PWD/tools/microbench/mb
timeout 300 $MB gemv quick 8x8g32LUT > $OUT/mb_8x8.log 2>&1; echo "mb rc=$?"; cat $OUT/mb_8x8.log
MBT=$PWD/tools/microbench/mb_trace
for W in 16 8; do
  MB_LUT_WAVES=$W timeout 120 $MBT lut_trace 4096 4096 32 > $OUT/lut_trace_4096x4096_w$W.log 2>&1; echo "== canonical waves $W"; tail -9 $OUT/lut_trace_4096x4096_w$W.log
  MB_LUT_WAVES=$W timeout 120 $MBT lut_trace 4096 4096 32 planar > $OUT/lut_trace_planar_4096x4096_w$W.log 2>&1; echo "== planar waves $W"; tail -9 $OUT/lut_trace_planar_4096x4096_w$W.log
done
MB_LUT_WAVES=16 timeout 120 $MBT lut_trace 4096 11008 32 planar > $OUT/lut_trace_planar_4096x11008_w16.log 2>&1; tail -9 $OUT/lut_trace_planar_4096x11008_w16.log
bash tools/gpu/gpu_pmc.sh 8x8g32LUTP 4096 r4c3_lutp > $OUT/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r4c3_lutp gemv_8x8_lut_kernel $OUT/lut_planar_kernel_pmc.json | tail -12
find gpurun_out/pmc_r4c3_lutp -name "*.db" -delete; find gpurun_out/pmc_r4c3_lutp -name "*kernel_trace*" -delete
