#!/bin/bash
# round 4, call 2: phase trace of the look-up-table kernel, the reference Triton timing with the raw op's prepack engaged, copy / pickle test
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r4c2
rm -rf $OUT; mkdir -p $OUT
MBT=$PWD/tools/microbench/mb_trace
timeout 120 $MBT lut_trace 4096 4096 32 > $OUT/lut_trace_4096x4096.log 2>&1; echo "trace rc=$?"; tail -9 $OUT/lut_trace_4096x4096.log
timeout 120 $MBT lut_trace 4096 11008 32 > $OUT/lut_trace_4096x11008.log 2>&1; tail -9 $OUT/lut_trace_4096x11008.log
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "copy_and_pickle or raw_op" --timeout=300 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 600 python tools/reference_triton.py --out $OUT/reference_triton.json > $OUT/reference_triton.log 2>&1; echo "triton rc=$?"; grep -E "scheme|_us|speedup" $OUT/reference_triton.log
