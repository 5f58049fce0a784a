#!/bin/bash
# g16 prepacked path: parity tests, then the microbenchmark (packed g16 vs direct g16)
set +e
OUT=gpurun_out/r3c11
mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "g16" > $OUT/pytest_g16.log 2>&1; echo "pytest g16 rc=$?"; tail -15 $OUT/pytest_g16.log
timeout 300 $MB/mb gemv quick 1x16g16 > $OUT/mb_gemv_g16.log 2>&1; echo "mb gemv g16 rc=$?"; grep -v "^# repacked" $OUT/mb_gemv_g16.log | head -80
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -k "packed or prepack or fast_lane or pipelined or shared_input or two_streams or abi" > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_sel.log
