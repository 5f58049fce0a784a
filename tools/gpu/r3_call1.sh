#!/bin/bash
# round 3, call 1: correctness of the LDS-DMA GEMM pipeline, its timing, and the packed-matvec switches (rotated fill, chain prefetch, 32 slices) + phase trace
set +e
OUT=gpurun_out/r3c1
mkdir -p $OUT
export TMPDIR=/tmp
MB=tools/microbench
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -k "matmat_dequant_mfma" > $OUT/pytest_gemm.log 2>&1; echo "pytest gemm rc=$?"; tail -15 $OUT/pytest_gemm.log
timeout 300 $MB/mb gemm > $OUT/mb_gemm.log 2>$OUT/mb_gemm.err; echo "mb gemm rc=$?"; cat $OUT/mb_gemm.log
for o in 4096 11008 1024; do
  timeout 300 $MB/mb gemv quick 1x16g8P $o > $OUT/mb_gemv_$o.log 2>&1; echo "mb gemv $o rc=$?"; grep -v "^# packed\|^# repacked" $OUT/mb_gemv_$o.log | head -60
done
timeout 200 $MB/mb_trace trace 4096 4096 > $OUT/trace_4096x4096.log 2>&1; echo "trace rc=$?"; grep -A7 "^# packed" $OUT/trace_4096x4096.log | grep -v "by block\|by wave" | head -80
timeout 200 $MB/mb_trace trace 4096 11008 > $OUT/trace_4096x11008.log 2>&1; echo "trace rc=$?"
for o in 4096 11008; do
  timeout 300 $MB/mb_s32 gemv quick 1x16g8P $o > $OUT/mb_s32_gemv_$o.log 2>&1; echo "mb_s32 gemv $o rc=$?"; grep -v "^# repacked" $OUT/mb_s32_gemv_$o.log | head -40
done
