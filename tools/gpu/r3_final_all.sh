#!/bin/bash
# final evidence of round 3 in one call: GPU suite, bench (+ rocprof stats, PMC traffic, no-packed), microbenchmarks, reference protocol, HF decode loop
set +e
export TMPDIR=/tmp
bash tools/gpu/r3_full.sh final5 > gpurun_out/r3_final5_stdout.log 2>&1
grep -E "passed|failed" gpurun_out/r3_final5/pytest_gpu.log | tail -1
bash tools/gpu/gpu_final_profiles.sh r3final > gpurun_out/r3final_stdout.log 2>&1
tail -2 gpurun_out/r3final_stdout.log | cut -c1-160
OUT=gpurun_out/r3ev4
rm -rf $OUT; mkdir -p $OUT
MB=$PWD/tools/microbench/mb
timeout 900 $MB gemv quick > $OUT/mb_gemv_quick.log 2>&1; echo "mb gemv rc=$?"
timeout 300 $MB gemm > $OUT/mb_gemm.log 2>&1; echo "mb gemm rc=$?"
timeout 300 $MB multi > $OUT/mb_multi.log 2>&1; echo "mb multi rc=$?"
grep -c MISMATCH $OUT/mb_gemv_quick.log $OUT/mb_multi.log
bash tools/gpu/r3_evidence3.sh > $OUT/evidence3_stdout.log 2>&1; tail -4 $OUT/evidence3_stdout.log | cut -c1-400
