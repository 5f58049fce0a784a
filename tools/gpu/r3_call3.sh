#!/bin/bash
# round 3, call 3: LDS-DMA GEMM block shapes (128 / 64 rows) x partial-store policy; correctness of both
set +e
OUT=gpurun_out/r3c3
mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -k "matmat_dequant_mfma" > $OUT/pytest_gemm.log 2>&1; echo "pytest gemm rc=$?"; tail -5 $OUT/pytest_gemm.log
MB_GEMM_SWEEP=1 timeout 300 $MB/mb gemm > $OUT/mb_gemm.log 2>$OUT/mb_gemm.err; echo "mb gemm rc=$?"; grep -v dequant $OUT/mb_gemm.log
