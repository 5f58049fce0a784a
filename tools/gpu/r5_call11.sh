#!/bin/bash
# Round 5, development call 11: K-split form of the fused K x 8 MFMA op at 64+ rows -- tests, then plans side by side.
set +e
TAG=${1:-r5c11}
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 -k "kx8 or sweep or randomized" > $OUT/pytest_sel.log 2>&1; echo "pytest(sel) rc=$?" | tee -a $OUT/pytest_sel.log
tail -12 $OUT/pytest_sel.log
timeout 900 python tools/gemm_kx8_benchmark.py ksplit > $OUT/gemm_kx8_ksplit.log 2> $OUT/gemm_kx8_ksplit.err; echo "ksplit bench rc=$?"; cat $OUT/gemm_kx8_ksplit.log; tail -3 $OUT/gemm_kx8_ksplit.err
