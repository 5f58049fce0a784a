#!/bin/bash
# Round 5, development call 5: fused 8x8 g32 MFMA kernel -- variants (waves, plane interleave), counters, full suite.
set +e
TAG=${1:-r5c5}
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x --timeout=300 -k "fused_8x8 or lut_rows" > $OUT/pytest_sel.log 2>&1; echo "pytest(sel) rc=$?" | tee -a $OUT/pytest_sel.log
tail -15 $OUT/pytest_sel.log
timeout 600 python tools/gemm_8x8_benchmark.py variants > $OUT/gemm_8x8_variants.log 2> $OUT/gemm_8x8_variants.err; echo "variants rc=$?"; cat $OUT/gemm_8x8_variants.log; tail -3 $OUT/gemm_8x8_variants.err
bash tools/gpu/gpu_pmc_cmd.sh ${TAG}_e8 gemm_8x8g32_rows16 $OUT/r05_gemm_8x8_mfma_pmc.json -- python $PWD/tools/gemm_8x8_benchmark.py one 4096 4096 16 > $OUT/pmc.log 2>&1; cat $OUT/r05_gemm_8x8_mfma_pmc.json
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -30
