#!/bin/bash
# round 4, call 7: compiled lane for planar 8x8, many-rows LUT test, full GPU suite
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r4c7
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -5
timeout 300 python tools/matmul_benchmark.py --module --nbits_per_codebook 8 --num_codebooks 8 --in_group_size 32 --json $OUT/matmul_benchmark_8x8g32_eager.json 2>&1 | grep -v "amdgpu.ids" | tail -6
