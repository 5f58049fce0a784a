#!/bin/bash
# full GPU suite + the reference's matmul benchmark protocol (eager and hipGraph), 1x16 and 2x8
set +e
OUT=gpurun_out/r2a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -5
for mode in "" "--graph"; do
  tag=eager; [ -n "$mode" ] && tag=graph
  timeout 600 python tools/matmul_benchmark.py --log_error --module $mode --json $OUT/matmul_benchmark_1x16_$tag.json 2>&1 | grep -v "amdgpu.ids\|Relative" 
  timeout 600 python tools/matmul_benchmark.py --log_error --module $mode --nbits_per_codebook 8 --num_codebooks 2 --json $OUT/matmul_benchmark_2x8_$tag.json 2>&1 | grep -v "amdgpu.ids\|Relative"
done
