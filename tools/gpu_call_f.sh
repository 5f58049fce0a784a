#!/bin/bash
# BASELINE config 1 on the GPU box's host cores: the reference's CPU benchmark protocol on libaqlm_cpu.so
set +e
OUT=gpurun_out/r2f
mkdir -p $OUT
nproc; lscpu | grep "Model name" | head -1
timeout 300 python tools/matmul_benchmark_cpu.py --log_error --max_seconds 3 --json $OUT/matmul_benchmark_cpu_2x8_1thread.json 2>&1 | grep -v Relative
N=$(nproc)
OMP_NUM_THREADS=$N timeout 300 python tools/matmul_benchmark_cpu.py --max_seconds 3 --nthreads $N --json $OUT/matmul_benchmark_cpu_2x8_allthreads.json 2>&1 | grep -v Relative
timeout 300 python tools/matmul_benchmark_cpu.py --log_error --max_seconds 3 --nbits_per_codebook 16 --num_codebooks 1 --json $OUT/matmul_benchmark_cpu_1x16_1thread.json 2>&1 | grep -v Relative
OMP_NUM_THREADS=$N timeout 300 python tools/matmul_benchmark_cpu.py --max_seconds 3 --nthreads $N --nbits_per_codebook 16 --num_codebooks 1 --json $OUT/matmul_benchmark_cpu_1x16_allthreads.json 2>&1 | grep -v Relative
