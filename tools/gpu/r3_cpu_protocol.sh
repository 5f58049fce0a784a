#!/bin/bash
# BASELINE config 1 on the GPU box's host cores: the reference's CPU benchmark protocol (benchmark/matmul_benchmark_cpu.py:
# 1 thread, fp32, dense F.linear beside it) on libaqlm_cpu.so, with and without the AVX-512 row sweep
set +e
OUT=gpurun_out/r3cpu
rm -rf $OUT; mkdir -p $OUT
lscpu | grep -E "Model name|^CPU\(s\)|Flags" | cut -c1-200 | sed 's/\(avx512[a-z_0-9]*\)/[\1]/g' | cut -c1-400 > $OUT/host.txt
timeout 200 python tools/matmul_benchmark_cpu.py --log_error --max_seconds 2.5 --json $OUT/matmul_benchmark_cpu_2x8_1thread.json 2>&1 | tail -4
AQLM_CPU_NO_AVX512=1 timeout 200 python tools/matmul_benchmark_cpu.py --log_error --max_seconds 2.5 --json $OUT/matmul_benchmark_cpu_2x8_1thread_avx2.json 2>&1 | tail -4
timeout 300 python tools/matmul_benchmark_cpu.py --log_error --nbits_per_codebook 16 --num_codebooks 1 --max_seconds 2.5 --json $OUT/matmul_benchmark_cpu_1x16_1thread.json 2>&1 | tail -4
timeout 200 python tools/matmul_benchmark_cpu.py --log_error --num_codebooks 8 --in_group_size 32 --max_seconds 2.5 --json $OUT/matmul_benchmark_cpu_8x8g32_1thread.json 2>&1 | tail -4
head -3 $OUT/host.txt | cut -c1-160
