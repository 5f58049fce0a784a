#!/bin/bash
# CPU kernels on the GPU box's host (EPYC 9575F), 1 thread, the reference's CPU benchmark protocol: default sweeps (AMD: scalar
# look-ups) against AQLM_CPU_SWEEP=gather (AVX-512 / AVX2 gathers) for 2x8g8 and 8x8g32, and the 1 x n kernel for 1x16g8
set +e
OUT=gpurun_out/r3cpu2
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; "$@" timeout 300 python tools/matmul_benchmark_cpu.py --log_error --max_seconds 2 --json $OUT/matmul_benchmark_cpu_$name.json ${EXTRA} 2>&1 | grep -i "quant forward\|speedup" | tr '\n' ' '; echo " [$name]"; }
EXTRA="--nbits_per_codebook 16 --num_codebooks 1" run 1x16_1thread env
EXTRA="" run 2x8_1thread env
EXTRA="" run 2x8_1thread_gather env AQLM_CPU_SWEEP=gather
EXTRA="--num_codebooks 8 --in_group_size 32" run 8x8g32_1thread env
EXTRA="--num_codebooks 8 --in_group_size 32" run 8x8g32_1thread_gather env AQLM_CPU_SWEEP=gather
EXTRA="--nthreads 8" run 2x8_8threads env
EXTRA="--nbits_per_codebook 16 --num_codebooks 1 --nthreads 8" run 1x16_8threads env
