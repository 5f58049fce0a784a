#!/bin/bash
# the reference's own benchmark protocol (benchmark/matmul_benchmark.py) eager + hipGraph, and the Hugging Face decode loop
set +e
OUT=gpurun_out/r3ev3
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/matmul_benchmark.py --module --json $OUT/matmul_benchmark_1x16_eager.json 2>&1 | grep -v "amdgpu.ids" | tail -8
timeout 600 python tools/matmul_benchmark.py --module --graph --json $OUT/matmul_benchmark_1x16_graph.json 2>&1 | grep -v "amdgpu.ids" | tail -8
timeout 600 python tools/matmul_benchmark.py --module --num_codebooks 2 --nbits_per_codebook 8 --json $OUT/matmul_benchmark_2x8_eager.json 2>&1 | grep -v "amdgpu.ids" | tail -6
timeout 600 python tools/matmul_benchmark.py --module --graph --num_codebooks 2 --nbits_per_codebook 8 --json $OUT/matmul_benchmark_2x8_graph.json 2>&1 | grep -v "amdgpu.ids" | tail -6
timeout 600 python tools/matmul_benchmark.py --module --in_group_size 16 --json $OUT/matmul_benchmark_1x16g16_eager.json 2>&1 | grep -v "amdgpu.ids" | tail -6
timeout 900 python tools/decode_benchmark.py --model llama3-8b --tokens 32 > $OUT/decode_llama3_8b.json 2> $OUT/decode_llama3_8b.err; echo "decode llama3 rc=$?"; head -c 1500 $OUT/decode_llama3_8b.json; echo
timeout 900 python tools/decode_benchmark.py --model llama2-7b --scheme 2x8g8 --tokens 32 > $OUT/decode_llama2_7b_2x8.json 2> $OUT/decode_llama2_7b.err; echo "decode llama2 rc=$?"; head -c 1500 $OUT/decode_llama2_7b_2x8.json; echo
