#!/bin/bash
# final evidence after the prepack change (only what depends on the packed layout is re-taken: GPU suite, bench + rocprof stats,
# mb gemv / multi, the reference protocol for 1x16, the Llama-3-8B decode loop); PMC traffic, direct-kernel bench, mb gemm and the
# 2x8 runs are unchanged (tools/gpu/r3_final_all.sh takes everything)
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r3f2
rm -rf $OUT; mkdir -p $OUT
R=$PWD
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -1
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof" -o bench -- python "$R/bench.py" --steps 200 --warmup 5 --no-detail --no-cpu > "$R/$OUT/rocprof_bench.json" 2> "$R/$OUT/rocprof.log"; echo "rocprof rc=$?"
cd "$R"
find $OUT -name "*kernel_trace*" -delete; find $OUT -name "*.db" -delete
timeout 600 python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
head -6 $OUT/prof/*kernel_stats.csv | cut -c1-200; head -c 600 $OUT/bench.json; echo
MB=$PWD/tools/microbench/mb
timeout 600 $MB gemv quick > $OUT/mb_gemv_quick.log 2>&1; echo "mb gemv rc=$?"
timeout 300 $MB multi > $OUT/mb_multi.log 2>&1; echo "mb multi rc=$?"
grep -c MISMATCH $OUT/mb_gemv_quick.log $OUT/mb_multi.log
timeout 300 python tools/matmul_benchmark.py --module --json $OUT/matmul_benchmark_1x16_eager.json 2>&1 | grep -v "amdgpu.ids" | tail -4
timeout 300 python tools/matmul_benchmark.py --module --graph --json $OUT/matmul_benchmark_1x16_graph.json 2>&1 | grep -v "amdgpu.ids" | tail -4
timeout 300 python tools/matmul_benchmark.py --module --in_group_size 16 --json $OUT/matmul_benchmark_1x16g16_eager.json 2>&1 | grep -v "amdgpu.ids" | tail -4
timeout 600 python tools/decode_benchmark.py --model llama3-8b --tokens 32 > $OUT/decode_llama3_8b.json 2> $OUT/decode_llama3_8b.err; echo "decode llama3 rc=$?"; head -c 1200 $OUT/decode_llama3_8b.json; echo
