#!/bin/bash
# Final evidence of a round in ONE gpurun call (~12 GPU-minutes):  bash tools/gpu/final_evidence.sh <tag>
#   1 GPU suite                      5 microbenchmarks (mb gemv quick / gemm / multi)
#   2 PMC traffic of the bench step  6 the reference's Triton gemv on this GPU (tools/reference_triton.py)
#   3 rocprofv3 --stats of the bench 7 the reference's benchmark protocol (tools/matmul_benchmark.py) + HF decode loops
#   4 bench line (+ --no-packed)     8 kernel counters (K x 8 replicated, look-up table planar / canonical)
# Outputs land under gpurun_out/<tag>/; copy what is to be judged into profiles/ (tools/gpu/README.md).
set +e
TAG=${1:-final}
RND=${2:-r06}   # prefix of the counter files copied to profiles/
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
MB=$R/tools/microbench/mb
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -1
# kernel counters first: bench.py reads the traffic of configs 3 / 4 from profiles/
for spec in "2x8g8 4096 kx8 gemv_kx8_rep_kernel ${RND}_2x8_rep_kernel_pmc.json" "8x8g32LUTP 4096 lutp gemv_8x8_lut_kernel ${RND}_8x8_lut_planar_kernel_pmc.json" "8x8g32LUT= 4096 lutc gemv_8x8_lut_kernel ${RND}_8x8_lut_kernel_pmc.json"; do
  set -- $spec
  bash tools/gpu/gpu_pmc.sh $1 $2 ${TAG}_$3 > $OUT/pmc_$3.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$3 $4 $OUT/$5 > /dev/null
  cp $OUT/$5 profiles/$5
  rm -rf gpurun_out/pmc_${TAG}_$3
done
cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_fetch" -o bench -- python "$R/bench.py" --steps 4 --warmup 1 --no-detail --no-cpu --soft-exit > "$R/$OUT/pmc_fetch.log" 2>&1; echo "pmc fetch rc=$?"
cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_write" -o bench -- python "$R/bench.py" --steps 4 --warmup 1 --no-detail --no-cpu --soft-exit > "$R/$OUT/pmc_write.log" 2>&1; echo "pmc write rc=$?"
cd "$R"; python tools/make_pmc_traffic.py $OUT > $OUT/pmc_traffic.json; cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof" -o bench -- python "$R/bench.py" --steps 200 --warmup 5 --no-detail --no-cpu --soft-exit > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/rocprof.log"; echo "rocprof rc=$?"
cd "$R"
find $OUT -name "*kernel_trace*" -delete; find $OUT -name "*counter_collection*" -delete; find $OUT -name "*.db" -delete
timeout 1200 python bench.py --steps 50 --warmup 10 --full-detail > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
# what the DRIVER runs, and the assertion it makes on it: the last stdout line parses and carries roofline + cpu_baseline (round 5
# lost its record to a library banner behind the line)
bash tools/gpu/r6_bench_check.sh $TAG/driver_form
timeout 600 python bench.py --steps 50 --warmup 10 --no-packed --no-detail --no-cpu > $OUT/bench_nopacked.json 2> $OUT/bench_nopacked.err; echo "bench(no-packed) rc=$?"
head -8 $OUT/prof/*kernel_stats.csv | cut -c1-200; head -c 700 $OUT/bench.json; echo
python3 - $OUT/bench.json <<'PY'
import json, sys
lines = open(sys.argv[1]).read().splitlines()
assert len(lines) == 1, f"bench.json: {len(lines)} lines on stdout"
d = json.loads(lines[-1])
assert d.get("roofline") and d.get("cpu_baseline"), "bench.json: roofline / cpu_baseline missing"
print("bench.json parses: value %.1f GB/s, frac %.4f, wall %s s, skipped %s" % (d["value"], d["roofline"]["frac"], d.get("bench_wall_s"), d.get("detail", {}).get("skipped")))
PY
timeout 900 $MB gemv quick > $OUT/mb_gemv_quick.log 2>&1; echo "mb gemv rc=$?"
timeout 300 $MB gemm > $OUT/mb_gemm.log 2>&1; echo "mb gemm rc=$?"
timeout 300 $MB multi > $OUT/mb_multi.log 2>&1; echo "mb multi rc=$?"
grep -c MISMATCH $OUT/mb_gemv_quick.log $OUT/mb_multi.log
timeout 900 python tools/scan_benchmark.py --json $OUT/scan_benchmark.json > $OUT/scan_benchmark.log 2>&1; echo "scan benchmark rc=$?"
timeout 600 python tools/reference_triton.py --out $OUT/reference_triton.json > $OUT/reference_triton.log 2>&1; echo "reference triton rc=$?"
bm() { name=$1; shift; timeout 600 python tools/matmul_benchmark.py --module "$@" --json $OUT/matmul_benchmark_$name.json 2>&1 | grep -i "speedup" | tr '\n' ' '; echo " [$name]"; }
bm 1x16_eager; bm 1x16_graph --graph
bm 2x8_eager --num_codebooks 2 --nbits_per_codebook 8; bm 2x8_graph --graph --num_codebooks 2 --nbits_per_codebook 8
bm 8x8g32_eager --num_codebooks 8 --nbits_per_codebook 8 --in_group_size 32; bm 8x8g32_graph --graph --num_codebooks 8 --nbits_per_codebook 8 --in_group_size 32
bm 1x16g16_eager --in_group_size 16
timeout 900 python tools/decode_benchmark.py --model llama3-8b --tokens 32 > $OUT/decode_llama3_8b.json 2> $OUT/decode_llama3_8b.err; echo "decode llama3 rc=$?"
timeout 900 python tools/decode_benchmark.py --model llama2-7b --scheme 2x8g8 --tokens 32 > $OUT/decode_llama2_7b_2x8.json 2> $OUT/decode_llama2_7b.err; echo "decode llama2 2x8 rc=$?"
timeout 900 python tools/decode_benchmark.py --model llama2-7b --scheme 8x8g32 --tokens 32 > $OUT/decode_llama2_7b_8x8g32.json 2> $OUT/decode_llama2_7b_8x8.err; echo "decode llama2 8x8g32 rc=$?"
cpu() { name=$1; shift; timeout 300 python tools/matmul_benchmark_cpu.py --log_error --max_seconds 2 --json $OUT/matmul_benchmark_cpu_$name.json "$@" 2>&1 | grep -i "speedup" | tr '\n' ' '; echo " [cpu $name]"; }
cpu 1x16_1thread --nbits_per_codebook 16 --num_codebooks 1; cpu 2x8_1thread; cpu 8x8g32_1thread --num_codebooks 8 --in_group_size 32
du -sh $OUT
