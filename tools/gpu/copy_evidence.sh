#!/bin/bash
# Copy what final_evidence.sh left under gpurun_out/<tag>/ into profiles/ under the round's names:  bash tools/gpu/copy_evidence.sh <tag> <round prefix>
set -e
O=gpurun_out/$1; P=profiles; R=${2:-r06}
cp $O/bench.json $P/${R}_final_bench.json; cp $O/bench_nopacked.json $P/${R}_final_bench_nopacked.json; cp $O/bench_under_rocprof.json $P/${R}_final_bench_under_rocprof.json
cp $O/prof/bench_kernel_stats.csv $P/${R}_final_bench_kernel_stats.csv; cp $O/pmc_traffic.json $P/pmc_traffic.json
cp $O/${R}_*.json $P/; cp $O/mb_gemv_quick.log $P/${R}_final_mb_gemv.log; cp $O/mb_gemm.log $P/${R}_final_mb_gemm.log; cp $O/mb_multi.log $P/${R}_final_mb_multi.log
cp $O/pytest_gpu.log $P/${R}_final_pytest_gpu.log; cp $O/smoke.log $P/${R}_final_smoke.log; cp $O/reference_triton.json $P/${R}_reference_triton.json
for f in $O/matmul_benchmark_*.json $O/decode_*.json; do cp $f $P/${R}_$(basename $f); done
cp $O/driver_form/bench.json $P/${R}_final_bench_driver_form.json; cp $O/scan_benchmark.json $P/${R}_scan_benchmark.json 2>/dev/null || true
python - <<P
import json
b = json.loads([l for l in open("$P/${R}_final_bench.json") if l.startswith("{")][-1])
print("value", round(b["value"], 1), "frac", round(b["roofline"]["frac"], 4), "traffic", b["roofline"]["traffic"])
print("worst histogram", b["detail"]["code_histograms"]["worst_vs_uniform"])
P
