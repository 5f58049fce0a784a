#!/bin/bash
set +e
OUT=gpurun_out/call7
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests -m gpu -q --timeout=600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/bench.err
timeout 900 python bench.py --steps 30 --warmup 5 --no-packed --no-detail --no-cpu > $OUT/bench_nopacked.json 2> $OUT/bench_nopacked.err; echo "bench(no-packed) rc=$?"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof" -o bench -- python "$R/bench.py" --steps 10 --warmup 2 --no-detail --no-cpu > "$R/$OUT/rocprof.log" 2>&1; echo "rocprof rc=$?"
cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_fetch" -o bench -- python "$R/bench.py" --steps 4 --warmup 1 --no-detail --no-cpu > "$R/$OUT/pmc_fetch.log" 2>&1; echo "pmc fetch rc=$?"
cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_write" -o bench -- python "$R/bench.py" --steps 4 --warmup 1 --no-detail --no-cpu > "$R/$OUT/pmc_write.log" 2>&1; echo "pmc write rc=$?"
cd "$R"
find $OUT -name "*kernel_trace*" -size +30M -delete
tail -5 $OUT/pytest_gpu.log; tail -2 $OUT/bench.err; head -c 600 $OUT/bench.json; echo; head -c 400 $OUT/bench_nopacked.json; echo; cat $OUT/prof/bench_kernel_stats.csv | head -8 | cut -c1-200
