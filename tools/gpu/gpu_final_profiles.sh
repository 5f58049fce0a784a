#!/bin/bash
# evidence of the round: bench JSON, rocprofv3 kernel stats of the same command (200 steps), FETCH_SIZE / WRITE_SIZE passes
set +e
TAG=${1:-r2final}
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_fetch" -o bench -- python "$R/bench.py" --steps 4 --warmup 1 --no-detail --no-cpu > "$R/$OUT/pmc_fetch.log" 2>&1; echo "pmc fetch rc=$?"
cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_write" -o bench -- python "$R/bench.py" --steps 4 --warmup 1 --no-detail --no-cpu > "$R/$OUT/pmc_write.log" 2>&1; echo "pmc write rc=$?"
cd "$R"; python tools/make_pmc_traffic.py $OUT > $OUT/pmc_traffic.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof" -o bench -- python "$R/bench.py" --steps 200 --warmup 5 --no-detail --no-cpu > "$R/$OUT/rocprof_bench.json" 2> "$R/$OUT/rocprof.log"; echo "rocprof rc=$?"
cd "$R"
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json   # bench.py reads it back as roofline.traffic (labelled with its source)
timeout 900 python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --steps 50 --warmup 10 --no-packed --no-detail --no-cpu > $OUT/bench_nopacked.json 2> $OUT/bench_nopacked.err; echo "bench(no-packed) rc=$?"
find $OUT -name "*kernel_trace*" -delete
find $OUT -name "*counter_collection*" -size +8M -delete
find $OUT -name "*.db" -delete
head -12 $OUT/prof/*kernel_stats.csv | cut -c1-220; head -c 1200 $OUT/pmc_traffic.json; echo; head -c 700 $OUT/bench.json; echo; head -c 400 $OUT/rocprof_bench.json; echo
