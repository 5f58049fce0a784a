#!/bin/bash
# final evidence of the round: bench JSON, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE passes
set +e
OUT=gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_fetch" -o bench -- python "$R/bench.py" --steps 4 --warmup 1 --no-detail --no-cpu > "$R/$OUT/pmc_fetch.log" 2>&1; echo "pmc fetch rc=$?"
cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_write" -o bench -- python "$R/bench.py" --steps 4 --warmup 1 --no-detail --no-cpu > "$R/$OUT/pmc_write.log" 2>&1; echo "pmc write rc=$?"
cd "$R"; python tools/make_pmc_traffic.py $OUT > $OUT/pmc_traffic.json; cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof" -o bench -- python "$R/bench.py" --steps 10 --warmup 2 --no-detail --no-cpu > "$R/$OUT/rocprof.log" 2>&1; echo "rocprof rc=$?"
cd "$R"
timeout 900 python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --steps 50 --warmup 10 --no-packed --no-detail --no-cpu > $OUT/bench_nopacked.json 2> $OUT/bench_nopacked.err; echo "bench(no-packed) rc=$?"
find $OUT -name "*kernel_trace*" -size +30M -delete
find $OUT -name "*counter_collection*" -size +30M -delete
head -8 $OUT/prof/*kernel_stats.csv | cut -c1-200; head -c 900 $OUT/pmc_traffic.json; echo; head -c 500 $OUT/bench.json; echo
