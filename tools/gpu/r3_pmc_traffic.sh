#!/bin/bash
# HBM traffic per matvec of the final code: FETCH_SIZE and WRITE_SIZE in separate --pmc passes over a short bench run
# (no --stats / trace domains next to --pmc), condensed by tools/make_pmc_traffic.py into profiles/pmc_traffic.json
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r3pmc
rm -rf $OUT; mkdir -p $OUT
R=$PWD
cd /tmp && timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_fetch" -o bench -- python "$R/bench.py" --steps 4 --warmup 1 --no-detail --no-cpu > "$R/$OUT/pmc_fetch.log" 2>&1; echo "pmc fetch rc=$?"
cd /tmp && timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_write" -o bench -- python "$R/bench.py" --steps 4 --warmup 1 --no-detail --no-cpu > "$R/$OUT/pmc_write.log" 2>&1; echo "pmc write rc=$?"
cd "$R"; python tools/make_pmc_traffic.py $OUT > $OUT/pmc_traffic.json; head -c 1500 $OUT/pmc_traffic.json; echo
find $OUT -name "*kernel_trace*" -delete; find $OUT -name "*counter_collection*" -delete; find $OUT -name "*.db" -delete
