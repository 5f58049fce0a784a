#!/bin/bash
# phase trace of the packed kernel (trace build): tools/gpu_call_e.sh "in out" ["in out" ...]
set +e
OUT=gpurun_out/r2e
mkdir -p $OUT
export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- "4096 4096" "4096 11008"; fi
for shape in "$@"; do
  timeout 300 tools/microbench/mb_trace trace $shape > $OUT/trace_${shape// /_}.log 2>&1; echo "trace rc=$?"
  grep -v "^# check" $OUT/trace_${shape// /_}.log | tail -${TAIL:-60}
done
