#!/bin/bash
# round 3, call 2: steady-state phase trace of the packed matvec; per-kernel durations of the large-batch op
set +e
OUT=gpurun_out/r3c2
mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench
for sh in "4096 4096" "4096 11008" "4096 1024"; do
  set -- $sh
  timeout 200 $MB/mb_trace trace $1 $2 > $OUT/trace_$1x$2.log 2>&1; echo "trace $1 $2 rc=$?"
  grep -A7 "^# packed" $OUT/trace_$1x$2.log | grep -v "by block\|by wave" | head -45
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_gemm -o g -- $MB/mb gemm > $OLDPWD/$OUT/prof_gemm.log 2>&1; echo "rocprof rc=$?"
cd $OLDPWD
find $OUT/prof_gemm -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-8 {} | head -14'
