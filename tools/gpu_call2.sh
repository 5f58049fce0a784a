#!/bin/bash
set +e
OUT=gpurun_out/call2
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
timeout 600 tools/microbench/mb gemv quick > $OUT/mb_gemv_quick.log 2>&1
R=$PWD
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof_mb" -o mb -- "$R/tools/microbench/mb" gemv quick > "$R/$OUT/rocprof_mb.log" 2>&1; echo "rocprof rc=$?"
cd "$R"
find $OUT/prof_mb -name "*kernel_trace*" -size +30M -delete
ls -la $OUT/prof_mb | head
tail -25 $OUT/pytest_gpu.log
cat $OUT/mb_gemv_quick.log
