#!/bin/bash
set +e
OUT=gpurun_out/call11
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof" -o decode -- python "$R/tools/decode_benchmark.py" --model llama3-8b --tokens 24 --no-dense > "$R/$OUT/decode.json" 2> "$R/$OUT/decode.err"; echo "rocprof rc=$?"
cd "$R"
find $OUT -name "*kernel_trace*" -size +40M -delete
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -c1-230
cat $OUT/decode.json | head -c 1500
