#!/bin/bash
# rocprofv3 kernel stats of the Hugging Face decode loop:  bash tools/gpu/gpu_prof_decode.sh [tag] [decode_benchmark.py arguments ...]
#   default: --model llama3-8b --tokens 24 --no-dense;  e.g.  gpu_prof_decode.sh b4 --model llama2-7b --scheme 2x8g8 --batch 4 --tokens 24 --no-dense
set +e
TAG=${1:-decode}; shift
ARGS=${@:---model llama3-8b --tokens 24 --no-dense}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof" -o decode -- python "$R/tools/decode_benchmark.py" $ARGS > "$R/$OUT/decode.json" 2> "$R/$OUT/decode.err"; echo "rocprof rc=$?"
cd "$R"
find $OUT -name "*kernel_trace*" -delete
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
head -16 "$f" | cut -c1-200
