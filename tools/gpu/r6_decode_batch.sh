#!/bin/bash
# Hugging Face decode loop of a Llama-3-8B-shaped 1x16 g8 model with 2 / 4 / 8 / 16 sequences (rows of every linear's input), hipGraph,
# against dense fp16 ("batch 8 and 16 >= 1.0 x dense"; round 5: 0.89 x):  bash tools/gpu/r6_decode_batch.sh [tag]
set +e
OUT=gpurun_out/${1:-r6dec}; mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; timeout 900 python tools/decode_benchmark.py --tokens 32 "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; python -c "
import json; r=json.load(open('$OUT/$name.json')); print({k:(round(v['tokens_per_s'],1), round(v['ms_per_token'],3)) for k,v in r.items() if isinstance(v,dict) and 'tokens_per_s' in v})"; }
for b in 2 4 8 16; do run decode_batch_llama3_8b_1x16_b$b --model llama3-8b --scheme 1x16g8 --batch $b; done
