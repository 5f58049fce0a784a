#!/bin/bash
# Round 5: Hugging Face decode loop at batch 4 / 8 (rows of every linear's input), hipGraph, with / without shared-input launches,
# round 5's small-batch kernels against their round-4 routes (tuning keys) and dense fp16.
set +e
OUT=gpurun_out/r5dec; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R4="kx8_xres=0,kx8_xres_phased=0,kx8_multi_xres_min_rows=0,kx8_mfma_min_rows=3"
run() { name=$1; shift; timeout 600 python tools/decode_benchmark.py --tokens 32 "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; python -c "
import json; r=json.load(open('$OUT/$name.json')); print({k:(round(v['tokens_per_s'],1), round(v['ms_per_token'],3)) for k,v in r.items() if isinstance(v,dict) and 'tokens_per_s' in v})"; }
run llama2_7b_2x8_b4 --model llama2-7b --scheme 2x8g8 --batch 4
run llama2_7b_2x8_b4_r4routes --model llama2-7b --scheme 2x8g8 --batch 4 --no-dense --tune $R4
run llama2_7b_2x8_b8 --model llama2-7b --scheme 2x8g8 --batch 8
run llama2_7b_2x8_b8_r4routes --model llama2-7b --scheme 2x8g8 --batch 8 --no-dense --tune $R4
run llama2_7b_8x8g32_b4 --model llama2-7b --scheme 8x8g32 --batch 4 --no-dense
run llama3_8b_1x16_b4 --model llama3-8b --scheme 1x16g8 --batch 4
