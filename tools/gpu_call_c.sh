#!/bin/bash
# round evidence: HF decode benchmark (both stacks) + the final profile set
set +e
OUT=gpurun_out/r2c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/decode_benchmark.py --model llama3-8b --scheme 1x16g8 --tokens 96 > $OUT/decode_llama3_8b.json 2> $OUT/decode_llama3_8b.err; echo "decode rc=$?"
timeout 900 python tools/decode_benchmark.py --model llama2-7b --scheme 2x8g8 --tokens 96 > $OUT/decode_llama2_7b_2x8.json 2> $OUT/decode_llama2_7b_2x8.err; echo "decode rc=$?"
python - <<'PY'
import json
for f in ('decode_llama3_8b','decode_llama2_7b_2x8'):
    try:
        d=json.load(open(f'gpurun_out/r2c/{f}.json'))
        print(f, {k:round(v['tokens_per_s'],1) for k,v in d.items() if isinstance(v,dict) and 'tokens_per_s' in v})
    except Exception as e: print(f, 'ERR', e)
PY
bash tools/gpu_final_profiles.sh r2final
timeout 600 tools/microbench/mb gemv quick > gpurun_out/r2final/mb_gemv_quick.log 2>&1; echo "mb rc=$?"
