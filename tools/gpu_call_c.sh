#!/bin/bash
# round evidence: the final profile set + microbenchmarks + decode benchmarks + the reference's benchmark protocol
set +e
OUT=gpurun_out/r2c
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/gpu_final_profiles.sh r2final
timeout 600 tools/microbench/mb gemv quick > gpurun_out/r2final/mb_gemv_quick.log 2>&1; echo "mb rc=$?"
timeout 900 python tools/decode_benchmark.py --model llama3-8b --scheme 1x16g8 --tokens 96 > $OUT/decode_llama3_8b.json 2> $OUT/decode_llama3_8b.err; echo "decode rc=$?"
timeout 900 python tools/decode_benchmark.py --model llama2-7b --scheme 2x8g8 --tokens 96 > $OUT/decode_llama2_7b_2x8.json 2> $OUT/decode_llama2_7b_2x8.err; echo "decode rc=$?"
for mode in "" "--graph"; do
  tag=eager; [ -n "$mode" ] && tag=graph
  timeout 600 python tools/matmul_benchmark.py --log_error --module $mode --json $OUT/matmul_benchmark_1x16_$tag.json > $OUT/matmul_benchmark_1x16_$tag.log 2>&1
  timeout 600 python tools/matmul_benchmark.py --log_error --module $mode --nbits_per_codebook 8 --num_codebooks 2 --json $OUT/matmul_benchmark_2x8_$tag.json > $OUT/matmul_benchmark_2x8_$tag.log 2>&1
done
python - <<'PY'
import json
for f in ('decode_llama3_8b','decode_llama2_7b_2x8'):
    try:
        d=json.load(open(f'gpurun_out/r2c/{f}.json'))
        print(f, {k:round(v['tokens_per_s'],1) for k,v in d.items() if isinstance(v,dict) and 'tokens_per_s' in v})
    except Exception as e: print(f, 'ERR', e)
for t in ('1x16_eager','2x8_eager','1x16_graph','2x8_graph'):
    try:
        d=json.load(open(f'gpurun_out/r2c/matmul_benchmark_{t}.json'))['results']
        print(t, {k:(round(v['dense_us'],1),round(v['quant_us'],1),round(v['speedup'],2),round(v.get('module_speedup',0),2)) for k,v in d.items()})
    except Exception as e: print(t, 'ERR', e)
PY
