#!/bin/bash
# full GPU test-suite + default bench (what the driver runs at round end)
set +e
OUT=gpurun_out/r2d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -5
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2d/bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['detail']['1x16g8 4096->4096']['cold_us'], d['detail']['1x16g8 4096->11008']['cold_us'])
for k in ('llama3_8b_1x16g8_linear_stack','llama3_8b_1x16g8_linear_stack_shared_input_launches','llama2_7b_2x8g8_linear_stack_shared_input_launches','llama2_7b_8x8g32_linear_stack_shared_input_launches'):
    print(k, d['detail'][k]['tokens_per_s'])
print(d['sharded_70b'])
PY
