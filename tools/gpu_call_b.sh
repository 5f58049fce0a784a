#!/bin/bash
set +e
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 tools/microbench/mb gemv quick 1x16g8P 8192 > $OUT/mb_70b_a.log 2>&1; grep "default\|^# packed" $OUT/mb_70b_a.log
timeout 900 tools/microbench/mb gemv quick 1x16g8P 1024 > $OUT/mb_70b_b.log 2>&1; grep "default\|^# packed" $OUT/mb_70b_b.log
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "headline" 2>&1 | tail -2
timeout 900 python bench.py --no-cpu > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2b/bench.json'))
print(d['value'], d['detail']['llama3_70b_1x16g8_linear_stack_one_gpu'])
PY
