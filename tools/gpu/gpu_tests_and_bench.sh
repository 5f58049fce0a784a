#!/bin/bash
set +e
OUT=gpurun_out/tests_bench
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/bench.err
tail -12 $OUT/pytest_gpu.log; tail -3 $OUT/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/tests_bench/bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"])
for k, v in d.get("detail", {}).items():
    print(k, json.dumps(v)[:330])
PY
