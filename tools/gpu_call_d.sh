#!/bin/bash
# full GPU suite + bench + packed microbench
set +e
OUT=gpurun_out/r2d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 tools/microbench/mb gemv quick 1x16g8P > $OUT/mb_quick.log 2>&1; echo "mb rc=$?"
grep -v "^# check.*worst-abs [0-9.e-]*$" $OUT/mb_quick.log
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -25 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2d/bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"])
for k, v in d.get("detail", {}).items():
    print(k, json.dumps(v)[:300])
print("sharded", json.dumps(d.get("sharded_70b"))[:300])
print("cpu", json.dumps(d.get("cpu_baseline"))[:300])
PY
