#!/bin/bash
# full GPU suite + bench line (+ optional rocprof stats); usage: r3_full.sh <tag>
set +e
TAG=${1:-full}
OUT=gpurun_out/r3_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/bench.err
tail -3 $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"])
for k, v in d.get("detail", {}).items():
    print(k, json.dumps(v)[:420])
print("cpu", json.dumps(d.get("cpu_baseline", {}))[:600])
PY
