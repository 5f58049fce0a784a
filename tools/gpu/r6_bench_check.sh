#!/bin/bash
# The driver's own bench command + the assertion the driver makes on it (last stdout line parses, carries roofline + cpu_baseline).
#   bash tools/gpu/r6_bench_check.sh <tag> [extra bench args]
set +e
TAG=${1:-r6bench}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python3 - $OUT/bench.json <<'PY'
import json, sys
lines = open(sys.argv[1]).read().splitlines()
assert len(lines) == 1, f"{len(lines)} lines on stdout"
d = json.loads(lines[-1])
assert d.get("roofline") and d.get("cpu_baseline"), "roofline / cpu_baseline missing"
print("PARSED OK value %.1f GB/s frac %.4f ms_per_step %.4f cpu %.1f GB/s bench_wall_s %s" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["cpu_baseline"]["value"], d.get("bench_wall_s")))
det = d.get("detail", {})
print("sections", det.get("section_seconds"), "skipped", det.get("skipped"))
print("errors", {k: v for k, v in det.items() if k.endswith("_error")}, d.get("sharded_70b", {}).get("error"))
PY
tail -5 $OUT/bench.err
