#!/bin/bash
# round 4, call 6: dense escape hatch test, bench detail with the 2..8-row crossover (prepacked matvec vs MFMA op)
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r4c6
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "dense or drop_canonical or prepack" --timeout=600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 900 python bench.py --steps 50 --warmup 10 --no-cpu > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4c6/bench.json").read().strip().splitlines()[-1])
d = r["detail"]["bs128_1x16g8_4096x4096"]
print({k: v for k, v in d.items() if k.endswith("_us")})
print(json.dumps(d["graph_by_rows"]))
print(json.dumps(d["small_batch_rows"], indent=0))
print(json.dumps(r["detail"]["batch_rows_1x16g8_4096x11008_prepacked"]))
PY
