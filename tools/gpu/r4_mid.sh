#!/bin/bash
# round 4, mid-round evidence: full GPU suite, bench line, counters of the planar look-up-table kernel
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r4mid
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
timeout 900 python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r4mid/bench.json").read().strip().splitlines()[-1])
    print("value", r["value"], "frac", r["roofline"]["frac"], "prepack_s", r["config"].get("prepack_s_total"), r["config"].get("bits_per_weight_resident"))
    d = r["detail"]
    for k in d:
        if "stack" in k:
            print(k, {kk: d[k][kk] for kk in d[k] if kk in ("ms_per_token", "tokens_per_s", "frac_of_8TBps")})
    print("parity", r.get("parity_mean_rel_vs_cpu_oracle"), "cpu", r["cpu_baseline"]["value"] if r.get("cpu_baseline") else None)
    print("sharded", {k: v for k, v in r["sharded_70b"].items() if "us" in k or "GBps" in k})
except Exception as e:
    print("bench parse failed", e)
PY
bash tools/gpu/gpu_pmc.sh 8x8g32LUTP 4096 r4mid_lutp > $OUT/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r4mid_lutp gemv_8x8_lut_kernel $OUT/lut_planar_kernel_pmc.json | tail -12
rm -rf gpurun_out/pmc_r4mid_lutp
