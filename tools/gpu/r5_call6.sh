#!/bin/bash
# Round 5, development call 6: fused 8x8 g32 MFMA kernel wired in (cost-model switch) -- full suite, smoke, benchmark table, bench detail.
set +e
TAG=${1:-r5c6}
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -8 $OUT/smoke.log
timeout 900 python tools/gemm_8x8_benchmark.py > $OUT/gemm_8x8_mfma.log 2> $OUT/gemm_8x8_mfma.err; echo "8x8 bench rc=$?"; tail -3 $OUT/gemm_8x8_mfma.err
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err

python -c "
import json,sys
b=json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][-1])
print('value',b['value']); print(json.dumps(b['detail']['small_batch_rows_8x8g32'],indent=0)[:3000])"
