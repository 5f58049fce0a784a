#!/bin/bash
# First GPU call: environment probe, smoke, parity tests, micro-benchmarks, bench, rocprof kernel trace.
set +e
OUT=gpurun_out/call1
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
{
  echo "== nproc: $(nproc)"; lscpu | grep -E "Model name|Socket|Thread|Core" ;
  echo "== reference present: $(ls -d /root/reference 2>&1 | head -1)"
  rocm-smi --showproductname 2>&1 | head -20
  python -c "import torch; print('torch', torch.__version__, 'hip', torch.version.hip, 'devices', torch.cuda.device_count(), torch.cuda.get_device_name(0)); p=torch.cuda.get_device_properties(0); print(p)"
  python -c "import numba" 2>&1 | tail -1
} > $OUT/env.log 2>&1
echo "--- smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
echo "--- pytest gpu"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
timeout 900 python -m pytest tests -m gpu -q --timeout=600 > $OUT/pytest_gpu_all.log 2>&1; echo "pytest(all) rc=$?" | tee -a $OUT/pytest_gpu_all.log
echo "--- microbench"
timeout 300 tools/microbench/mb l2gather > $OUT/mb_l2gather.log 2>&1
timeout 300 tools/microbench/mb ldsgather > $OUT/mb_ldsgather.log 2>&1
timeout 300 tools/microbench/mb stream > $OUT/mb_stream.log 2>&1
timeout 600 tools/microbench/mb gemv > $OUT/mb_gemv.log 2>&1
echo "--- bench"
timeout 900 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/bench.err
echo "--- rocprof"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$OUT/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-detail --no-cpu > "$GRAFT_REPO_ROOT/$OUT/rocprof.log" 2>&1; echo "rocprof rc=$?"
cd "$GRAFT_REPO_ROOT"
find $OUT/prof -name "*.csv" | head; find $OUT/prof -name "*kernel_trace*" -size +20M -delete
tail -5 $OUT/smoke.log; tail -15 $OUT/pytest_gpu.log; tail -3 $OUT/bench.err; head -c 1500 $OUT/bench.json
