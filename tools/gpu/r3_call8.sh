#!/bin/bash
set +e
OUT=gpurun_out/r3c8
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "fast_lane or two_streams or inference_mode or sharded_layer_repacks or packed_fused_finalize or raw_op or hipgraph or shared_input or module" > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_sel.log
timeout 600 python tools/matmul_benchmark.py --log_error --module --json $OUT/matmul_benchmark_1x16_eager.json 2>&1 | grep -v "amdgpu.ids" | tail -12
AQLM_AMD_NO_FRONT=1 timeout 600 python tools/matmul_benchmark.py --module --json $OUT/matmul_benchmark_1x16_eager_nofront.json 2>&1 | grep -v "amdgpu.ids" | tail -6
