#!/bin/bash
set +e
OUT=gpurun_out/r5sweep; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
AQLM_FUZZ_SEEDS=640 AQLM_FUZZ_GROUP_SEEDS=120 timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout=900 -k randomized > $OUT/randomized_sweep.log 2>&1; echo "sweep rc=$?"
tail -5 $OUT/randomized_sweep.log; grep -E "^FAILED" $OUT/randomized_sweep.log | head -20
