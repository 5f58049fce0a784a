#!/bin/bash
set +e
OUT=gpurun_out/r2f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -q -x --timeout=600 -k "xgmi or sharded" > $OUT/pytest_xgmi.log 2>&1; echo "pytest rc=$?"
tail -30 $OUT/pytest_xgmi.log
