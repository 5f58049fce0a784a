#!/bin/bash
# first look at the v5 packed kernel: on-device sanity vs the direct kernel, timings, then the packed parity tests
set +e
OUT=gpurun_out/r2a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 tools/microbench/mb gemv quick 1x16g8P > $OUT/mb_packed.log 2>&1; echo "mb rc=$?"
cat $OUT/mb_packed.log
timeout 900 python -m pytest tests/test_hip_parity.py -q -x --timeout=600 -k "prepack or packed or drop_canonical" > $OUT/pytest_packed.log 2>&1; echo "pytest rc=$?"
tail -40 $OUT/pytest_packed.log
