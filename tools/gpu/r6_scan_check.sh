#!/bin/bash
# parity tests of the slice-scan kernel + its timings next to the other routes:  bash tools/gpu/r6_scan_check.sh <tag> [shapes] [rows]
OUT=gpurun_out/${1:-r6scan}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_scan_1x16.py -x -q --timeout=600 > $OUT/pytest_scan.log 2>&1; tail -4 $OUT/pytest_scan.log
timeout 900 python tools/scan_benchmark.py --shapes ${2:-4096x4096,4096x11008,11008x4096} --rows ${3:-1,2,4,8,16,32,128} --json $OUT/scan.json > $OUT/scan.log 2>&1; tail -40 $OUT/scan.log
