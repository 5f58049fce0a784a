#!/bin/bash
# Round 5, development call 2: new tests first, 8x8 multi-row microbenchmark, 24-bit vs 32-bit entries per shape (A/B/A/B), full suite.
set +e
TAG=${1:-r5c2}
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench/mb
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 -k "skewed or prepack or lut_rows or notices or derived_state or raw_op or 8x8_module or accumulator" > $OUT/pytest_sel.log 2>&1; echo "pytest(sel) rc=$?" | tee -a $OUT/pytest_sel.log
tail -30 $OUT/pytest_sel.log
timeout 600 $MB lutrows > $OUT/mb_lutrows.log 2>&1; echo "mb lutrows rc=$?"; cat $OUT/mb_lutrows.log | grep -v "^#"
for i in 1 2; do
  timeout 400 $MB gemv quick 1x16g8P= > $OUT/mb_eb4_$i.log 2>&1; echo "mb eb4 $i rc=$?"
  MB_TUNE=packed_entry_bytes=3 timeout 400 $MB gemv quick 1x16g8P= > $OUT/mb_eb3_$i.log 2>&1; echo "mb eb3 $i rc=$?"
done
for f in $OUT/mb_eb4_1.log $OUT/mb_eb3_1.log $OUT/mb_eb4_2.log $OUT/mb_eb3_2.log; do echo "== $f"; grep "^1x16g8P" $f | awk '{print $2, $3, $4, $5, $6, $7}' | head -40; done
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -30
