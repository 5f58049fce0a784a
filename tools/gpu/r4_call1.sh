#!/bin/bash
# round 4, call 1: conflict-free look-up-table kernel (parity, microbenchmark, counters) + the reference's Triton gemv on this GPU
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r4c1
rm -rf $OUT; mkdir -p $OUT
R=$PWD
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "lut or 8x8 or kx8" --timeout=300 > $OUT/pytest_lut.log 2>&1; echo "pytest lut rc=$?"
tail -3 $OUT/pytest_lut.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
MB=$PWD/tools/microbench/mb
timeout 300 $MB gemv quick 8x8 > $OUT/mb_8x8.log 2>&1; echo "mb rc=$?"; cat $OUT/mb_8x8.log
timeout 600 python tools/reference_triton.py --out $OUT/reference_triton.json > $OUT/reference_triton.log 2>&1; echo "triton rc=$?"; tail -60 $OUT/reference_triton.log
timeout 300 python -m pytest tests/test_reference_triton.py -m gpu -q --timeout=280 > $OUT/pytest_triton.log 2>&1; echo "pytest triton rc=$?"; tail -3 $OUT/pytest_triton.log
bash tools/gpu/gpu_pmc.sh 8x8g32LUT 4096 r4c1_lut > $OUT/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r4c1_lut gemv_8x8_lut_kernel $OUT/lut_kernel_pmc.json | tail -30
find gpurun_out/pmc_r4c1_lut -name "*.db" -delete; find gpurun_out/pmc_r4c1_lut -name "*kernel_trace*" -delete
