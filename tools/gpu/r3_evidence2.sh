#!/bin/bash
# counters + kernel stats of the K x 8 kernels, the look-up-table kernel and the large-batch MFMA kernel (separate rocprofv3 passes:
# --pmc with --kernel-trace only; --stats in a run of its own), then the final microbenchmark logs
set +e
OUT=gpurun_out/r3ev2
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
MB=$R/tools/microbench/mb
cd /tmp
pmc() { # tag, counters..., then "--", then the command
  tag=$1; shift
  ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
  shift
  timeout 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d "$R/$OUT/$tag" -o p -- "$@" > "$R/$OUT/$tag.log" 2>&1
  echo "$tag rc=$?"
}
for k in "kx8 gemv quick 2x8g8 4096" "lut gemv quick 8x8g32LUT 4096" "gemm gemm"; do
  set -- $k; t=$1; shift
  pmc ${t}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -- $MB "$@"
  pmc ${t}_sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- $MB "$@"
  pmc ${t}_tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- $MB "$@"
  pmc ${t}_tcc2 FETCH_SIZE -- $MB "$@"
  pmc ${t}_tcc3 WRITE_SIZE -- $MB "$@"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/${t}_stats" -o p -- $MB "$@" > "$R/$OUT/${t}_stats.log" 2>&1; echo "${t} stats rc=$?"
done
cd "$R"
find $OUT -name "*kernel_trace*" -delete
find $OUT -name "*.db" -delete
python tools/pmc_summary.py $OUT gemv_kx8_rep_kernel $OUT/pmc_kx8_rep.json > /dev/null; python tools/pmc_summary.py $OUT gemv_8x8_lut $OUT/pmc_8x8_lut.json > /dev/null; python tools/pmc_summary.py $OUT gemm_1x16_glds_kernel $OUT/pmc_gemm_glds.json > /dev/null
find $OUT -name "*counter_collection*" -delete   # summarised above; the raw per-dispatch tables are tens of MB
find $OUT -name "*agent_info*" -delete
for f in $OUT/pmc_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['dispatches'], json.dumps(d['derived']))"; done
for t in kx8 lut gemm; do f=$(find $OUT/${t}_stats -name "*kernel_stats.csv" | head -1); echo "== $t"; head -6 "$f" | cut -c1-200; done
# final microbenchmark logs
timeout 900 $MB gemv quick > $OUT/mb_gemv_quick.log 2>&1; echo "mb gemv rc=$?"
timeout 600 $MB gemm > $OUT/mb_gemm.log 2>&1; echo "mb gemm rc=$?"
timeout 600 $MB multi > $OUT/mb_multi.log 2>&1; echo "mb multi rc=$?"
grep -c MISMATCH $OUT/mb_gemv_quick.log
du -sh $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "packed or prepack or g16 or sharded or xgmi" > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_sel.log
grep "1x16g16P\|MISMATCH" $OUT/mb_gemv_quick.log | grep " 1 default  \|MISMATCH" | head -20
