#!/bin/bash
# instruction-rate micro-benchmarks + PMC counters of the v5 packed kernel (4096->11008)
set +e
OUT=gpurun_out/r2c
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
timeout 120 tools/microbench/mb rates > $OUT/mb_rates.log 2>&1; echo "rates rc=$?"; cat $OUT/mb_rates.log
cd /tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/$OUT/$n" -o p -- "$R/tools/microbench/mb" gemv quick 1x16g8P 11008 > "$R/$OUT/$n.log" 2>&1
  echo "$n rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run tcc2 FETCH_SIZE
cd "$R"
python tools/pmc_summary.py $OUT gemv_1x16_packed_kernel $OUT/packed_kernel_pmc.json
# keep the merged output small
find $OUT -name "*.csv" -size +2M -delete
