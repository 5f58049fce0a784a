#!/bin/bash
# PMC passes on one microbench case: usage gpu_pmc.sh <scheme-substring> <out_features> <tag>
set +e
S=$1; O=$2; TAG=$3
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -c "" $OUT/counters_list.txt
cd /tmp
run() { # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/$OUT/$n" -o p -- "$R/tools/microbench/mb" gemv quick $S $O > "$R/$OUT/$n.log" 2>&1
  echo "$n rc=$?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcc2 FETCH_SIZE
run tcc3 WRITE_SIZE TCP_TCC_READ_REQ_sum
cd "$R"
ls $OUT/*/ | head -30
