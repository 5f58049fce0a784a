#!/bin/bash
# SQ / TCC counter passes (separate --pmc runs, --kernel-trace only) over an arbitrary command:
#   gpu_pmc_cmd.sh <tag> <kernel-name substring> <out.json> -- <command ...>
set +e
TAG=$1; PAT=$2; JSON=$3; shift 4
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
run() { n=$1; shift; c=$1; shift
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/$OUT/$n" -o p -- "$@" > "$R/$OUT/$n.log" 2>&1; echo "$n rc=$?"; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "$@"
run sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "$@"
run sq3 "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" "$@"
run tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "$@"
run tcc2 "FETCH_SIZE" "$@"
run tcc3 "WRITE_SIZE TCP_TCC_READ_REQ_sum" "$@"
cd "$R"
python tools/pmc_summary.py $OUT "$PAT" "$JSON" > /dev/null
rm -rf $OUT
cat "$JSON" | head -60
