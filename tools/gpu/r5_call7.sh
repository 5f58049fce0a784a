#!/bin/bash
# Round 5, development call 7: counters of the fused K x 8 MFMA kernels (X-resident at 16 rows, streaming at 128 rows), 2x8 g8 4096 x 4096.
set +e
TAG=${1:-r5c7}
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/gpu/gpu_pmc_cmd.sh ${TAG}_xres gemm_kx8_xres_kernel $OUT/r05_gemm_kx8_xres_16rows_pmc.json -- python $PWD/tools/gemm_kx8_xres_benchmark.py one 2 4096 4096 16 > $OUT/pmc_xres.log 2>&1
python -c "import json; j=json.load(open('$OUT/r05_gemm_kx8_xres_16rows_pmc.json')); print(json.dumps(j['derived'],indent=0)); c=j['counters_mean_per_dispatch']; print({k:round(c[k]) for k in ('SQ_BUSY_CYCLES','SQ_LDS_IDX_ACTIVE','SQ_LDS_BANK_CONFLICT','SQ_VALU_MFMA_BUSY_CYCLES','SQ_INSTS_LDS','SQ_INSTS_VMEM_RD','SQ_WAVE_CYCLES','SQ_WAIT_INST_LDS','SQ_WAIT_INST_ANY') if k in c})"
bash tools/gpu/gpu_pmc_cmd.sh ${TAG}_str gemm_kx8_rows16_kernel $OUT/r05_gemm_kx8_rows16_128rows_pmc.json -- python $PWD/tools/gemm_kx8_xres_benchmark.py one 2 4096 4096 128 > $OUT/pmc_str.log 2>&1
python -c "import json; j=json.load(open('$OUT/r05_gemm_kx8_rows16_128rows_pmc.json')); print(json.dumps(j['derived'],indent=0)); c=j['counters_mean_per_dispatch']; print({k:round(c[k]) for k in ('SQ_BUSY_CYCLES','SQ_LDS_IDX_ACTIVE','SQ_LDS_BANK_CONFLICT','SQ_VALU_MFMA_BUSY_CYCLES','SQ_INSTS_LDS','SQ_INSTS_VMEM_RD','SQ_WAVE_CYCLES','SQ_WAIT_INST_LDS','SQ_WAIT_INST_ANY') if k in c})"
timeout 300 python -m pytest tests -m gpu -q -x --timeout=300 -k "fused_8x8" 2>&1 | tail -3
