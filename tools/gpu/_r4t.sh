#!/bin/bash
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r4t
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "mfma or matmat_dequant or raw_op" --timeout=600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 600 python - <<'PY'
import sys, json, torch
sys.path.insert(0, ".")
import bench
from aqlm_amd.inference_kernels import hip_kernel as hk
dev = torch.device("cuda:0")
for flag in (True, False, True, False):
    hk.FUSED_MFMA_TICKETS = flag
    d = bench.large_batch_detail(dev, 20)
    print("tickets", flag, json.dumps(d["graph_by_rows"]), {k: round(v["mfma_op_us"], 2) for k, v in d["small_batch_rows"]["4096->4096"].items()}, {k: round(v["mfma_op_us"], 2) for k, v in d["small_batch_rows"]["4096->11008"].items()})
PY
