#!/bin/bash
# same-box A/B of bench.py's stacks with two builds of the library
set +e
OUT=gpurun_out/r3ab7
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
AB=$PWD/tools/microbench/ab
for pass in 1 2; do
  for c in 60a3f40 HEAD; do
    if [ $c = HEAD ]; then unset AQLM_AMD_HIP_LIB; else export AQLM_AMD_HIP_LIB=$AB/libaqlm_hip_$c.so; fi
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu > $OUT/bench_${c}_$pass.json 2> $OUT/bench_${c}_$pass.err; echo "bench $c rc=$?"
    python - <<PY
import json
d=json.load(open("$OUT/bench_${c}_$pass.json"))
print("$c pass $pass", round(d["value"],1), {k.replace("_linear_stack","").replace("_shared_input_launches","+S"):round(v["tokens_per_s"],1) for k,v in d["detail"].items() if "tokens_per_s" in v})
PY
  done
done
