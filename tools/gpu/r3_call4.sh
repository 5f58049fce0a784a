#!/bin/bash
# round 3, call 4: kernel-by-kernel timeline of the large-batch op (durations and gaps per batch size)
set +e
OUT=gpurun_out/r3c4
mkdir -p $OUT
export TMPDIR=/tmp
MB=$PWD/tools/microbench
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/prof_gemm -o g -- $MB/mb gemm > $R/$OUT/prof_gemm.log 2>&1; echo "rocprof rc=$?"
cd $R
python3 - <<'PY'
import csv, collections, glob
f = glob.glob('gpurun_out/r3c4/prof_gemm/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# pair main -> finalize for the LDS-DMA pipeline, by template args
stat = collections.defaultdict(lambda: collections.defaultdict(list))
prev = None
for r in rows:
    n = r['Kernel_Name']
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if prev is not None:
        pn, ps, pe = prev
        if 'gemm_1x16_glds_kernel' in pn and 'glds_finalize' in n:
            key = pn.split('glds_kernel')[1][:22]
            stat[key]['main'].append(pe - ps); stat[key]['gap'].append(s - pe); stat[key]['fin'].append(e - s)
        if 'glds_finalize' in pn and 'gemm_1x16_glds_kernel' in n:
            key = n.split('glds_kernel')[1][:22]
            stat[key]['gap_before_main'].append(s - pe)
        if 'gemm_1x16_mfma_kernel' in pn and 'gemm_finalize' in n:
            key = 'old' + pn.split('mfma_kernel')[1][:18]
            stat[key]['main'].append(pe - ps); stat[key]['gap'].append(s - pe); stat[key]['fin'].append(e - s)
    prev = (n, s, e)
for k, d in sorted(stat.items()):
    print(k, {kk: round(sorted(v)[len(v)//2] / 1e3, 2) for kk, v in d.items()}, len(d['main']))
PY
