#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per dispatch of the kernels whose name contains a substring.
Usage: tools/pmc_summary.py <dir with pass sub-directories> <kernel-name substring> [out.json]"""
import csv
import glob
import json
import os
import sys


def main():
    root, pat = sys.argv[1], sys.argv[2]
    acc, cnt = {}, {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if pat not in row["Kernel_Name"]:
                    continue
                k = row["Counter_Name"]
                acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"])
                cnt[k] = cnt.get(k, 0) + 1
    out = {"kernel_substring": pat, "dispatches": max(cnt.values()) if cnt else 0,
           "counters_mean_per_dispatch": {k: acc[k] / cnt[k] for k in sorted(acc)}}
    c = out["counters_mean_per_dispatch"]
    d = {}
    if "SQ_INSTS_VALU" in c:
        d["valu_wave_instructions_per_simd"] = c["SQ_INSTS_VALU"] / 1024
    if "SQ_ACTIVE_INST_VALU" in c:
        d["valu_active_us_per_simd_at_2.4GHz"] = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / 2400
    if "SQ_LDS_IDX_ACTIVE" in c:
        d["lds_busy_us_per_cu_at_2.4GHz"] = c["SQ_LDS_IDX_ACTIVE"] / 256 / 2400
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_fraction"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "TCC_HIT_sum" in c:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "FETCH_SIZE" in c:
        d["hbm_read_bytes_gfx950_corrected"] = 2 * c["FETCH_SIZE"] * 1024
    out["derived"] = d
    s = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(s + "\n")
    print(s)


if __name__ == "__main__":
    main()
