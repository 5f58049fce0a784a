#!/usr/bin/env python
"""Build profiles/pmc_traffic.json from two rocprofv3 counter passes over bench.py (FETCH_SIZE, WRITE_SIZE).

    python tools/make_pmc_traffic.py <dir with pmc_fetch/ and pmc_write/> > profiles/pmc_traffic.json

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request, so read bytes =
2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is already bytes.  The factor is re-checked in the same trace on a torch
elementwise copy kernel whose traffic is known (bench.py casts the int32 random codes to int16 when it builds a layer).
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    files = glob.glob(f"{path}/**/*counter_collection.csv", recursive=True)
    if not files:
        return {}
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = acc[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def is_timed_matvec(name):
    """Kernels of the timed step only: the 1x16 matvec kernels (packed / direct) and their finalize.  The set-up kernels of the
    same process (the parameter checksums of round 5, the generic kernel of the parity tripwire) are listed but not averaged in --
    round 5's first summary did average the 64 checksum launches in and reported 17.7 MB for what is 20.4 MB per matvec."""
    return "gemv_1x16" in name or ("gemv_kernel" in name and "generic" not in name) or "finalize" in name


def main():
    root = sys.argv[1]
    fetch = per_kernel(f"{root}/pmc_fetch", "FETCH_SIZE")
    write = per_kernel(f"{root}/pmc_write", "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over "
                     "`python bench.py --steps 4 --warmup 1 --no-detail --no-cpu`",
           "correction": "gfx950: read bytes = 2 x FETCH_SIZE x 1024 (FETCH_SIZE counts 64 B per 128-B request); write bytes = "
                         "WRITE_SIZE x 1024", "per_kernel": {}}
    total = 0.0
    launches = 0
    for name in sorted(set(fetch) | set(write)):
        # load-time repack kernels (pk_*_kernel) are not the hot path; pk_g8 / pk_g16 are the namespaces of the two packed builds
        if "aqlm::" not in name or "prepack" in name or re.search(r"::pk_[a-z0-9]+_kernel", name):
            continue
        timed = is_timed_matvec(name)
        f, nf = fetch.get(name, (0.0, 0))
        w, _ = write.get(name, (0.0, 0))
        hbm = 2 * f * 1024 + w * 1024
        out["per_kernel"][name[:110]] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes": hbm, "launches": nf, "in_the_timed_step": timed}
        if not timed:
            continue
        total += hbm * nf
        if "finalize" not in name:
            launches += nf
    out["gemv_1x16_hbm_bytes_per_launch"] = total / max(launches, 1)
    out["how"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 4 --warmup 1 "
                  "--no-detail --no-cpu`, read bytes = 2 x FETCH_SIZE x 1024 (gfx950), tools/make_pmc_traffic.py")
    out["note"] = ("per matvec of the bench step = (all matvec kernels of the timed launches) / number of matvecs; "
                   "algorithmic bytes per matvec 8 820 224: the prepacked path (32-bit entries) reads ~4.5-4.8 B per "
                   "code plus the row-start table and the codebook slices")
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
