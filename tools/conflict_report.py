"""Bank-conflict report of the DEVICE's prepacked layout: mean LDS cycles per 16-lane service group for the x reads and the
codebook reads separately (1.0 = conflict-free, ~2.85 = random).  Run on the GPU box: python tools/conflict_report.py"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import packed_model as pm
from aqlm_amd.inference_kernels import hip_kernel as hk
from aqlm_amd import _native


def report(fin, fout, arrange=1, xcopies=0):
    _native.set_tuning("packed_arrange", arrange)
    _native.set_tuning("packed_xcopies", xcopies)
    g = torch.Generator().manual_seed(1)
    codes = torch.randint(0, 65536, (fout, fin // 8, 1), generator=g).to(torch.int32)
    codes_dev = (codes - (codes >= 32768) * 65536).to(torch.int16).cuda()
    P = hk.prepack_1x16(codes_dev, 8)
    d = P.desc
    raw = P.buf.cpu().numpy().view(np.uint8)[: d.used_bytes]
    G = pm.decode_device_buffer(raw, fout, fin, d.waves, d.steps, d.entry_bytes)
    slot, code, winfo = G["slot"].astype(np.int64), G["code"].astype(np.int64), G["winfo"]
    nst = slot.shape[0]
    out = {}
    for name, f in (("x", slot), ("cb", code)):
        tot = n = 0
        hist = np.zeros(17)
        for st in range(0, nst, max(1, nst // 8)):
            for w in range(d.waves):
                for t in range(int(winfo[st, w, 2])):
                    for k in range(4):
                        v = f[st, w, t, :, k]
                        for grp in pm.SERVICE_GROUPS:
                            s = np.unique(v[grp])
                            m = np.bincount(s % 16, minlength=16).max()
                            tot += m; n += 1; hist[m] += 1
        out[name] = (tot / n, (hist / n).round(3)[1:6])
    print(f"{fin}->{fout} arrange={arrange} xcopies={xcopies}: waves {d.waves} steps {d.steps} x copies {d.x_copies}")
    for k, (m, h) in out.items():
        print(f"   {k:2s}: {m:.3f} cycles per service group; share with 1..5 cycles: {h}")
    _native.set_tuning("packed_arrange", 1)
    _native.set_tuning("packed_xcopies", 0)


if __name__ == "__main__":
    for fin, fout in ((4096, 4096), (4096, 11008), (8192, 4096)):
        for arr, xc in ((0, 0), (2, 0), (1, 0), (1, 4)):  # ascending j | greedy deal only | greedy + local search (default) | + 4 copies of x
            report(fin, fout, arr, xc)
