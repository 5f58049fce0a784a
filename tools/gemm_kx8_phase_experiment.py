"""One-off: the phased X-resident K x 8 kernel with forced tiles per workgroup / quads per phase against the default plan
(tuning keys kx8_phase_tpb, kx8_phase_quads), 2x8 g8, hipGraph over 24 layers, us per call."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import aqlm_amd.inference_kernels.hip_kernel as hk
from aqlm_amd import _native
from tools.gemm_variants_benchmark import dev, timeit


def layers(fin, fout, n):
    gen = torch.Generator(device=dev).manual_seed(fin + fout)
    return [(torch.randint(-128, 128, (fout, fin // 8, 2), generator=gen, device=dev, dtype=torch.int32).to(torch.int8),
             torch.randn((2, 256, 1, 8), generator=gen, device=dev).half()) for _ in range(n)]


for fin, fout in ((4096, 11008), (4096, 4096), (4096, 14336), (11008, 4096)):
    ls = layers(fin, fout, 24)
    scales = torch.ones((fout, 1, 1, 1), device=dev, dtype=torch.float16)
    for B in (4, 8, 12, 16):
        x = torch.randn((B, fin), device=dev).half()
        res = {}
        for rep in range(2):
            for name, (tpb, q) in {"default": (0, 0), "tpb1 max": (1, 0), "tpb1 q16": (1, 16), "tpb1 q8": (1, 8), "tpb2 q16": (2, 16), "tpb3 max": (3, 0)}.items():
                _native.set_tuning("kx8_phase_tpb", tpb)
                _native.set_tuning("kx8_phase_quads", q)
                try:
                    t = timeit(lambda c, cb: hk.code2x8_matmat_dequant(x, c, cb, scales, None), ls)
                except Exception:  # noqa: BLE001
                    t = float("nan")
                res[name] = min(res.get(name, 1e9), t)
        _native.set_tuning("kx8_phase_tpb", 0)
        _native.set_tuning("kx8_phase_quads", 0)
        print(f"2x8g8 {fin}->{fout} B={B}: " + "  ".join(f"{k} {v:.2f}" for k, v in res.items()), flush=True)
