"""Large-batch ops of the 8-bit schemes: the fused dequant -> MFMA kernel (aqlm_hip_gemm_kx8_mfma) against the dequantise + library-GEMM
route of the same op (the reference's pipeline) and a dense fp16 GEMM on rotating weights; hipGraph replay over 24 layers.

    python tools/gemm_kx8_benchmark.py > profiles/r04_gemm_kx8_shapes.log
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aqlm_amd.inference_kernels.hip_kernel as hk
from tools.gemm_variants_benchmark import timeit, dev
def layers(fin, fout, K, n):
    gen = torch.Generator(device=dev).manual_seed(fin + fout)
    out = []
    for _ in range(n):
        codes = torch.randint(-128, 128, (fout, fin // 8, K), generator=gen, device=dev, dtype=torch.int32).to(torch.int8)
        out.append((codes, torch.randn((K, 256, 1, 8), generator=gen, device=dev).half()))
    return out
if len(sys.argv) > 1 and sys.argv[1] == "ksplit":  # the K-split form at 64+ rows: plan, never, forced 2 / 4, one / two row tiles per block
    from aqlm_amd import _native
    for fin, fout in ((4096, 4096), (4096, 11008), (11008, 4096), (8192, 8192), (4096, 1024)):
        ls = layers(fin, fout, 2, 24)
        scales = torch.ones((fout, 1, 1, 1), device=dev, dtype=torch.float16)
        for B in (64, 128, 256):
            x = torch.randn((B, fin), device=dev).half()
            res = {}
            cases = {"no split (round 4)": (1, 0), "plan": (0, 0)}
            if len(sys.argv) > 2 and sys.argv[2] == "all":
                cases.update({"2 slices x 2 tiles": (2, 2), "4 x 2": (4, 2), "2 x 1": (2, 1), "4 x 1": (4, 1)})
            for rep in range(2):
                for name, (ks, rt) in cases.items():
                    _native.set_tuning("kx8_ksplit", ks)
                    _native.set_tuning("kx8_rt", rt)
                    try:
                        t = timeit(lambda c, cb: hk.code2x8_matmat_dequant(x, c, cb, scales, None), ls)
                    except Exception as e:  # noqa: BLE001
                        t = float("nan")
                    res[name] = min(res.get(name, 1e9), t)
            for key in ("kx8_ksplit", "kx8_rt"):
                _native.set_tuning(key, 0)
            print(f"2x8g8 {fin}->{fout} B={B}: " + "  ".join(f"{k} {v:.2f}" for k, v in res.items()), flush=True)
    sys.exit(0)

for K in (2,):
    for fin, fout in ((4096, 4096), (4096, 11008), (11008, 4096), (8192, 8192)):
        ls = layers(fin, fout, K, 24)
        scales = torch.ones((fout, 1, 1, 1), device=dev, dtype=torch.float16)
        Ws = [torch.randn((fout, fin), device=dev).half() for _ in range(24)]
        op = hk.code2x8_matmat_dequant if K == 2 else hk.code1x8_matmat_dequant
        for B in (8, 16, 32, 64, 128, 256):
            x = torch.randn((B, fin), device=dev).half()
            hk.USE_FUSED_KX8_MFMA = True
            t_f = timeit(lambda c, cb: op(x, c, cb, scales, None), ls)
            hk.USE_FUSED_KX8_MFMA = False
            t_l = timeit(lambda c, cb: op(x, c, cb, scales, None), ls)
            hk.USE_FUSED_KX8_MFMA = True
            it = iter(range(10**9))
            t_d = timeit(lambda c, cb: torch.nn.functional.linear(x, Ws[next(it) % 24]), ls)
            print(f"{K}x8g8 {fin}->{fout} B={B}: fused {t_f:.2f} us  dequant+gemm {t_l:.2f} us  dense fp16 (rotating weights) {t_d:.2f} us", flush=True)
