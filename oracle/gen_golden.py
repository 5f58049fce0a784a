#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python implementation.

Runs only in the authoring container (needs /root/reference; the GPU box does not have it).
The committed .npz files are what travels.  Usage:

    python oracle/gen_golden.py            # writes tests/golden/aqlm_ref_golden.npz

What is recorded, per case (seeded inputs come from oracle.aqlm_oracle.make_layer):
  * sha256 of every input array (so a test can prove it regenerated the same inputs),
  * y_ref32   = reference dequantize_gemm(x, codes, codebooks, scales, bias) with all float
                tensors up-cast to float32 (values are fp16/bf16-representable)        [dequantization.py:9-21]
  * y_refnat  = the same call in the storage dtype (fp16 / bf16) -- the reference's own CPU
                result for that dtype, recorded to document its rounding error
  * W_ref32   = reference _dequantize_weight(unpack_int_data(codes), codebooks, scales) [utils.py:43-70]
  * gin_ref32 = d/dx of sum(F.linear(x, W) * g)  == g @ W  (backward operator)
  * pack/unpack known-answer vectors                                                   [utils.py:23-31]
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/inference_lib/src"
if not os.path.isdir(REF_SRC):
    sys.exit("reference not present; golden vectors can only be regenerated in the authoring container")
# the reference must win over this repo's drop-in `aqlm` alias package
sys.path = [REF_SRC] + [p for p in sys.path if os.path.abspath(p or ".") != os.path.dirname(HERE)]
sys.path.insert(1, HERE)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from aqlm.inference_kernels.dequantization import dequantize_gemm  # noqa: E402  (reference)
from aqlm.utils import _dequantize_weight, pack_int_data, unpack_int_data  # noqa: E402  (reference)

import aqlm  # noqa: E402

assert aqlm.__file__.startswith(REF_SRC), aqlm.__file__
import aqlm_oracle as orc  # noqa: E402

CASES = [
    # name, seed, in, out, K, nbits, g, batch, bias, out_group, dtype
    ("c1x16g8_f16", 11, 512, 96, 1, 16, 8, 3, True, 1, "float16"),
    ("c1x16g8_f16_nobias", 12, 1088, 40, 1, 16, 8, 1, False, 1, "float16"),
    ("c1x16g16_f16", 13, 1024, 64, 1, 16, 16, 2, True, 1, "float16"),
    ("c1x16g8_bf16", 14, 512, 64, 1, 16, 8, 2, True, 1, "bfloat16"),
    ("c2x8g8_f16", 21, 512, 128, 2, 8, 8, 2, True, 1, "float16"),
    ("c2x8g8_bf16", 22, 576, 72, 2, 8, 8, 1, False, 1, "bfloat16"),
    ("c1x8g8_f16", 23, 256, 64, 1, 8, 8, 4, True, 1, "float16"),
    ("c8x8g32_f16", 24, 1024, 64, 8, 8, 32, 2, True, 1, "float16"),
    ("c4x8g16_f16", 25, 512, 48, 4, 8, 16, 1, True, 1, "float16"),
    ("c2x8g8_og2_f32", 26, 128, 32, 2, 8, 8, 2, True, 2, "float32"),
    ("c1x12g8_f32", 27, 256, 24, 1, 12, 8, 1, True, 1, "float32"),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def to_torch(a, dtype):
    if a is None:
        return None
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dtype) if t.is_floating_point() else t


def main():
    out = {}
    torch.manual_seed(0)
    for name, seed, fin, fout, K, nbits, g, batch, bias, ogs, dt in CASES:
        np_dt = {"float16": np.float16, "float32": np.float32, "bfloat16": "bfloat16"}[dt]
        L = orc.make_layer(seed, fin, fout, K, nbits, g, batch=batch, bias=bias, out_group_size=ogs, float_dtype=np_dt)
        tdt = getattr(torch, dt)
        codes = torch.from_numpy(L["codes"])
        for key in ("codes", "codebooks", "scales", "x", "bias"):
            if L[key] is not None:
                out[f"{name}/sha_{key}"] = np.array(sha(L[key]))
        f32 = {k: to_torch(L[k], torch.float32) for k in ("codebooks", "scales", "x", "bias")}
        nat = {k: to_torch(L[k], tdt) for k in ("codebooks", "scales", "x", "bias")}
        with torch.no_grad():
            y32 = dequantize_gemm(f32["x"], codes, f32["codebooks"], f32["scales"], f32["bias"])
            ynat = dequantize_gemm(nat["x"], codes, nat["codebooks"], nat["scales"], nat["bias"])
            W32 = _dequantize_weight(unpack_int_data(codes, nbits), f32["codebooks"], f32["scales"])
        # backward operator through autograd of the reference definition
        rng = np.random.default_rng(seed + 1000)
        gout = rng.standard_normal((batch, fout), dtype=np.float32)
        xg = f32["x"].clone().requires_grad_(True)
        yy = F.linear(xg, _dequantize_weight(unpack_int_data(codes, nbits), f32["codebooks"], f32["scales"]))
        (yy * torch.from_numpy(gout)).sum().backward()
        out[f"{name}/y_ref32"] = y32.numpy()
        out[f"{name}/y_refnat"] = ynat.float().numpy()
        out[f"{name}/W_ref32"] = W32.numpy()
        out[f"{name}/gout"] = gout
        out[f"{name}/gin_ref32"] = xg.grad.numpy()
        out[f"{name}/cfg"] = np.array([seed, fin, fout, K, nbits, g, batch, int(bias), ogs])
        out[f"{name}/dtype"] = np.array(dt)
        print(f"{name}: y {tuple(y32.shape)} |y|~{y32.abs().mean():.3f} nat-vs-32 meanrel "
              f"{((ynat.float() - y32).abs().mean() / y32.abs().mean()).item():.2e}")

    # pack / unpack known answers (utils.py:23-31)
    for nbits in (8, 12, 16):
        vals = np.array([0, 1, 2 ** (nbits - 1) - 1, 2 ** (nbits - 1), 2 ** (nbits - 1) + 1, 2**nbits - 1], dtype=np.int64)
        packed = pack_int_data(torch.from_numpy(vals.copy()), nbits)
        out[f"kat/pack{nbits}_in"] = vals
        out[f"kat/pack{nbits}_out"] = packed.numpy()
        out[f"kat/unpack{nbits}_out"] = unpack_int_data(packed, nbits).numpy()

    dst = os.path.join(os.path.dirname(HERE), "tests", "golden", "aqlm_ref_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
