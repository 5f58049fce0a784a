"""The oracle (numpy + C restatements) pinned against outputs of the reference's own Python code.

tests/golden/aqlm_ref_golden.npz was produced by oracle/gen_golden.py, which imports
/root/reference/inference_lib/src/aqlm (utils.py:43-70, dequantization.py:9-21) and runs it on seeded inputs.
"""
import hashlib

import numpy as np
import pytest

from oracle import aqlm_oracle as orc
from oracle import c_oracle


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def case_names(golden):
    return sorted({k.split("/")[0] for k in golden if k.endswith("/cfg")})


def regen(golden, name):
    seed, fin, fout, K, nbits, g, batch, bias, ogs = [int(v) for v in golden[f"{name}/cfg"]]
    dt = str(golden[f"{name}/dtype"])
    np_dt = {"float16": np.float16, "float32": np.float32, "bfloat16": "bfloat16"}[dt]
    L = orc.make_layer(seed, fin, fout, K, nbits, g, batch=batch, bias=bool(bias), out_group_size=ogs, float_dtype=np_dt)
    for key in ("codes", "codebooks", "scales", "x", "bias"):
        if L[key] is not None:
            assert _sha(L[key]) == str(golden[f"{name}/sha_{key}"]), f"seeded input {key} differs from the golden run"
    return L, dict(K=K, nbits=nbits, g=g, ogs=ogs, dtype=dt)


def meanrel(a, b):
    return float(np.mean(np.abs(a - b)) / np.mean(np.abs(b)))


CASES = ["c1x16g8_f16", "c1x16g8_f16_nobias", "c1x16g16_f16", "c1x16g8_bf16", "c2x8g8_f16", "c2x8g8_bf16",
         "c1x8g8_f16", "c8x8g32_f16", "c4x8g16_f16", "c2x8g8_og2_f32", "c1x12g8_f32"]


def test_case_list_matches_golden(golden):
    assert case_names(golden) == sorted(CASES)


@pytest.mark.parametrize("nbits", [8, 12, 16])
def test_pack_unpack_kat(golden, nbits):
    vals = golden[f"kat/pack{nbits}_in"]
    packed = orc.pack_int_data(vals, nbits)
    assert packed.dtype == golden[f"kat/pack{nbits}_out"].dtype
    np.testing.assert_array_equal(packed, golden[f"kat/pack{nbits}_out"])
    np.testing.assert_array_equal(orc.unpack_int_data(packed, nbits), golden[f"kat/unpack{nbits}_out"])
    np.testing.assert_array_equal(orc.unpack_int_data(packed, nbits), vals)


@pytest.mark.parametrize("name", CASES)
def test_numpy_oracle_matches_reference(golden, name):
    L, cfg = regen(golden, name)
    # fp64 oracle vs the reference run in fp32: agreement to fp32 round-off
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    assert meanrel(y64, golden[f"{name}/y_ref32"]) < 2e-6
    np.testing.assert_allclose(y64, golden[f"{name}/y_ref32"], rtol=0, atol=2e-4 * np.abs(y64).mean())
    W64 = orc.dequantize_weight(L["codes_unsigned"], L["codebooks"], L["scales"])
    np.testing.assert_allclose(W64, golden[f"{name}/W_ref32"], rtol=1e-5, atol=1e-5)
    gin = orc.dequantize_gemm_transposed(golden[f"{name}/gout"], L["codes"], L["codebooks"], L["scales"], None)
    np.testing.assert_allclose(gin, golden[f"{name}/gin_ref32"], rtol=0, atol=3e-4 * np.abs(gin).mean())
    # the reference's own storage-dtype result is within its documented error of the fp64 oracle
    tol = {"float16": 1e-3, "bfloat16": 8e-3, "float32": 1e-5}[cfg["dtype"]]
    assert meanrel(golden[f"{name}/y_refnat"], y64) < tol


@pytest.mark.parametrize("name", [c for c in CASES if "og2" not in c])
def test_lut_restatement_equals_dequant(golden, name):
    """numba_kernel.py:37-48 restated (numpy + C) == the pinned dequantize_gemm, up to fp32 summation order."""
    L, cfg = regen(golden, name)
    codes_alt = orc.permute_codes_for_lut(L["codes"])
    y_ref = golden[f"{name}/y_ref32"]
    y_np = orc.lut_gemm(L["x"], codes_alt, L["codebooks"], L["scales"], L["bias"], acc_dtype=np.float64)
    assert meanrel(y_np, y_ref) < 2e-6
    if cfg["nbits"] in (8, 16):
        k = c_oracle.LutGemv(L["codebooks"], codes_alt, L["scales"], cfg["nbits"], nthreads=2)
        d = c_oracle.DequantGemv(L["codebooks"], L["codes"], L["scales"], L["bias"], cfg["nbits"], nthreads=2)
        for b in range(L["x"].shape[0]):
            y_c = k(L["x"][b]).copy()
            if L["bias"] is not None:
                y_c += L["bias"].astype(np.float32)
            assert meanrel(y_c, y_ref[b]) < 2e-5
            assert meanrel(d(L["x"][b]), y_ref[b]) < 2e-5
        Wc = c_oracle.dequant_weight(L["codebooks"], L["codes"], L["scales"], cfg["nbits"])
        np.testing.assert_allclose(Wc, golden[f"{name}/W_ref32"], rtol=1e-5, atol=1e-5)


def test_algorithmic_bytes_table():
    # SURVEY.md section 8(d) table
    assert orc.algorithmic_bytes(4096, 4096, 1, 16, 8) == 5_267_456
    assert orc.algorithmic_bytes(4096, 11008, 1, 16, 8) == 12_372_992
    assert orc.algorithmic_bytes(4096, 14336, 1, 16, 8) == 15_794_176
    assert orc.algorithmic_bytes(14336, 4096, 1, 16, 8) == 15_773_696
    assert orc.algorithmic_bytes(8192, 28672, 1, 16, 8) == 59_899_904
    assert orc.algorithmic_bytes(4096, 4096, 2, 8, 8) == 4_227_072
    assert orc.algorithmic_bytes(4096, 4096, 8, 8, 32) == 4_349_952
    assert orc.algorithmic_bytes(4096, 4096, 1, 16, 8, batch=128) == 7_348_224
