"""GPU parity: the HIP path (through the C ABI) against the pinned oracle and the reference's golden outputs.

Tolerances (BASELINE.json north_star: "fp16 output within 1e-3 rel"):
  * fp16:  mean|y - y_ref| / mean|y_ref| <= 1e-3 against the fp64 oracle AND against the reference's fp32 run,
           plus a per-element bound  |y - y64| <= 2e-3 * mean|y64| + 4 * ulp_fp16(|y64|)  to catch localised errors;
  * bf16:  same with 8e-3 / 1.6e-2 and bf16 ulps (a bf16 output has 8 mantissa bits: rounding alone is ~2e-3 rel;
           the reference's own bf16 CPU result is 2-3e-3 off, see tests/golden generator log);
  * integer / layout properties (row permutation, batch consistency, zero input): bit-exact.
"""
import os

import numpy as np
import pytest

from oracle import aqlm_oracle as orc
from oracle import c_oracle

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hk():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    from aqlm_amd.inference_kernels import hip_kernel

    return hip_kernel


DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _raw_op_on_the_direct_kernel(request):
    """`hk.code1x16_matmat` packs large layers on its own (transparent prepack cache).  The tests of this file name the
    kernel they exercise: the raw op means the DIRECT kernel here, except in the tests marked `raw_prepack`."""
    if not torch.cuda.is_available():
        yield
        return
    from aqlm_amd.inference_kernels import hip_kernel

    old = hip_kernel.RAW_OP_PREPACK
    hip_kernel.RAW_OP_PREPACK = request.node.get_closest_marker("raw_prepack") is not None
    hip_kernel.clear_raw_op_prepack_cache()
    yield
    hip_kernel.RAW_OP_PREPACK = old
    hip_kernel.clear_raw_op_prepack_cache()


def tdtype(name):
    return {"float16": torch.float16, "bfloat16": torch.bfloat16}[name]


def to_dev(L, dtype):
    """numpy layer dict -> torch tensors on the GPU in `dtype` (values are exactly representable)."""
    f = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dtype).to(DEV)
    return dict(
        codes=torch.from_numpy(L["codes"]).to(DEV),
        codebooks=f(L["codebooks"]),
        scales=f(L["scales"]),
        bias=f(L["bias"]),
        x=f(L["x"]),
    )


def ulp(y, dtype):
    mant = 10 if dtype == torch.float16 else 7
    e = np.floor(np.log2(np.maximum(np.abs(y), 2.0**-14)))
    return 2.0 ** (e - mant)


def check_close(y, y64, dtype, what="", el_scale=1.0):
    y = np.asarray(y, dtype=np.float64)
    y64 = np.asarray(y64, dtype=np.float64)
    assert y.shape == y64.shape, (y.shape, y64.shape)
    assert np.isfinite(y).all(), f"{what}: non-finite output"
    mean_tol, el_tol = (1e-3, 2e-3) if dtype == torch.float16 else (8e-3, 1.6e-2)
    el_tol *= el_scale
    m = np.mean(np.abs(y - y64)) / np.mean(np.abs(y64))
    assert m <= mean_tol, f"{what}: mean-rel {m:.3e} > {mean_tol}"
    bound = el_tol * np.mean(np.abs(y64)) + 4 * ulp(y64, dtype)
    bad = np.abs(y - y64) > bound
    assert not bad.any(), f"{what}: {bad.sum()} elements outside bound, worst {np.abs(y - y64).max():.4g}"
    return m


def check_rounded(y, y64, dtype, what=""):
    """The arithmetic contract of the fused kernels -- exact fp16 / bf16 products, fp32 sums, `acc * scale + bias` in fp32, ONE rounding
    -- stated without a hand-picked tolerance: the output must be the correctly rounded fp64 result, except where the fp32
    accumulation noise (~ 2^-24 of the sum of |terms|, a few 1e-6 of the output scale) moves a value across a rounding boundary.
    So: every element within one unit in the last place of the storage type (+ that noise) of the rounded oracle, and almost all of
    them equal to it bit for bit.  (check_close's bf16 bounds -- 8e-3 mean, 1.6e-2 per element -- only say "one rounding of 8 bits".)"""
    y = np.asarray(y, dtype=np.float64)
    y64 = np.asarray(y64, dtype=np.float64)
    yr = torch.from_numpy(y64).to(dtype).double().numpy()   # round-to-nearest-even into the storage type
    sigma = float(np.sqrt(np.mean(y64 * y64)))
    err = np.abs(y - yr)
    bound = ulp(y64, dtype) + 2e-5 * sigma
    bad = err > bound
    assert not bad.any(), f"{what}: {bad.sum()} elements more than one ulp (+ fp32 noise) from the rounded oracle, worst {err.max():.4g}"
    exact = float(np.mean(y == yr))
    assert exact >= (0.97 if dtype == torch.float16 else 0.995), f"{what}: only {exact:.4f} of the outputs equal the correctly rounded result"
    return exact


def run_forward(hk, K, nbits, g, T, batch_shape=None):
    x = T["x"] if batch_shape is None else T["x"].reshape(*batch_shape, T["x"].shape[-1])
    if (K, nbits) == (1, 16):
        return hk.code1x16_matmat(x, T["codes"], T["codebooks"], T["scales"], T["bias"])
    if nbits == 8:
        return hk.codekx8_matmat(x, T["codes"], T["codebooks"], T["scales"], T["bias"])
    return hk.generic_matmat(x, T["codes"], T["codebooks"], T["scales"], T["bias"])


# ------------------------------------------------------------------ golden fixtures (reference outputs)
GOLDEN = ["c1x16g8_f16", "c1x16g8_f16_nobias", "c1x16g16_f16", "c1x16g8_bf16", "c2x8g8_f16", "c2x8g8_bf16",
          "c1x8g8_f16", "c8x8g32_f16", "c4x8g16_f16"]


@pytest.mark.parametrize("name", GOLDEN)
def test_golden_forward_dequant_backward(hk, golden, name):
    seed, fin, fout, K, nbits, g, batch, bias, ogs = [int(v) for v in golden[f"{name}/cfg"]]
    dt = tdtype(str(golden[f"{name}/dtype"]))
    L = orc.make_layer(seed, fin, fout, K, nbits, g, batch=batch, bias=bool(bias),
                       float_dtype=np.float16 if dt == torch.float16 else "bfloat16")
    T = to_dev(L, dt)
    y = run_forward(hk, K, nbits, g, T).float().cpu().numpy()
    check_close(y, golden[f"{name}/y_ref32"], dt, f"{name} forward vs reference")
    # generic kernel must agree too
    yg = hk.generic_matmat(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"]).float().cpu().numpy()
    check_close(yg, golden[f"{name}/y_ref32"], dt, f"{name} generic vs reference")
    # dequantised weight
    W = (hk.code1x16_dequant if nbits == 16 else hk.code2x8_dequant)(T["codes"], T["codebooks"], T["scales"])
    Wref = golden[f"{name}/W_ref32"]
    rt = 2.0**-10 if dt == torch.float16 else 2.0**-7
    np.testing.assert_allclose(W.float().cpu().numpy(), Wref, rtol=rt, atol=1e-6)
    # backward operator
    gout = torch.from_numpy(golden[f"{name}/gout"]).to(dt).to(DEV)
    fn = hk.code1x16_matmat_dequant_transposed if nbits == 16 else hk.code2x8_matmat_dequant_transposed
    gin = fn(gout, T["codes"], T["codebooks"], T["scales"], None).float().cpu().numpy()
    gref = orc.dequantize_gemm_transposed(gout.float().cpu().numpy(), L["codes"], L["codebooks"], L["scales"], None)
    check_close(gin, gref, dt, f"{name} backward vs oracle")


# ------------------------------------------------------------------ scheme / shape sweep vs the oracle
SWEEP = [
    # K, nbits, g, in, out, batch, bias, dtype
    (1, 16, 8, 4096, 256, 1, True, "float16"),
    (1, 16, 8, 4096, 250, 3, False, "float16"),       # out not a multiple of rows-per-block
    (1, 16, 8, 11008, 96, 1, True, "float16"),        # 1376 groups: 172 units, ragged last iteration
    (1, 16, 8, 14336, 64, 2, True, "bfloat16"),
    (1, 16, 8, 520, 64, 1, True, "float16"),          # 65 groups: not a multiple of 8 -> generic route
    (1, 16, 16, 4096, 128, 5, True, "float16"),
    (1, 16, 16, 2048, 128, 8, True, "bfloat16"),
    (2, 8, 8, 4096, 256, 1, True, "float16"),
    (2, 8, 8, 11008, 100, 7, False, "float16"),
    (2, 8, 8, 4096, 128, 4, True, "bfloat16"),
    (1, 8, 8, 4096, 256, 6, True, "float16"),
    (1, 8, 8, 1000, 64, 1, True, "float16"),          # 125 groups -> generic route
    (8, 8, 32, 4096, 256, 1, True, "float16"),
    (8, 8, 32, 4096, 72, 2, True, "bfloat16"),
    (8, 8, 32, 8192, 64, 3, False, "float16"),
    (4, 8, 8, 1024, 64, 2, True, "float16"),          # no tuned instance -> generic route via kx8 entry
    (2, 8, 16, 1024, 64, 1, True, "float16"),
    (3, 5, 8, 512, 40, 2, True, "float16"),           # odd nbits in int8 containers (generic)
    (1, 12, 8, 512, 40, 2, True, "float16"),          # 12-bit codes in int16 containers (generic)
]


@pytest.mark.parametrize("K,nbits,g,fin,fout,batch,bias,dt", SWEEP)
def test_sweep_vs_oracle(hk, K, nbits, g, fin, fout, batch, bias, dt):
    dtype = tdtype(dt)
    L = orc.make_layer(1000 + fin + fout + K, fin, fout, K, nbits, g, batch=batch, bias=bias,
                       float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    T = to_dev(L, dtype)
    y = run_forward(hk, K, nbits, g, T).float().cpu().numpy()
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    check_close(y, y64, dtype, f"{K}x{nbits}g{g} {fin}->{fout} b{batch}")


def test_leading_dims_and_noncontiguous_input(hk):
    L = orc.make_layer(5, 1024, 64, 1, 16, 8, batch=6, bias=True)
    T = to_dev(L, torch.float16)
    y = run_forward(hk, 1, 16, 8, T, batch_shape=(2, 3))
    assert y.shape == (2, 3, 64)
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"]).reshape(2, 3, 64)
    check_close(y.float().cpu().numpy(), y64, torch.float16, "leading dims")
    wide = torch.zeros(6, 2048, dtype=torch.float16, device=DEV)
    wide[:, ::2] = T["x"]
    y2 = hk.code1x16_matmat(wide[:, ::2], T["codes"], T["codebooks"], T["scales"], T["bias"])
    assert torch.equal(y2, y.reshape(6, 64))


def test_error_conventions(hk):
    L = orc.make_layer(6, 512, 32, 1, 16, 8, batch=1, bias=False)
    T = to_dev(L, torch.float16)
    with pytest.raises(NotImplementedError, match="only support float16 and bfloat16"):
        hk.code1x16_matmat(T["x"].float(), T["codes"], T["codebooks"].float(), T["scales"].float(), None)
    with pytest.raises(NotImplementedError):
        hk.code2x8_matmat(T["x"], T["codes"], T["codebooks"], T["scales"], None)  # wrong codebook shape
    bad_cb = torch.zeros(1, 65536, 1, 4, dtype=torch.float16, device=DEV)
    with pytest.raises(NotImplementedError, match="8 or 16"):
        hk.code1x16_matmat(T["x"][:, :256], T["codes"], bad_cb, T["scales"], None)
    with pytest.raises(ValueError):
        hk.code1x16_matmat(T["x"][:, :256], T["codes"], T["codebooks"], T["scales"], None)


# ------------------------------------------------------------------ full-size (BASELINE shapes) parity + properties
FULL = [
    (1, 16, 8, 4096, 4096, "float16"),
    (1, 16, 8, 4096, 11008, "float16"),
    (1, 16, 8, 14336, 4096, "bfloat16"),
    (2, 8, 8, 4096, 11008, "float16"),
    (8, 8, 32, 4096, 4096, "float16"),
]


@pytest.mark.parametrize("K,nbits,g,fin,fout,dt", FULL)
def test_full_size_vs_c_oracle_and_properties(hk, K, nbits, g, fin, fout, dt):
    dtype = tdtype(dt)
    L = orc.make_layer(77, fin, fout, K, nbits, g, batch=4, bias=True,
                       float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    T = to_dev(L, dtype)
    y = run_forward(hk, K, nbits, g, T)
    # (1) against the C restatement of dequantize_gemm (pinned through the numpy oracle)
    ref = c_oracle.DequantGemv(L["codebooks"], L["codes"], L["scales"], L["bias"], nbits, nthreads=0)
    yh = y.float().cpu().numpy()
    for b in range(4):
        check_close(yh[b], ref(L["x"][b]).copy(), dtype, f"full {K}x{nbits}g{g} {fin}->{fout} row {b}")
    # ... and against the fp64 oracle as a ROUNDING statement (no tolerance of our own choosing): the kernels' outputs are the correctly
    # rounded result up to fp32 accumulation noise -- fp16 and bf16 alike
    if fin * fout <= 1 << 26:
        y64f = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
        check_rounded(yh, y64f, dtype, f"full {K}x{nbits}g{g} {fin}->{fout}")
    # (2) batch consistency: a row of the batched launch == the same row launched alone (bit-exact)
    T1 = dict(T, x=T["x"][2:3].contiguous())
    y_single = run_forward(hk, K, nbits, g, T1)[0]
    if K == 8:  # 1..8 rows of 8x8 schemes take the look-up-table kernel (round 5: one launch of rows x the single-row workgroups)
        if g == 32 and fin % 256 == 0 and fin >= 2048:
            # ... except 8x8 g32 from 3 rows on: the fused MFMA kernel (same exact products and fp32 sums in another order than the
            # table kernel of a single row; within the fused kernel a row's bits depend neither on its neighbours nor on their number)
            check_close(y_single.float().cpu().numpy(), y[2].double().cpu().numpy(), dtype, "one row (table kernel) vs row of four (fused MFMA)")
            y3 = run_forward(hk, K, nbits, g, dict(T, x=T["x"][1:4].contiguous()))
            assert torch.equal(y3[1], y[2])
        else:
            assert torch.equal(y_single, y[2])
        with_gather = hk._gemv(T1["x"], T["codes"], T["codebooks"], T["scales"], T["bias"], "kx8")[0]  # same maths, different summation order
        check_close(with_gather.float().cpu().numpy(), y[2].float().cpu().numpy().astype(np.float64), dtype, "lut vs LDS-gather kernel")
    elif nbits == 8:
        # single rows of big 1x8 / 2x8 layers take the replicated-LDS kernel (another reduction tree): close to the
        # batched result, and the plain LDS kernel (forced) reproduces the batched row bit for bit
        from aqlm_amd import _native

        # (and batches of 2+ rows take the fused MFMA kernel: same exact products, another summation order)
        check_close(y_single.float().cpu().numpy(), y[2].float().cpu().numpy().astype(np.float64), dtype, "replicated vs batched")
        _native.set_tuning("kx8_replicas", 0)
        _native.set_tuning("kx8_mfma_min_rows", 0)
        try:
            y_plain = run_forward(hk, K, nbits, g, T)
            assert torch.equal(run_forward(hk, K, nbits, g, T1)[0], y_plain[2])
            check_close(y.float().cpu().numpy(), y_plain.float().cpu().numpy().astype(np.float64), dtype, "mfma vs plain matvec kernel, batched")
        finally:
            _native.set_tuning("kx8_replicas", 1)
            _native.set_tuning("kx8_mfma_min_rows", 2)
    else:
        assert torch.equal(y_single, y[2])
    # (3) zero input -> exactly the bias
    Tz = dict(T, x=torch.zeros_like(T["x"][:1]))
    assert torch.equal(run_forward(hk, K, nbits, g, Tz)[0], T["bias"])
    # (4) permuting output rows of (codes, scales, bias) permutes y bit-exactly
    perm = torch.randperm(fout, generator=torch.Generator().manual_seed(3)).to(DEV)
    Tp = dict(T, codes=T["codes"][perm].contiguous(), scales=T["scales"][perm].contiguous(), bias=T["bias"][perm].contiguous())
    assert torch.equal(run_forward(hk, K, nbits, g, Tp), y[:, perm])
    # (5) the dequantised weight reproduces the matvec:  y ~= x @ W^T + bias
    W = (hk.code1x16_dequant if nbits == 16 else hk.code2x8_dequant)(T["codes"], T["codebooks"], T["scales"])
    # (W is rounded to the storage dtype, so this is a looser, mean-only comparison of two approximations)
    y_w = (T["x"].double() @ W.double().T + T["bias"].double()).cpu().numpy()
    m = np.mean(np.abs(yh - y_w)) / np.mean(np.abs(y_w))
    assert m <= (2e-3 if dtype == torch.float16 else 1.2e-2), f"gemv vs dequant@x mean-rel {m:.3e}"


# ------------------------------------------------------------------ large-batch (MFMA) op
@pytest.mark.parametrize("g,fin,fout,B,dt", [
    (8, 512, 128, 32, "float16"),
    (8, 4096, 4096, 128, "float16"),
    (8, 4096, 1000, 100, "float16"),     # ragged rows and batch
    (8, 11008, 256, 7, "bfloat16"),
    (16, 4096, 512, 128, "float16"),
    (8, 1024, 384, 300, "float16"),      # > FUSED_MFMA_MAX_ROWS rows: dequant + library GEMM route (and, forced, slabs)
    (8, 520, 64, 16, "float16"),         # in % 64 != 0 -> dequant + F.linear route
])
def test_matmat_dequant_mfma(hk, g, fin, fout, B, dt):
    dtype = tdtype(dt)
    L = orc.make_layer(4242 + B, fin, fout, 1, 16, g, batch=B, bias=True,
                       float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    T = to_dev(L, dtype)
    y = hk.code1x16_matmat_dequant(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"]).float().cpu().numpy()
    W64 = orc.dequantize_weight(L["codes_unsigned"], L["codebooks"], L["scales"])
    y64 = L["x"].astype(np.float64) @ W64.T + L["bias"].astype(np.float64)
    check_close(y, y64, dtype, f"mfma g{g} {fin}->{fout} B{B}")
    if B > hk.FUSED_MFMA_MAX_ROWS:   # the fused kernel's 128-column slabs stay covered through the same op
        old, hk.FUSED_MFMA_MAX_ROWS = hk.FUSED_MFMA_MAX_ROWS, 1 << 30
        try:
            y2 = hk.code1x16_matmat_dequant(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"]).float().cpu().numpy()
        finally:
            hk.FUSED_MFMA_MAX_ROWS = old
        check_close(y2, y64, dtype, f"mfma slabs g{g} {fin}->{fout} B{B}")


def test_matmat_dequant_mfma_kernel_variants(hk):
    """The large-batch 1x16 op has three kernels behind one entry (tuning knob `gemm_variant`): the K-split LDS-DMA pipeline (3),
    the 16-row no-split kernel of round 4 (2; the default picks it by batch and layer size) and the register-staged kernel of
    round 1 (1).  Each must agree with the oracle on every shape it takes -- ragged row counts, batches that are not multiples
    of 16, both group sizes, K that is / is not a multiple of the 16-row kernel's step, K too short for its rings (falls through
    to the pipeline) -- and with the others to fp32 round-off."""
    from aqlm_amd import _native

    shapes = [(8, 4096, 1000, 100, "float16"), (16, 1024, 256, 20, "bfloat16"), (8, 512, 48, 128, "float16"),
              (8, 11008, 300, 33, "float16"), (16, 2240, 77, 7, "float16"), (8, 1792, 64, 16, "bfloat16"),
              (16, 4096, 128, 64, "float16"), (8, 4096, 4096, 9, "float16"), (8, 960, 40, 48, "float16")]
    for g, fin, fout, B, dt in shapes:
        dtype = tdtype(dt)
        L = orc.make_layer(5150 + B + fout, fin, fout, 1, 16, g, batch=B, bias=True,
                           float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
        T = to_dev(L, dtype)
        y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
        ys = {}
        for variant in (0, 1, 2, 3):
            _native.set_tuning("gemm_variant", variant)
            try:
                ys[variant] = hk.code1x16_matmat_dequant(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
                assert torch.equal(ys[variant], hk.code1x16_matmat_dequant(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"]))
            finally:
                _native.set_tuning("gemm_variant", 0)
            check_close(ys[variant].float().cpu().numpy(), y64, dtype, f"mfma variant {variant} g{g} {fin}->{fout} B{B}")
        # a strided input (rows of a wider tensor) through the default choice
        wide = torch.zeros(B, fin + 64, dtype=dtype, device=DEV)
        wide[:, 32:32 + fin] = T["x"]
        assert torch.equal(hk.code1x16_matmat_dequant(wide[:, 32:32 + fin], T["codes"], T["codebooks"], T["scales"], T["bias"]), ys[0])


@pytest.mark.parametrize("g,fin,fout,B,dt", [
    (8, 192, 16, 1, "float16"),          # the shortest K the pipeline takes (3 chunks), one row tile, one column
    (8, 256, 129, 17, "float16"),        # 4 chunks: prologue + one main iteration + tail; ragged row block
    (8, 4096, 130, 33, "bfloat16"),      # out % 4 != 0: scalar partial / Y stores, K split
    (16, 2048, 2048, 64, "float16"),     # g = 16: half-entry fragments, 4-byte code DMA
    (8, 14336, 4096, 128, "float16"),    # uneven K slices (224 chunks over 8 blocks = 28 each), full tile
    (8, 5120, 13824, 96, "float16"),     # more row blocks than one round of the chip; 6 batch tiles
    (8, 2048, 28672, 48, "float16"),     # no K split: the block writes Y itself
])
def test_matmat_dequant_mfma_pipeline_shapes(hk, g, fin, fout, B, dt):
    """Edge cases of the LDS-DMA pipeline (gemm_1x16_glds_kernel): shortest K, ragged rows / batch, unaligned out,
    both group sizes, uneven K slices, direct epilogue."""
    dtype = tdtype(dt)
    L = orc.make_layer(9000 + B + fout, fin, fout, 1, 16, g, batch=B, bias=True,
                       float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    T = to_dev(L, dtype)
    if fin * fout <= 1 << 24:
        y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    else:  # large layers: W from the C restatement (fp32, exact sums of fp16 entries times an fp16 scale), product in fp64
        W = c_oracle.dequant_weight(L["codebooks"], L["codes"], L["scales"], 16)
        y64 = L["x"].astype(np.float64) @ W.T.astype(np.float64) + L["bias"].astype(np.float64)
    y = hk.code1x16_matmat_dequant(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"]).float().cpu().numpy()
    check_close(y, y64, dtype, f"mfma pipeline g{g} {fin}->{fout} B{B}")


@pytest.mark.parametrize("K,g", [(2, 8), (1, 8), (8, 32)])
def test_matmat_dequant_kx8(hk, K, g):
    L = orc.make_layer(99, 2048, 192, K, 8, g, batch=40, bias=True)
    T = to_dev(L, torch.float16)
    y = hk.code2x8_matmat_dequant(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"]).float().cpu().numpy()
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    check_close(y, y64, torch.float16, f"{K}x8 dequant+gemm")


@pytest.mark.parametrize("K,fin,fout,B,dt,bias", [
    (2, 4096, 4096, 128, "float16", False),   # the headline large-batch shape of the scheme
    (2, 4096, 300, 7, "float16", True),       # the smallest batch the module sends (gemv rule + 1), ragged rows
    (2, 11008, 512, 33, "bfloat16", True),    # K = 172 chunks (a multiple of 4), three batch tiles computed as four
    (1, 2048, 192, 40, "float16", True),      # one codebook: W is exact
    (2, 384, 64, 16, "float16", False),       # the shortest K the kernel takes (3 steps of 2 chunks)
    (2, 640, 48, 100, "bfloat16", True),      # K = 10 chunks: steps of 2 chunks only
    (1, 1280, 77, 200, "float16", True),      # two slabs of 128 rows, ragged rows
    (2, 1024, 16, 256, "float16", False),     # two full slabs, a single row block
    (2, 1024, 8200, 40, "float16", True),     # tall layer: two 16-row tiles per block, the last block half empty
    (1, 512, 11008, 128, "bfloat16", False),  # tall layer, full batch tiles, bf16 sums
])
def test_matmat_dequant_kx8_fused_mfma(hk, K, fin, fout, B, dt, bias):
    """The 8-bit schemes' large-batch ops on the fused dequant -> MFMA kernel (aqlm_hip_gemm_kx8_mfma, round 4): against the fp64
    oracle, against the dequantise + library-GEMM route of the same op (the reference's pipeline, cuda_kernel.cpp:450-484), bit
    for bit repeatable, batch rows independent of their neighbours, strided inputs; shapes the kernel does not take fall through."""
    dtype = tdtype(dt)
    L = orc.make_layer(8800 + fin + B, fin, fout, K, 8, 8, batch=B, bias=bias, float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    T = to_dev(L, dtype)
    op = hk.code2x8_matmat_dequant if K == 2 else hk.code1x8_matmat_dequant
    assert hk._fused_kx8_mfma(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"], hk._dtype_id(T["x"])) is not None
    y = op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    assert torch.equal(y, op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"]))
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    check_close(y.float().cpu().numpy(), y64, dtype, f"fused {K}x8 mfma {fin}->{fout} B{B}")   # exact products, fp32 sums: the strict bound
    hk.USE_FUSED_KX8_MFMA = False
    try:
        y_lib = op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    finally:
        hk.USE_FUSED_KX8_MFMA = True
    den = float(y_lib.float().abs().mean())
    assert float((y.float() - y_lib.float()).abs().mean()) / den < (2e-3 if dtype == torch.float16 else 1e-2)
    if B >= 49:  # the K-split form (round 5: two row tiles per block, 2 / 4 K slices, fp32 partials + finalize) against the one-launch form
        hk.USE_FUSED_KX8_KSPLIT = False
        try:
            y_one = op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
        finally:
            hk.USE_FUSED_KX8_KSPLIT = True
        check_close(y_one.float().cpu().numpy(), y64, dtype, f"fused {K}x8 mfma, no K split {fin}->{fout} B{B}")
        check_close(y.float().cpu().numpy(), y_one.double().cpu().numpy(), dtype, "K-split vs one-launch form")
        from aqlm_amd import _native
        for ks, rt in ((2, 1), (4, 2)):   # forced plans (a plan that does not divide the steps falls back to fewer slices)
            _native.set_tuning("kx8_ksplit", ks)
            _native.set_tuning("kx8_rt", rt)
            try:
                y_f = op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
            finally:
                _native.set_tuning("kx8_ksplit", 0)
                _native.set_tuning("kx8_rt", 0)
            check_close(y_f.float().cpu().numpy(), y64, dtype, f"fused {K}x8 mfma, {ks} K slices x {rt} tiles")
    # a row's result does not depend on the rows around it (same kernel instance: same batch-tile count)
    if B >= 3:
        x2 = T["x"].clone()
        x2[1:] = torch.flip(x2[1:], dims=(0,))
        y2 = op(x2, T["codes"], T["codebooks"], T["scales"], T["bias"])
        assert torch.equal(y2[0], y[0]) and torch.equal(y2[1], y[B - 1])
    wide_x = torch.zeros(B, fin + 32, dtype=dtype, device=DEV)
    wide_x[:, 16:16 + fin] = T["x"]
    assert torch.equal(op(wide_x[:, 16:16 + fin], T["codes"], T["codebooks"], T["scales"], T["bias"]), y)
    # outside the kernel: in_features not a multiple of 128 -> dequantise + GEMM, same answer within the bound
    if fin == 2048:
        L3 = orc.make_layer(17, 448, 40, K, 8, 8, batch=9, bias=True)
        T3 = to_dev(L3, torch.float16)
        assert hk._fused_kx8_mfma(T3["x"], T3["codes"], T3["codebooks"], T3["scales"], T3["bias"], hk._dtype_id(T3["x"])) is None
        check_close(op(T3["x"], T3["codes"], T3["codebooks"], T3["scales"], T3["bias"]).float().cpu().numpy(),
                    orc.dequantize_gemm(L3["x"], L3["codes"], L3["codebooks"], L3["scales"], L3["bias"]), torch.float16, "kx8 fall-through")


@pytest.mark.parametrize("K,fin,fout,dt,bias", [
    (2, 4096, 4096, "float16", True),      # one tile per CU
    (2, 4096, 11008, "float16", False),    # 688 tiles on 256 CUs: workgroups walk 2-3 tiles, the reduction buffers alternate
    (1, 4096, 1000, "bfloat16", True),     # one codebook, ragged last tile
    (2, 11008, 640, "float16", True),      # 86 quads (not a multiple of the 8 waves); 6 rows are 132 KiB of X: the largest image that fits --
                                           # 7+ rows run the kernel in PHASES (round 5): 16 rows = 344 KiB of X in three pieces
    (2, 384, 40, "float16", True),         # 3 quads: five of the eight waves only take part in the barriers
    (2, 14336, 4096, "float16", False),    # Llama-3-8B down projection: 5+ rows in phases (16 rows: 448 KiB in four), one tile per workgroup
    (1, 8192, 6000, "bfloat16", True),     # 375 tiles: two tiles per workgroup in the phased form, the last workgroups hold one; 10+ rows phased
    (2, 4096, 14336, "float16", True),     # 896 tiles: from 7 rows on the phased form with FOUR tiles per workgroup (2..6 rows: single phase)
])
def test_fused_kx8_x_resident_small_batches(hk, K, fin, fout, dt, bias):
    """The fused K x 8 MFMA op at <= 16 rows (round 5: X resident in LDS, aqlm_hip_gemm_kx8_mfma / the 3+ row route of
    aqlm_hip_gemv_kx8): fp64 oracle, the streaming 16-row kernel (tuning key kx8_xres = 0: same exact products, another summation
    order), and batch invariance -- a row's bits depend neither on the other rows nor on how many there are (2..16), which the
    streaming kernel behind round 4's 3-row switch did not give (VERDICT r04 weak #1a)."""
    from aqlm_amd import _native

    dtype = tdtype(dt)
    L = orc.make_layer(9100 + fin + fout, fin, fout, K, 8, 8, batch=32, bias=bias, float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    T = to_dev(L, dtype)
    op = hk.code2x8_matmat_dequant if K == 2 else hk.code1x8_matmat_dequant
    raw = hk.code2x8_matmat if K == 2 else hk.code1x8_matmat
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    rows_max = 16   # (what does not fit the LDS at once runs in phases: same deal of the quads to the waves, same bits)
    y_full = op(T["x"][:rows_max], T["codes"], T["codebooks"], T["scales"], T["bias"])
    check_close(y_full.float().cpu().numpy(), y64[:rows_max], dtype, f"x-resident {K}x8 {fin}->{fout}, {rows_max} rows")
    check_rounded(y_full.float().cpu().numpy(), y64[:rows_max], dtype, f"x-resident {K}x8 {fin}->{fout}, {rows_max} rows")
    assert torch.equal(y_full, op(T["x"][:rows_max], T["codes"], T["codebooks"], T["scales"], T["bias"]))
    for B in (2, 3, 4, 5, 6, 7, 8, 10, 13, 16):   # across the single-phase / phased boundary of the big-K layers too
        if B > rows_max:
            continue
        yb = op(T["x"][:B], T["codes"], T["codebooks"], T["scales"], T["bias"])
        assert torch.equal(yb, y_full[:B]), f"rows of a {B}-row call differ from the same rows of a {rows_max}-row call"
        if B <= 8:   # the decode route: aqlm_hip_gemv_kx8 hands 2+ rows to the same kernel
            assert torch.equal(raw(T["x"][:B], T["codes"], T["codebooks"], T["scales"], T["bias"]), yb)
    # 17 .. 32 rows (round 5): the phased form with two batch tiles -- the same arithmetic per row, so the first 16 rows repeat bit for bit
    if fin >= 1024 and (fout + 15) // 16 <= 768:   # (more than three tiles per CU: the streaming kernel, another summation order)
        y32 = op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
        check_close(y32.float().cpu().numpy(), y64, dtype, f"x-resident {K}x8 {fin}->{fout}, 32 rows")
        assert torch.equal(y32[:16], y_full)
        for B in (17, 24, 31):
            assert torch.equal(op(T["x"][:B], T["codes"], T["codebooks"], T["scales"], T["bias"]), y32[:B]), f"{B}-row call differs from the 32-row call"
    x2 = T["x"][:rows_max].clone()
    x2[1:] = torch.flip(x2[1:], dims=(0,))
    y2 = op(x2, T["codes"], T["codebooks"], T["scales"], T["bias"])
    assert torch.equal(y2[0], y_full[0]) and torch.equal(y2[1], y_full[rows_max - 1])
    wide_x = torch.zeros(rows_max, fin + 32, dtype=dtype, device=DEV)
    wide_x[:, 16:16 + fin] = T["x"][:rows_max]
    assert torch.equal(op(wide_x[:, 16:16 + fin], T["codes"], T["codebooks"], T["scales"], T["bias"]), y_full)
    _native.set_tuning("kx8_xres", 0)
    try:
        y_stream = op(T["x"][:rows_max], T["codes"], T["codebooks"], T["scales"], T["bias"])
    finally:
        _native.set_tuning("kx8_xres", 1)
    check_close(y_full.float().cpu().numpy(), y_stream.double().cpu().numpy(), dtype, "x-resident vs streaming kernel")
    # a NaN in one row of x poisons that row only
    xn = T["x"][:4].clone()
    xn[2, 5] = float("nan")
    yn = op(xn, T["codes"], T["codebooks"], T["scales"], T["bias"])
    assert torch.isnan(yn[2]).all() and torch.equal(yn[0], y_full[0]) and torch.equal(yn[3], y_full[3])


@pytest.mark.parametrize("fin,fout,rows,dt,bias", [
    (4096, 4096, 16, "float16", True),      # one tile per CU, the full first batch tile
    (4096, 11008, 6, "float16", False),     # 688 tiles on 256 CUs: workgroups walk 2-3 tiles; the module's largest decode call
    (11008, 640, 33, "bfloat16", True),     # 43 chunks of 8 groups (not a multiple of the 8 waves); three batch tiles computed as four
    (2048, 200, 64, "float16", True),       # the shortest K the kernel takes (one chunk per wave), ragged last tile, a full slab
    (4096, 1000, 100, "float16", False),    # two slabs of 64 rows (the second: 36 rows = three batch tiles computed as four)
    (2304, 72, 3, "bfloat16", True),        # 9 chunks: wave 0 takes two, the others one
])
def test_fused_8x8g32_mfma(hk, fin, fout, rows, dt, bias, monkeypatch):
    """8x8 g32 at 3+ rows on the fused dequant -> MFMA kernel (aqlm_hip_gemm_8x8_mfma, round 5; VERDICT r04 missing #4): fp64
    oracle at the strict bound (exact products, fp32 sums), the look-up-table route (same terms, another order), bit-exact
    repeatability and batch invariance (a row's bits depend neither on its neighbours nor on the number of rows), strided inputs,
    the raw op / the large-batch op / the module all arriving at the same kernel, NaN rows, shapes outside the kernel."""
    from aqlm_amd import QuantizedLinear
    monkeypatch.setattr(hk, "FUSED_8X8_MFMA_MAX_ROWS", 128)  # the operator stops at one slab of 64 rows (two cost what dequantise + GEMM costs); the entry takes any number
    dtype = tdtype(dt)
    L = orc.make_layer(9900 + fin + rows, fin, fout, 8, 8, 32, batch=rows, bias=bias, float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    T = to_dev(L, dtype)
    args = (T["codes"], T["codebooks"], T["scales"], T["bias"])
    y = hk._fused_8x8_mfma(T["x"], *args, hk._dtype_id(T["x"]))
    assert y is not None
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    check_close(y.float().cpu().numpy(), y64, dtype, f"fused 8x8g32 mfma {fin}->{fout} rows {rows}")
    check_rounded(y.float().cpu().numpy(), y64, dtype, f"fused 8x8g32 mfma {fin}->{fout} rows {rows}")
    assert torch.equal(y, hk._fused_8x8_mfma(T["x"], *args, hk._dtype_id(T["x"])))
    # the ops that lead to it: the large-batch op always, the decode op from FUSED_8X8_MFMA_MIN_ROWS rows on
    assert torch.equal(hk.code2x8_matmat_dequant(T["x"], *args), y)
    assert torch.equal(hk.codekx8_matmat(T["x"][:8], *args), y[:8])
    # batch invariance across row counts and batch-tile counts
    for B in (1, 2, 3, 5, 16, 17, 40):
        if B < rows:
            assert torch.equal(hk._fused_8x8_mfma(T["x"][:B], *args, hk._dtype_id(T["x"])), y[:B]), f"{B}-row call differs"
    x2 = T["x"].clone()
    x2[1:] = torch.flip(x2[1:], dims=(0,))
    y2 = hk._fused_8x8_mfma(x2, *args, hk._dtype_id(x2))
    assert torch.equal(y2[0], y[0]) and torch.equal(y2[1], y[rows - 1])
    wide_x = torch.zeros(rows, fin + 32, dtype=dtype, device=DEV)
    wide_x[:, 16:16 + fin] = T["x"]
    assert torch.equal(hk._fused_8x8_mfma(wide_x[:, 16:16 + fin], *args, hk._dtype_id(wide_x)), y)
    # the look-up-table route on the same rows
    hk.USE_FUSED_8X8_MFMA = False
    try:
        y_lut = hk.codekx8_matmat(T["x"][:min(rows, 6)], *args)
    finally:
        hk.USE_FUSED_8X8_MFMA = True
    check_close(y[:min(rows, 6)].float().cpu().numpy(), y_lut.double().cpu().numpy(), dtype, "fused 8x8 vs look-up-table route")
    # a NaN in one row of x poisons that row only
    xn = T["x"][:3].clone()
    xn[1, 7] = float("nan")
    yn = hk._fused_8x8_mfma(xn, *args, hk._dtype_id(xn))
    assert torch.isnan(yn[1]).all() and torch.equal(yn[0], y[0]) and torch.equal(yn[2], y[2])
    # the module: planar codes for 1-2 rows, this kernel from 3 rows on (compiled fast lane included), hipGraph capture
    if fin == 4096 and fout == 4096:
        m = QuantizedLinear(fin, fout, 32, 1, 8, 8, bias=bias, device=DEV, dtype=dtype)
        with torch.no_grad():
            m.codes.copy_(T["codes"]); m.codebooks.copy_(T["codebooks"]); m.scales.copy_(T["scales"].reshape(m.scales.shape))
            if bias:
                m.bias.copy_(T["bias"])
        with torch.no_grad():
            assert torch.equal(m(T["x"][:4]), y[:4]) and torch.equal(m(T["x"][:6].reshape(2, 3, fin)).reshape(6, fout), y[:6])
            check_close(m(T["x"][:2]).float().cpu().numpy(), y64[:2], dtype, "module, 2 rows (table kernel)")
            assert torch.equal(m(T["x"]), y)
            s = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                yg = m(T["x"][:5])
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(yg, y[:5])
            m.drop_canonical_codes()   # the planar copy only: every decode call stays on the table kernel
            check_close(m(T["x"][:4]).float().cpu().numpy(), y64[:4], dtype, "module without canonical codes, 4 rows")
    # outside the kernel: K below 2048, other group sizes, in_features not a multiple of 256 -> None, and the ops still answer
    if fin == 2048:
        L3 = orc.make_layer(23, 1024, 40, 8, 8, 32, batch=9, bias=True)
        T3 = to_dev(L3, torch.float16)
        a3 = (T3["codes"], T3["codebooks"], T3["scales"], T3["bias"])
        assert hk._fused_8x8_mfma(T3["x"], *a3, hk._dtype_id(T3["x"])) is None
        check_close(hk.code2x8_matmat_dequant(T3["x"], *a3).float().cpu().numpy(),
                    orc.dequantize_gemm(L3["x"], L3["codes"], L3["codebooks"], L3["scales"], L3["bias"]), torch.float16, "8x8 fall-through")
        L4 = orc.make_layer(24, 2048, 40, 8, 8, 8, batch=4, bias=False)
        T4 = to_dev(L4, torch.float16)
        assert hk._fused_8x8_mfma(T4["x"], T4["codes"], T4["codebooks"], T4["scales"], None, hk._dtype_id(T4["x"])) is None


def test_8x8g32_shared_input_group_steps_aside_for_the_fused_kernel(hk):
    """A shared-input group of 8x8 g32 layers serves 1-2 rows through the table kernel (one launch for all members); from the row
    count at which a member's own fused MFMA kernel is faster, the group must not take the call (round 5: measured in the Hugging
    Face decode loop at 4 rows, 389 tokens/s member by member vs 314 through the group)."""
    import aqlm

    fin = 4096
    holder = torch.nn.Module()
    Ls = {}
    for k, (n, fo) in enumerate([("gate_proj", 512), ("up_proj", 768)]):
        Ls[n] = orc.make_layer(4900 + k, fin, fo, 8, 8, 32, batch=5, bias=False)
        m, _ = _module_from(Ls[n], 8, 8, 32, fin, fo, torch.float16)
        setattr(holder, n, m)
    x = to_dev(Ls["gate_proj"], torch.float16)["x"]
    with torch.no_grad():
        alone = {n: getattr(holder, n)(x) for n in Ls}
        alone1 = {n: getattr(holder, n)(x[:1]) for n in Ls}
        groups = aqlm.fuse_shared_input_linears(holder)
        assert len(groups) == 1
        g = groups[0]
        assert g.applicable(x[:1]) and g.applicable(x[:2]) and not g.applicable(x)
        for n in Ls:
            assert torch.equal(getattr(holder, n)(x), alone[n])
        assert g.launches == 0
        x1 = x[:1]   # (the group recognises its siblings' calls by the identity of the input tensor)
        for n in Ls:
            assert torch.equal(getattr(holder, n)(x1), alone1[n])
        assert g.launches == 1 and g.served == 1
        aqlm.unfuse_shared_input_linears(holder)


def test_kx8_mfma_route_inside_hipgraph(hk):
    """3+ row calls of the 8-bit ops (fused MFMA kernel behind aqlm_hip_gemv_kx8) and the large-batch op are captured and replayed
    like every other entry: no allocation, no synchronisation, same result as the eager call."""
    for K, rows, fin, fout in ((2, 4, 4096, 512), (1, 8, 1024, 300), (2, 40, 2048, 192)):
        L = orc.make_layer(640 + rows, fin, fout, K, 8, 8, batch=rows, bias=True)
        T = to_dev(L, torch.float16)
        op = (torch.ops.aqlm.code2x8_matmat_dequant if K == 2 else torch.ops.aqlm.code1x8_matmat_dequant) if rows > 8 else \
             (torch.ops.aqlm.code2x8_matmat if K == 2 else torch.ops.aqlm.code1x8_matmat)
        s = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            yg = op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(yg, op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"]))
        check_close(yg.float().cpu().numpy(), orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"]), torch.float16,
                    f"{K}x8 {rows} rows under hipGraph")


# ------------------------------------------------------------------ module level: QuantizedLinear + autograd + graphs
def _module_from(L, K, nbits, g, fin, fout, dtype):
    from aqlm import QuantizedLinear

    m = QuantizedLinear(fin, fout, g, 1, K, nbits, bias=L["bias"] is not None, device=DEV, dtype=dtype)
    T = to_dev(L, dtype)
    with torch.no_grad():
        m.codes.copy_(T["codes"]); m.codebooks.copy_(T["codebooks"]); m.scales.copy_(T["scales"])
        if T["bias"] is not None:
            m.bias.copy_(T["bias"])
    return m, T


@pytest.mark.parametrize("K,nbits,g", [(1, 16, 8), (2, 8, 8), (8, 8, 32)])
def test_quantized_linear_forward_backward(hk, K, nbits, g):
    fin, fout = 1024, 192
    L = orc.make_layer(31, fin, fout, K, nbits, g, batch=12, bias=True)
    m, T = _module_from(L, K, nbits, g, fin, fout, torch.float16)
    W64 = orc.dequantize_weight(L["codes_unsigned"], L["codebooks"], L["scales"])
    for rows in (3, 12):  # gemv rule (<= 6 rows) and gemm rule
        x = T["x"][:rows].clone().requires_grad_(True)
        y = m(x)
        y64 = L["x"][:rows].astype(np.float64) @ W64.T + L["bias"].astype(np.float64)
        check_close(y.detach().float().cpu().numpy(), y64, torch.float16, f"module fwd rows={rows}")
        gout = torch.randn(rows, fout, generator=torch.Generator().manual_seed(1)).to(torch.float16).to(DEV)
        y.backward(gout)
        gin64 = gout.float().cpu().numpy().astype(np.float64) @ W64
        check_close(x.grad.float().cpu().numpy(), gin64, torch.float16, f"module bwd rows={rows}")


def test_hipgraph_capture_and_side_stream(hk):
    L = orc.make_layer(8, 4096, 512, 1, 16, 8, batch=1, bias=True)
    T = to_dev(L, torch.float16)
    eager = hk.code1x16_matmat(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        side = hk.code1x16_matmat(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    s.synchronize()
    assert torch.equal(side, eager)
    g = torch.cuda.CUDAGraph()
    static_x = T["x"].clone()
    with torch.cuda.graph(g):
        out = hk.code1x16_matmat(static_x, T["codes"], T["codebooks"], T["scales"], T["bias"])
    static_x.copy_(T["x"])
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    static_x.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], T["bias"])


def test_raw_c_abi_strided_rows(hk):
    """Call the C ABI directly with non-trivial x / y row strides."""
    from aqlm_amd import _native

    L = orc.make_layer(15, 2048, 128, 2, 8, 8, batch=3, bias=True)
    T = to_dev(L, torch.float16)
    xbuf = torch.zeros(3, 4096, dtype=torch.float16, device=DEV)
    xbuf[:, :2048] = T["x"]
    ybuf = torch.full((3, 256), 7.0, dtype=torch.float16, device=DEV)
    rc = _native.lib.aqlm_hip_gemv_kx8(T["codes"].data_ptr(), T["codebooks"].data_ptr(), T["scales"].data_ptr(),
                                       T["bias"].data_ptr(), xbuf.data_ptr(), ybuf.data_ptr(), 128, 2048, 2, 8, 3,
                                       4096, 256, _native.F16, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _native.last_error()
    torch.cuda.synchronize()
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    check_close(ybuf[:, :128].float().cpu().numpy(), y64, torch.float16, "raw abi")
    assert (ybuf[:, 128:] == 7.0).all()


# ------------------------------------------------------------------ prepacked (slice-bucketed) 1x16 path, format v7
@pytest.mark.parametrize("entry_bytes", [3, 4])
@pytest.mark.parametrize("fin,fout", [(512, 96), (4096, 300), (11008, 64), (64, 40), (14336, 80), (1024, 2000)])
def test_prepack_matches_the_format_model(hk, fin, fout, entry_bytes):
    """Integer / byte work: the packed buffer (24-bit and 32-bit entries) is held to the numpy model of format v7 --
    tables and bookkeeping bit for bit, the entries up to the order inside a row and the x copy they name, which are the
    repack's bank-aware choice -- and unpacking it must give the codes back.  Small layers use the slices unevenly by
    chance: the repack relabels them, and the relabelling must be the model's (same greedy, same ties)."""
    from aqlm_amd import _native
    from tests import packed_model as pm

    L = orc.make_layer(321 + fin, fin, fout, 1, 16, 8, batch=1, bias=False)
    cu = L["codes_unsigned"][:, :, 0].copy()
    cu[1, :] &= 0x0FFF                      # row 1: everything in slice 0 (a row of many lane-steps, empty elsewhere)
    codes = torch.from_numpy(orc.pack_int_data(cu[:, :, None], 16)).to(DEV)
    _native.set_tuning("packed_entry_bytes", entry_bytes)
    try:
        packed = hk.prepack_1x16(codes)
    finally:
        _native.set_tuning("packed_entry_bytes", 0)
    assert packed is not None
    d = packed.desc
    groups = list(d.slice_groups)[:16]
    new_of_old = pm.plan_labels(cu)
    assert d.relabelled == (new_of_old is not None)
    assert d.variable_geometry == (groups != [16] * 16)
    if d.variable_geometry:
        entry_bytes = 4  # (the variable geometry exists for 4-byte entries only)
    P = pm.pack(cu, groups=groups, new_of_old=new_of_old)
    assert (d.magic, d.version, d.out_features, d.in_features, d.slices_log2, d.entry_bytes) == (pm.MAGIC, pm.VERSION, fout, fin, 4, entry_bytes)
    assert (d.waves, d.steps, d.rows_per_group) == (P["NW"], P["T"], P["RG"])
    G = pm.decode_device_buffer(packed.buf.cpu().numpy(), fout, fin, int(d.waves), int(d.steps), entry_bytes, groups, d.relabelled)
    if d.relabelled:  # the permutation travels with the buffer
        np.testing.assert_array_equal(G["old_of_new"], P["old_of_new"])
    np.testing.assert_array_equal(G["winfo"][:, :, :3], P["winfo"][:, :, :3])
    np.testing.assert_array_equal(G["rowstart"], P["rowstart"])
    # bookkeeping (row-end flags, start rows): position-based -> equal to the model's
    np.testing.assert_array_equal(G["mask"], P["mask"])
    np.testing.assert_array_equal(G["frow"], P["frow"])
    # payload: same number of null entries as the model; walking the buffer like the kernel does gives the codes back
    # and the right sums
    stride = pm.x_stride(fin // 8)
    assert G["spare_ok"] and int(G["copy"].max()) < int(d.x_copies) <= max(1, min(4, 4095 // stride))
    j = G["j"]
    assert int(j.max()) == fin // 8 and int(G["copy"][j == fin // 8].max()) == 0          # nulls name copy 0
    payload = ((j + pm.XB) << 16) | G["code"]                                       # -> the model's entry encoding
    assert int((j == fin // 8).sum()) == int((P["ent"] == ((pm.XB + fin // 8) << 16)).sum())
    Pg = dict(P, ent=payload)
    np.testing.assert_array_equal(pm.unpack(Pg), cu)
    if fin * fout <= 4096 * 300:
        rng = np.random.default_rng(5)
        cb, xx = rng.standard_normal((65536, 8)), rng.standard_normal((1, fin))
        np.testing.assert_allclose(pm.simulate(Pg, cb, xx), xx @ cb[cu].reshape(fout, fin).T, rtol=0, atol=1e-9)
    if d.relabelled:  # the codebook image the kernels read: written by set_codebook_range, entry `new` = entry old_of_new[new]
        cbk = torch.randn(1, 65536, 1, 8, dtype=torch.float16, device=DEV)
        packed.set_codebook_range(cbk)
        assert packed.desc.flags & _native.PACKED_HAS_CODEBOOK
        img = pm.decode_device_buffer(packed.buf.cpu().numpy(), fout, fin, int(d.waves), int(d.steps), entry_bytes, groups, True)["codebook_image"]
        np.testing.assert_array_equal(img, cbk.view(torch.int16).cpu().numpy().view(np.uint16).reshape(65536, 8)[P["old_of_new"]])
    # and it must pay off: fewer LDS bank-group collisions per 16-lane service group than the ascending-j order has
    got = pm.conflict_cycles(dict(P, ent=((G["slot"] + pm.XB) << 16) | G["code"]))
    assert got <= pm.conflict_cycles(P) + 0.02, (got, pm.conflict_cycles(P))
    print(f"LDS cycles per service group and read: packed {got:.3f}, ascending order {pm.conflict_cycles(P):.3f}")
    # lossless, on the GPU and through a re-attached descriptor
    back = hk.unpack_1x16(packed)
    assert torch.equal(back, codes)
    again = hk.PackedCodes.from_buffer(packed.buf.clone())
    assert torch.equal(hk.unpack_1x16(again), codes)
    assert hk.prepack_1x16(torch.zeros(8, 4095, 1, dtype=torch.int16, device=DEV)) is None  # 32760 features: j needs 13 bits


def test_prepack_local_search_lowers_bank_conflicts(hk):
    """The prepack's second ordering pass (pk_improve_kernel) changes no result and no parity test can see it -- and a
    miscompiled cost difference once made it a silent no-op.  Measure what it is for on the device layout: LDS cycles per
    16-lane service group and read (tests/packed_model.py), greedy deal alone vs greedy + local search."""
    from aqlm_amd import _native
    from tests import packed_model as pm

    fin, fout = 4096, 2048
    g = torch.Generator().manual_seed(11)
    cu = torch.randint(0, 65536, (fout, fin // 8, 1), generator=g, dtype=torch.int32)
    codes = (cu - (cu >= 32768) * 65536).to(torch.int16).to(DEV)
    cycles = {}
    try:
        for arrange in (2, 1):
            _native.set_tuning("packed_arrange", arrange)
            packed = hk.prepack_1x16(codes, relabel=False)
            d = packed.desc
            G = pm.decode_device_buffer(packed.buf.cpu().numpy(), fout, fin, int(d.waves), int(d.steps), int(d.entry_bytes))
            P = dict(ent=((G["slot"].astype(np.int64) + pm.XB) << 16) | G["code"], in_groups=fin // 8, NW=int(d.waves), winfo=G["winfo"])
            cycles[arrange] = pm.conflict_cycles(P, max_streams=4)
            assert torch.equal(hk.unpack_1x16(packed), codes)
    finally:
        _native.set_tuning("packed_arrange", 1)
    print(f"LDS cycles per service group and read: greedy {cycles[2]:.3f}, greedy + local search {cycles[1]:.3f}")
    assert cycles[2] < 2.1 and cycles[1] < 0.86 * cycles[2], cycles



def zipf_codes(fout, in_groups, alpha, sorted_labels, seed):
    """Codes as k-means + beam search leave them (src/aq.py:286-356 of the reference: not uniform): entry of rank r is used
    with probability ~ (r + 1)^-alpha; labels sorted by frequency, or shuffled."""
    rng = np.random.default_rng(seed)
    p = np.arange(1, 65537, dtype=np.float64) ** (-alpha)
    p /= p.sum()
    labels = np.arange(65536) if sorted_labels else rng.permutation(65536)
    return labels[rng.choice(65536, size=(fout, in_groups), p=p)].astype(np.int64)


@pytest.mark.parametrize("alpha,sorted_labels", [(0.8, False), (0.8, True), (1.2, False), (1.2, True)])
def test_prepack_balances_skewed_code_histograms_like_the_model(hk, alpha, sorted_labels):
    """Format v7 against its numpy model on codes that use the codebook unevenly: the relabelling is the model's LPT deal, the
    geometry (workgroups per slice) is what the descriptor says and shortens the longest stream, tables / bookkeeping are
    bit-exact, the buffer unpacks to the codes, and the model's kernel walk on the permuted codebook gives W x."""
    from aqlm_amd import _native
    from tests import packed_model as pm

    fin, fout = 2048, 1536
    cu = zipf_codes(fout, fin // 8, alpha, sorted_labels, 7)
    codes = torch.from_numpy(orc.pack_int_data(cu[:, :, None], 16)).to(DEV)
    packed = hk.prepack_1x16(codes)
    assert packed is not None
    d = packed.desc
    groups = list(d.slice_groups)[:16]
    assert not pm.balanced_enough(cu)
    new_of_old = pm.plan_labels(cu)
    assert new_of_old is not None and d.relabelled
    steps = pm.slice_steps(new_of_old[cu])
    assert sum(groups) == 256 and min(groups) >= 8
    if alpha >= 1.2:  # one entry outweighs a slice: labels alone cannot balance it
        assert d.variable_geometry and groups == pm.plan_geometry(steps, fout, min(groups) if min(groups) > 8 else 8)
    else:
        assert not d.variable_geometry
    P = pm.pack(cu, groups=groups, new_of_old=new_of_old)
    assert (d.waves, d.steps, d.rows_per_group, d.entry_bytes) == (P["NW"], P["T"], P["RG"], 4)
    # the balanced layout is what it is for: the longest stream within 15 % of the mean one (format v6 on these codes: 2-14x)
    ls, a = pm.lane_steps(new_of_old[cu], P["geom"])
    totals = a[:, -1]
    assert totals.max() <= 1.15 * totals.mean() + 8, (int(totals.max()), float(totals.mean()))
    _, a6 = pm.lane_steps(cu)
    print(f"longest / mean stream: v7 {totals.max() / totals.mean():.2f}, labels as they are {a6[:, -1].max() / a6[:, -1].mean():.2f}; groups {groups}")
    G = pm.decode_device_buffer(packed.buf.cpu().numpy(), fout, fin, int(d.waves), int(d.steps), 4, groups, True)
    np.testing.assert_array_equal(G["old_of_new"], P["old_of_new"])
    np.testing.assert_array_equal(G["winfo"][:, :, :3], P["winfo"][:, :, :3])
    np.testing.assert_array_equal(G["rowstart"], P["rowstart"])
    np.testing.assert_array_equal(G["mask"], P["mask"])
    np.testing.assert_array_equal(G["frow"], P["frow"])
    Pg = dict(P, ent=((G["j"] + pm.XB) << 16) | G["code"])
    np.testing.assert_array_equal(pm.unpack(Pg), cu)
    assert torch.equal(hk.unpack_1x16(packed), codes)
    assert torch.equal(hk.unpack_1x16(hk.PackedCodes.from_buffer(packed.buf.clone())), codes)
    rng = np.random.default_rng(5)
    cb, xx = rng.standard_normal((65536, 8)), rng.standard_normal((1, fin))
    np.testing.assert_allclose(pm.simulate(Pg, cb, xx), xx @ cb[cu].reshape(fout, fin).T, rtol=0, atol=1e-9)
    # the two opt-outs: labels as they are (format v6 behaviour), and relabelling on the 16 x 16 geometry
    v6 = hk.prepack_1x16(codes, relabel=False, uniform_only=True)
    assert v6 is None or (not v6.desc.relabelled and not v6.desc.variable_geometry and torch.equal(hk.unpack_1x16(v6), codes))
    uni = hk.prepack_1x16(codes, uniform_only=True)
    assert uni is not None and uni.desc.relabelled and not uni.desc.variable_geometry and torch.equal(hk.unpack_1x16(uni), codes)


def test_prepack_model_default_keeps_one_copy_until_the_codes_are_needed(hk):
    """`prepack_model()` (the deployment call) leaves ONE copy of the 1x16 codes by default -- the canonical codes
    are dropped, decode calls (<= 6 rows) never miss them --, and a layer that is then called with 7+ rows takes them back for good at
    that call (a transient unpack costs 2.4-2.9 x the op it would precede: `detail.unpack_1x16_us`) instead of unpacking on every
    call; `drop_canonical=True` never restores; planar 8x8 layers keep both copies by default (same size; the fused MFMA kernel reads
    the canonical layout).  Results are the same bits before and after, `state_dict()` always carries the canonical codes."""
    from aqlm.checkpoint import prepack_model

    fin, fout = 2048, 1536
    L = orc.make_layer(41, fin, fout, 1, 16, 8, batch=9, bias=True)
    m, T = _module_from(L, 1, 16, 8, fin, fout, torch.float16)
    L8 = orc.make_layer(42, fin, 256, 8, 8, 32, batch=2, bias=False)
    m8, T8 = _module_from(L8, 8, 8, 32, fin, 256, torch.float16)
    holder = torch.nn.ModuleDict({"a": m, "b": m8})
    with torch.no_grad():
        prepack_model(holder, min_codes=100_000, drop_canonical=False)              # both copies: the references of every route
        ref3, ref9 = m(T["x"][:3]), m(T["x"])
        assert m._packed_codes is not None and not m._codes_dropped
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        rep = prepack_model(holder, min_codes=100_000)
        assert m._codes_dropped and not m._codes_drop_strict and m.codes.numel() == 0 and rep["codes_dropped_layers"] == 1
        assert not m8._codes_dropped and m8.codes.numel() > 0                       # planar 8x8: both copies stay
        assert torch.equal(m(T["x"][:3]), ref3) and m._codes_dropped                # decode calls never need the codes
        assert all(torch.equal(v, sd[k]) for k, v in m.state_dict().items()) and m._codes_dropped   # state_dict: transient unpack
        assert torch.equal(m(T["x"]), ref9)                                         # 9 rows: the codes come back ...
        assert not m._codes_dropped and torch.equal(m.codes, T["codes"])            # ... for good
        assert torch.equal(m(T["x"][:3]), ref3) and torch.equal(m(T["x"]), ref9)
        prepack_model(holder, min_codes=100_000, drop_canonical=True)               # explicit: never restored, 8x8 included
        assert m._codes_dropped and m._codes_drop_strict and m8._codes_dropped
        assert torch.equal(m(T["x"]), ref9) and m._codes_dropped
        prepack_model(holder, min_codes=100_000, drop_canonical=False)
        assert m._codes_dropped                                                     # False keeps what is there; it does not undo a drop


def test_stale_codebook_image_in_a_capture_or_a_compiled_graph(hk):
    """a relabelled (format v7) buffer's kernels read a derived codebook IMAGE.  (a) The codebook changes and the next
    forward happens inside a hipGraph capture, where the image cannot be rewritten (it reads a bound back): the module runs the
    direct kernel on the canonical codes and the live codebook instead of raising; the first eager forward rewrites the image.
    (b) A compiled model bakes the descriptor in: the dispatcher op compares the codebook fingerprint that travels with it and
    refreshes the image when the codebook was updated in place between two calls."""
    from aqlm import QuantizedLinear

    fin, fout = 2048, 1536
    L = orc.make_layer(61, fin, fout, 1, 16, 8, batch=2, bias=True)
    cu = zipf_codes(fout, fin // 8, 1.0, True, 3)[:, :, None]
    L = dict(L, codes=orc.pack_int_data(cu, 16), codes_unsigned=cu)
    T = to_dev(L, torch.float16)
    import aqlm_amd.inference as inf

    old, inf.PREPACK_MIN_CODES = inf.PREPACK_MIN_CODES, 100_000
    try:
        m = QuantizedLinear(fin, fout, 8, 1, 1, 16, bias=True, device=DEV, dtype=torch.float16)
        with torch.no_grad():
            m.codes.copy_(T["codes"]); m.codebooks.copy_(T["codebooks"]); m.scales.copy_(T["scales"].reshape(m.scales.shape)); m.bias.copy_(T["bias"])
            x = T["x"][:1].contiguous()
            y0 = m(x)
            assert m._packed_codes is not None and m._packed_codes.desc.relabelled
            m.codebooks.mul_(0.5)                                   # versioned in-place update: the image is stale now
            want = hk.code1x16_matmat(x, m.codes, m.codebooks, m.scales, m.bias)   # direct kernel, live codebook
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                yg = m(x)                                           # (a) no exception
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(yg, want)
            y1 = m(x)                                               # eager: the image is rewritten, the packed kernel is back
            check_close(y1.float().cpu().numpy(), want.double().cpu().numpy(), torch.float16, "packed kernel on the refreshed image")
            assert m._packed_codes.range_is_current(m.codebooks)
            # (b) compiled: trace with the current codebook, update it in place, call again
            cm = torch.compile(m, fullgraph=True)
            yc0 = cm(x)
            assert torch.equal(yc0, y1)
            m.codebooks.mul_(2.0)
            yc1 = cm(x)
            check_close(yc1.float().cpu().numpy(), y0.double().cpu().numpy(), torch.float16, "compiled model after an in-place codebook update")
    finally:
        inf.PREPACK_MIN_CODES = old


def test_prepack_deals_out_row_correlated_labels(hk):
    """VERDICT r05 weak #1 on the device: label use correlated with the row (rows of block b draw 90 % of their codes from the labels
    [4096 b, 4096 (b + 1)); global usage flat).  The repack must relabel (forced deal, equal to the model's), end with the longest
    stream within 15 % of the mean, unpack losslessly -- on a small layer against the model bit for bit, and on a full-size layer
    (4096 -> 4096: 13.8 x longest / mean with the checkpoint's labels) through the kernel against the oracle."""
    from tests import packed_model as pm

    fin, fout = 2048, 1536
    cu = pm.rowblock_codes(fout, fin // 8, 0.9, 5)
    codes = torch.from_numpy(orc.pack_int_data(cu[:, :, None], 16)).to(DEV)
    packed = hk.prepack_1x16(codes)
    assert packed is not None and packed.desc.relabelled
    d = packed.desc
    groups = list(d.slice_groups)[:16]
    new_of_old = pm.plan_labels(cu)
    assert new_of_old is not None
    P = pm.pack(cu, groups=groups, new_of_old=new_of_old)
    assert (d.waves, d.steps, d.rows_per_group, d.entry_bytes) == (P["NW"], P["T"], P["RG"], 4)
    G = pm.decode_device_buffer(packed.buf.cpu().numpy(), fout, fin, int(d.waves), int(d.steps), 4, groups, True)
    np.testing.assert_array_equal(G["old_of_new"], P["old_of_new"])
    np.testing.assert_array_equal(G["rowstart"], P["rowstart"])
    assert torch.equal(hk.unpack_1x16(packed), codes)
    ls, a = pm.lane_steps(new_of_old[cu], P["geom"])
    _, a6 = pm.lane_steps(cu)
    assert a[:, -1].max() <= 1.15 * a[:, -1].mean() + 8 and a6[:, -1].max() > 4 * a6[:, -1].mean()
    # full size, through the kernel
    fin, fout = 4096, 4096
    L = orc.make_layer(5100, fin, fout, 1, 16, 8, batch=4, bias=True)
    cu = pm.rowblock_codes(fout, fin // 8, 0.9, 6)[:, :, None]
    L = dict(L, codes=orc.pack_int_data(cu, 16), codes_unsigned=cu)
    T = to_dev(L, torch.float16)
    packed = hk.prepack_1x16(T["codes"], codebooks=T["codebooks"])
    assert packed is not None and packed.desc.relabelled, "a row-correlated layer fell off the packed path"
    uniform_steps = hk.prepack_1x16(to_dev(orc.make_layer(5101, fin, fout, 1, 16, 8, batch=1, bias=False), torch.float16)["codes"]).desc.steps
    assert packed.desc.steps * packed.desc.waves <= 1.15 * uniform_steps * 16 + 16, (int(packed.desc.steps), int(packed.desc.waves), int(uniform_steps))
    assert torch.equal(hk.unpack_1x16(packed), T["codes"])
    ref = c_oracle.DequantGemv(L["codebooks"], L["codes"], L["scales"], L["bias"], 16, nthreads=0)
    y4 = hk.code1x16_matmat_packed(T["x"], packed, T["codebooks"], T["scales"], T["bias"])
    for b in range(4):
        check_close(y4[b].float().cpu().numpy(), ref(L["x"][b]).copy(), torch.float16, f"row-correlated codes, row {b}")
    y1 = hk.code1x16_matmat_packed(T["x"][:1], packed, T["codebooks"], T["scales"], T["bias"])
    assert torch.equal(y4[0], y1[0])


@pytest.mark.parametrize("alpha,sorted_labels", [(0.5, False), (0.5, True), (0.8, False), (0.8, True), (1.0, False), (1.0, True),
                                                 (1.2, False), (1.2, True)])
def test_gemv_1x16_packed_on_skewed_code_histograms(hk, alpha, sorted_labels):
    """A full-size layer (4096 -> 4096) whose codes follow a Zipf law: every case takes the prepacked kernel (format v6 left
    alpha >= 0.8 and every frequency-sorted labelling to the direct kernel), unpacks losslessly and meets the oracle -- 1 and
    4 rows, rows bit-identical to single-row launches, shared-input launches bit-identical to separate ones."""
    fin, fout = 4096, 4096
    L = orc.make_layer(5000 + int(alpha * 10), fin, fout, 1, 16, 8, batch=4, bias=True)
    cu = zipf_codes(fout, fin // 8, alpha, sorted_labels, 11 + int(alpha * 10))[:, :, None]
    L = dict(L, codes=orc.pack_int_data(cu, 16), codes_unsigned=cu)
    T = to_dev(L, torch.float16)
    packed = hk.prepack_1x16(T["codes"], codebooks=T["codebooks"])
    assert packed is not None, "a skewed layer fell off the packed path"
    assert packed.desc.relabelled or (alpha <= 0.8 and not sorted_labels)  # (mild skew with shuffled labels may already run the minimum number of wave-steps)
    assert packed.desc.variable_geometry == (alpha >= 1.0)
    assert torch.equal(hk.unpack_1x16(packed), T["codes"])
    ref = c_oracle.DequantGemv(L["codebooks"], L["codes"], L["scales"], L["bias"], 16, nthreads=0)
    y1 = hk.code1x16_matmat_packed(T["x"][:1], packed, T["codebooks"], T["scales"], T["bias"])
    check_close(y1[0].float().cpu().numpy(), ref(L["x"][0]).copy(), torch.float16, f"zipf {alpha} sorted={sorted_labels}")
    y4 = hk.code1x16_matmat_packed(T["x"], packed, T["codebooks"], T["scales"], T["bias"])
    for b in range(4):
        check_close(y4[b].float().cpu().numpy(), ref(L["x"][b]).copy(), torch.float16, f"zipf {alpha} row {b}")
    assert torch.equal(y4[0], y1[0])
    for _ in range(5):
        assert torch.equal(y1, hk.code1x16_matmat_packed(T["x"][:1], packed, T["codebooks"], T["scales"], T["bias"]))
    yz = hk.code1x16_matmat_packed(torch.zeros_like(T["x"][:1]), packed, T["codebooks"], T["scales"], T["bias"])
    assert torch.equal(yz[0], T["bias"])
    # a retrained codebook: the image follows the tensor (set_codebook_range is keyed on its version)
    cb2 = (T["codebooks"].float() * 0.5).half()
    y_half = hk.code1x16_matmat_packed(T["x"][:1], packed, cb2, T["scales"], None)
    y_full = hk.code1x16_matmat_packed(T["x"][:1], packed, T["codebooks"], T["scales"], None)
    check_close(y_half.float().cpu().numpy() * 2.0, y_full.float().cpu().numpy().astype(np.float64), torch.float16, "codebook image follows the tensor")
    # shared-input launch next to a uniform layer: bit-identical to separate launches
    L2 = orc.make_layer(77, fin, 1024, 1, 16, 8, batch=1, bias=False)
    T2 = to_dev(L2, torch.float16)
    pk2 = hk.prepack_1x16(T2["codes"], codebooks=T2["codebooks"])
    for B in (1, 2):
        outs = hk.code1x16_matmat_packed_multi(T["x"][:B], [packed, pk2], [T["codebooks"], T2["codebooks"]],
                                               [T["scales"], T2["scales"]], [T["bias"], None])
        assert torch.equal(outs[0], y4[:B])
        assert torch.equal(outs[1], hk.code1x16_matmat_packed(T["x"][:B], pk2, T2["codebooks"], T2["scales"], None))
    # the module: same layer through QuantizedLinear (prepacks itself, fast lane and all), hipGraph replay included
    from aqlm import QuantizedLinear

    m = QuantizedLinear(fin, fout, 8, 1, 1, 16, bias=True, device=DEV, dtype=torch.float16)
    with torch.no_grad():
        m.codes.copy_(T["codes"]); m.codebooks.copy_(T["codebooks"]); m.scales.copy_(T["scales"].reshape(m.scales.shape)); m.bias.copy_(T["bias"])
        ym = m(T["x"][:1])
        assert m._packed_codes is not None and torch.equal(ym, y1)
        ym2 = m(T["x"][:1])
        assert torch.equal(ym2, y1)
        g = torch.cuda.CUDAGraph()
        sx = T["x"][:1].clone()
        with torch.cuda.graph(g):
            yg = m(sx)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(yg, y1)


PACKED_SHAPES = [
    (4096, 4096, "float16", True),
    (4096, 1000, "float16", False),      # ragged row groups
    (4096, 37, "bfloat16", True),        # fewer rows than 16 x waves
    (8192, 512, "float16", True),
    (11008, 640, "float16", True),       # 1376 input groups: x is not a whole number of LDS-DMA chunks
    (14336, 1024, "bfloat16", True),
    (64, 256, "float16", True),          # 8 input groups: almost every (row, slice) bucket is a single null lane-step
    (1024, 3000, "float16", True),
    (512, 7168, "bfloat16", False),
    (2048, 1500, "float16", True),
    (520, 64, "float16", True),          # 65 input groups (not a multiple of 8)
    (256, 57344, "float16", False),      # 3584 rows per row group: the one-row image puts the slice first (no x window)
]


@pytest.mark.parametrize("entry_bytes", [3, 4])
@pytest.mark.parametrize("fin,fout,dt,bias", PACKED_SHAPES)
def test_gemv_1x16_packed(hk, fin, fout, dt, bias, entry_bytes):
    from aqlm_amd import _native

    dtype = tdtype(dt)
    L = orc.make_layer(700 + fin + fout, fin, fout, 1, 16, 8, batch=8, bias=bias,
                       float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    # skew the slice populations: rows whose codes all sit in one slice / in two slices (long runs of lane-steps in one
    # stream, single null lane-steps in the others)
    cu = L["codes_unsigned"].copy()
    cu[1, :, 0] = cu[1, :, 0] & 0x0FFF
    cu[2, ::2, 0] = (cu[2, ::2, 0] & 0x0FFF) | (5 << 12)
    cu[fout - 1, :, 0] = (cu[fout - 1, :, 0] & 0x0FFF) | (15 << 12)
    L = dict(L, codes=orc.pack_int_data(cu, 16), codes_unsigned=cu)
    T = to_dev(L, dtype)
    _native.set_tuning("packed_entry_bytes", entry_bytes)
    try:
        packed = hk.prepack_1x16(T["codes"])
    finally:
        _native.set_tuning("packed_entry_bytes", 0)
    assert packed.desc.entry_bytes == entry_bytes
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    y1 = hk.code1x16_matmat_packed(T["x"][:1], packed, T["codebooks"], T["scales"], T["bias"])
    check_close(y1.float().cpu().numpy(), y64[:1], dtype, f"packed 1x16g8 {fin}->{fout}")
    if bias:
        yz = hk.code1x16_matmat_packed(torch.zeros_like(T["x"][:1]), packed, T["codebooks"], T["scales"], T["bias"])
        assert torch.equal(yz[0], T["bias"])
    # determinism: the summation order is fixed (LDS adds of one wave in program order, carries in wave order)
    for _ in range(10):
        assert torch.equal(y1, hk.code1x16_matmat_packed(T["x"][:1], packed, T["codebooks"], T["scales"], T["bias"]))
    # 2..8 rows per launch: every row equals the same row launched alone, bit for bit
    for B in (2, 3, 5, 8):
        yb = hk.code1x16_matmat_packed(T["x"][:B], packed, T["codebooks"], T["scales"], T["bias"])
        check_close(yb.float().cpu().numpy(), y64[:B], dtype, f"packed batch {B}")
        assert torch.equal(yb[0], y1[0])
        alone = hk.code1x16_matmat_packed(T["x"][B - 1:B].contiguous(), packed, T["codebooks"], T["scales"], T["bias"])
        assert torch.equal(yb[B - 1], alone[0])
    # more than 8 rows: chunks of 8; 3-D input
    y3d = hk.code1x16_matmat_packed(T["x"].reshape(2, 4, fin), packed, T["codebooks"], T["scales"], T["bias"])
    assert y3d.shape == (2, 4, fout)
    check_close(y3d.reshape(8, fout).float().cpu().numpy(), y64, dtype, "packed 3-D")


# 16-element codebook vectors (the reference kernel's second template instance, cuda_kernel.cu:476-521; BASELINE config
# "1x16g16"): the second instantiation of csrc/gemv_packed.hip -- 32 slices of 2048 x 32 B, lane-parity half order.
PACKED_G16_SHAPES = [
    (4096, 4096, "float16", True),
    (4096, 11008, "float16", True),
    (4096, 1000, "bfloat16", False),     # ragged row groups
    (8192, 1024, "float16", True),
    (11008, 640, "float16", True),       # 688 input groups
    (128, 256, "float16", True),         # 8 input groups: null lane-steps almost everywhere
    (1040, 70, "bfloat16", True),        # 65 input groups, fewer rows than 8 x waves
    (28672, 512, "float16", True),       # the widest x image (56 KiB) behind the slice
    (1024, 28672, "float16", True),      # 3584 rows per row group: the one-row image puts the slice first (no x window)
]


@pytest.mark.parametrize("fin,fout,dt,bias", PACKED_G16_SHAPES)
def test_gemv_1x16_g16_packed(hk, fin, fout, dt, bias):
    dtype = tdtype(dt)
    L = orc.make_layer(900 + fin + fout, fin, fout, 1, 16, 16, batch=8, bias=bias,
                       float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    cu = L["codes_unsigned"].copy()
    cu[1, :, 0] = cu[1, :, 0] & 0x07FF                             # a row that lives in one slice
    cu[2, ::2, 0] = (cu[2, ::2, 0] & 0x07FF) | (5 << 11)
    cu[fout - 1, :, 0] = (cu[fout - 1, :, 0] & 0x07FF) | (31 << 11)
    L = dict(L, codes=orc.pack_int_data(cu, 16), codes_unsigned=cu)
    T = to_dev(L, dtype)
    packed = hk.prepack_1x16(T["codes"], 16)
    assert packed is not None and packed.in_group_size == 16 and packed.slices == 32 and packed.desc.entry_bytes == 4
    # lossless, also through a re-attached descriptor
    assert torch.equal(hk.unpack_1x16(packed), T["codes"])
    assert torch.equal(hk.unpack_1x16(hk.PackedCodes.from_buffer(packed.buf.clone())), T["codes"])
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    for fused in (True, False):  # single kernel (fixed-point cells) and the two-kernel form (fp32 slice partials)
        hk.set_fused_finalize(fused)
        try:
            if fused:
                packed.set_codebook_range(T["codebooks"])
            y1 = hk.code1x16_matmat_packed(T["x"][:1], packed, T["codebooks"], T["scales"], T["bias"])
            check_close(y1.float().cpu().numpy(), y64[:1], dtype, f"packed 1x16g16 {fin}->{fout} fused={fused}")
            if bias:
                yz = hk.code1x16_matmat_packed(torch.zeros_like(T["x"][:1]), packed, T["codebooks"], T["scales"], T["bias"])
                assert torch.equal(yz[0], T["bias"])
            for _ in range(5):
                assert torch.equal(y1, hk.code1x16_matmat_packed(T["x"][:1], packed, T["codebooks"], T["scales"], T["bias"]))
            for B in (2, 3, 5, 8):
                yb = hk.code1x16_matmat_packed(T["x"][:B], packed, T["codebooks"], T["scales"], T["bias"])
                check_close(yb.float().cpu().numpy(), y64[:B], dtype, f"packed g16 batch {B}")
                assert torch.equal(yb[0], y1[0])
                alone = hk.code1x16_matmat_packed(T["x"][B - 1:B].contiguous(), packed, T["codebooks"], T["scales"], T["bias"])
                assert torch.equal(yb[B - 1], alone[0])
        finally:
            hk.set_fused_finalize(True)
    # the direct (L2-gather) kernel on the same layer agrees to rounding
    yd = hk._gemv(T["x"][:1], T["codes"], T["codebooks"], T["scales"], T["bias"], "1x16")
    check_close(y1.float().cpu().numpy(), yd.float().cpu().numpy().astype(np.float64), dtype, "packed g16 vs direct")


def test_g16_layers_share_one_launch_and_the_module_prepacks_them(hk):
    """q/k/v-style shared-input launch of two 1x16g16 layers == separate launches, bit for bit; QuantizedLinear prepacks a
    large g16 layer like a g8 one and can drop / restore its canonical codes."""
    from aqlm_amd import QuantizedLinear

    fin = 4096
    Ls = [orc.make_layer(77 + k, fin, fo, 1, 16, 16, batch=2, bias=(k == 0)) for k, fo in enumerate((4096, 1024))]
    Ts = [to_dev(L, torch.float16) for L in Ls]
    pks = [hk.prepack_1x16(T["codes"], 16, codebooks=T["codebooks"]) for T in Ts]
    x = Ts[0]["x"]
    outs = hk.code1x16_matmat_packed_multi(x, pks, [T["codebooks"] for T in Ts], [T["scales"] for T in Ts],
                                           [Ts[0]["bias"], None])
    for k in range(2):
        assert torch.equal(outs[k], hk.code1x16_matmat_packed(x, pks[k], Ts[k]["codebooks"], Ts[k]["scales"],
                                                              Ts[0]["bias"] if k == 0 else None))
    m = QuantizedLinear(fin, 4096, 16, 1, 1, 16, bias=True, device=DEV, dtype=torch.float16)
    with torch.no_grad():
        m.codes.copy_(Ts[0]["codes"]); m.codebooks.copy_(Ts[0]["codebooks"]); m.scales.copy_(Ts[0]["scales"])
        m.bias.copy_(Ts[0]["bias"])
    y = m(x[:1])
    assert m._packed_codes is not None and m._packed_codes.in_group_size == 16
    assert torch.equal(y, outs[0][:1])
    assert m.drop_canonical_codes()
    assert torch.equal(m(x[:1]), y)
    m.restore_canonical_codes()
    assert torch.equal(m.codes, Ts[0]["codes"])


# The shipped kernel at the shapes the bench and the 70B configuration run (BASELINE.json configs 2 and 5), against the
# C restatement of the reference's dequantize_gemm.
HEADLINE = [(4096, 4096), (4096, 11008), (4096, 14336), (14336, 4096), (4096, 1024), (8192, 28672), (1024, 28672),
            (2048, 28672), (11008, 4096), (8192, 8192), (28672, 8192), (5120, 13824)]


@pytest.mark.parametrize("dt", ["float16", "bfloat16"])
def test_packed_fused_finalize(hk, dt):
    """The single-kernel form of the prepacked matvec: the 16 slice workgroups of a row add fixed-point slice sums into
    one 64-bit cell of the packed buffer and the last arrival writes y.  Held to the oracle, to the two-kernel form, to
    bit-exact repeatability, to "cells are zero at rest", and to inputs at both ends of the storage type's range (the
    fixed-point scale is derived from max|x| and max|codebook|, so neither overflow nor loss of resolution may occur)."""
    from aqlm_amd import _native
    from tests import packed_model as pm

    dtype = tdtype(dt)
    fin, fout = 4096, 1536
    L = orc.make_layer(777, fin, fout, 1, 16, 8, batch=8, bias=True, float_dtype=np.float16 if dt == "float16" else "bfloat16")
    T = to_dev(L, dtype)
    packed = hk.prepack_1x16(T["codes"])
    assert packed is not None and packed.desc.codebook_absmax == 0.0
    lay = pm.layout(fout, fin // 8, int(packed.desc.waves), int(packed.desc.steps), int(packed.desc.entry_bytes))
    cells = lambda: packed.buf[lay["off_acc"]: lay["off_acc"] + 8 * fout * 8]
    assert int(cells().max()) == 0 and lay["used"] == packed.buf.numel()

    def run(x):
        return hk.code1x16_matmat_packed(x, packed, T["codebooks"], T["scales"], T["bias"])

    hk.set_fused_finalize(False)
    try:
        y_two = run(T["x"])                       # also records the codebook range in the descriptor
    finally:
        hk.set_fused_finalize(True)
    absmax = float(T["codebooks"].abs().max())
    assert packed.desc.codebook_absmax == pytest.approx(absmax)
    y = run(T["x"])
    assert int(cells().max()) == 0, "the accumulator cells must be back to zero when the kernel has finished"
    x64 = T["x"].double().cpu().numpy()
    cb64, sc64, bi64 = (T[k].double().cpu().numpy() for k in ("codebooks", "scales", "bias"))
    y64 = orc.dequantize_gemm(x64, L["codes"], cb64, sc64, bi64)
    check_close(y.float().cpu().numpy(), y64, dtype, "fused finalize")
    check_close(y.float().cpu().numpy(), y_two.double().cpu().numpy(), dtype, "fused vs two-kernel finalize")
    for b in range(8):                            # a row's result does not depend on its neighbours in the launch
        assert torch.equal(run(T["x"][b:b + 1])[0], y[b])
    for _ in range(20):                           # integer adds commute: the arrival order of the workgroups is invisible
        assert torch.equal(run(T["x"]), y)
    # magnitudes: the largest finite inputs with tiny scales, and inputs near the bottom of the normal range
    big = torch.full_like(T["x"], 3.0e4) * torch.sign(T["x"])
    sc_small = T["scales"] * 1e-4
    yb = hk.code1x16_matmat_packed(big, packed, T["codebooks"], sc_small, None)
    yb64 = orc.dequantize_gemm(big.double().cpu().numpy(), L["codes"], cb64, sc_small.double().cpu().numpy(), None)
    assert np.isfinite(yb64).all() and np.abs(yb64).max() < (6e4 if dt == "float16" else 1e30)
    check_close(yb.float().cpu().numpy(), yb64, dtype, "fused finalize, |x| = 3e4")
    small = T["x"] * 2.0 ** -12
    ys = hk.code1x16_matmat_packed(small, packed, T["codebooks"], T["scales"], None)
    ys64 = orc.dequantize_gemm(small.double().cpu().numpy(), L["codes"], cb64, sc64, None)
    check_close(ys.float().cpu().numpy(), ys64, dtype, "fused finalize, |x| ~ 2^-12")
    assert torch.equal(hk.code1x16_matmat_packed(torch.zeros_like(T["x"][:2]), packed, T["codebooks"], T["scales"], T["bias"]),
                       T["bias"].expand(2, fout))
    # non-finite inputs surface as NaN rows of that input row only, and leave clean cells behind
    xn = T["x"].clone()
    xn[1, 17] = float("nan")
    xn[2, 4000] = float("inf")
    yn = run(xn)
    assert torch.isnan(yn[1]).all() and not torch.isfinite(yn[2]).any() and torch.equal(yn[0], y[0]) and torch.equal(yn[3:], y[3:])
    assert int(cells().max()) == 0
    assert torch.equal(run(T["x"]), y)
    # a retrained codebook (in-place update): the recorded range is refreshed before the next launch
    with torch.no_grad():
        T["codebooks"].mul_(8.0)
    y8 = run(T["x"])
    assert packed.desc.codebook_absmax == pytest.approx(8.0 * absmax)
    y8_64 = orc.dequantize_gemm(x64, L["codes"], 8.0 * cb64, sc64, bi64)
    check_close(y8.float().cpu().numpy(), y8_64, dtype, "fused finalize after a codebook update")
    # an unknown range (descriptor says 0) is the two-kernel path, through the raw C ABI as well
    d0 = _native.PackedDesc.from_ints(packed.desc.as_ints())
    d0.codebook_absmax = 0.0
    pk0 = hk.PackedCodes(packed.buf, d0)
    pk0._range_of = (T["codebooks"].data_ptr(), T["codebooks"]._version)
    check_close(hk.code1x16_matmat_packed(T["x"], pk0, T["codebooks"], T["scales"], T["bias"]).float().cpu().numpy(), y8_64, dtype,
                "two-kernel path when the codebook range is unknown")


@pytest.mark.raw_prepack
def test_raw_op_packs_large_layers_transparently(hk, monkeypatch):
    """aqlm::code1x16_matmat with canonical codes (the reference's stateless signature, what its benchmark script calls):
    large layers run on the prepacked kernel through a cache keyed by the codes tensor; in-place edits and dead tensors
    invalidate it; nothing is packed during hipGraph capture; fresh view objects on every call switch it off."""
    import gc

    fin, fout = 4096, 4096
    L = orc.make_layer(31337, fin, fout, 1, 16, 8, batch=2, bias=True)
    T = to_dev(L, torch.float16)
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    stats = hk._RAW_STATS
    p0, h0 = stats["packs"], stats["hits"]
    y_a = torch.ops.aqlm.code1x16_matmat(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    assert stats["packs"] == p0 + 1 and stats["bytes"] > 2 * T["codes"].numel()
    y_b = hk.code1x16_matmat(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    assert stats["packs"] == p0 + 1 and stats["hits"] == h0 + 1
    assert torch.equal(y_a, y_b)
    check_close(y_a.float().cpu().numpy(), y64, torch.float16, "raw op, packed through the cache")
    y_direct = hk._gemv(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"], "1x16")
    check_close(y_a.float().cpu().numpy(), y_direct.float().cpu().numpy().astype(np.float64), torch.float16, "cache vs direct")
    # 7 rows and more go to the MFMA kernel (same op, other summation order); a bf16 input on an fp16 layer raises
    x9 = T["x"][:1].expand(9, fin).contiguous()
    y9 = hk.code1x16_matmat(x9, T["codes"], T["codebooks"], T["scales"], T["bias"])
    check_close(y9[0].float().cpu().numpy(), y64[0], torch.float16, "raw op, 9 rows")
    assert torch.equal(y9[0], y9[8])
    with pytest.raises(NotImplementedError):
        hk.code1x16_matmat(T["x"].bfloat16(), T["codes"], T["codebooks"], T["scales"], T["bias"])
    # in-place edit of the codes: the cached buffer is stale and must not be used
    L2 = orc.make_layer(31338, fin, fout, 1, 16, 8, batch=2, bias=True)
    T["codes"].copy_(torch.from_numpy(L2["codes"]).to(DEV))
    y_c = hk.code1x16_matmat(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    assert stats["packs"] == p0 + 2
    y64c = orc.dequantize_gemm(L["x"], L2["codes"], L["codebooks"], L["scales"], L["bias"])
    check_close(y_c.float().cpu().numpy(), y64c, torch.float16, "raw op after an in-place edit of the codes")
    # a dead tensor leaves nothing behind
    key = id(T["codes"])
    assert key in hk._RAW_PACKED
    held = stats["bytes"]
    del T["codes"]
    gc.collect()
    assert key not in hk._RAW_PACKED and stats["bytes"] < held
    # small layers and hipGraph capture: direct kernel, nothing packed
    Ls = orc.make_layer(5, 512, 256, 1, 16, 8, batch=1, bias=False)
    Ts = to_dev(Ls, torch.float16)
    p1 = stats["packs"]
    hk.code1x16_matmat(Ts["x"], Ts["codes"], Ts["codebooks"], Ts["scales"], None)
    T3 = to_dev(orc.make_layer(31339, fin, fout, 1, 16, 8, batch=1, bias=False), torch.float16)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hk._gemv(T3["x"], T3["codes"], T3["codebooks"], T3["scales"], None, "1x16")  # warm-up outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        yg = hk.code1x16_matmat(T3["x"], T3["codes"], T3["codebooks"], T3["scales"], None)
    g.replay()
    torch.cuda.synchronize()
    assert stats["packs"] == p1
    assert torch.equal(yg, hk._gemv(T3["x"], T3["codes"], T3["codebooks"], T3["scales"], None, "1x16"))
    # a caller that hands over a new view object every time never hits: the cache gives up instead of packing per call
    monkeypatch.setattr(hk, "RAW_OP_PREPACK_MAX_MISSES", 3)
    hk.clear_raw_op_prepack_cache()
    p2 = stats["packs"]
    for _ in range(6):
        hk.code1x16_matmat(T3["x"], T3["codes"].view(fout, fin // 8, 1), T3["codebooks"], T3["scales"], None)
        gc.collect()
    assert stats["packs"] == p2 + 3
    # bounded: with room for one packed layer only, a second layer evicts the first (least recently used), the byte count stays
    # under the cap and the evicted layer is simply packed again when it comes back
    monkeypatch.setattr(hk, "RAW_OP_PREPACK_MAX_MISSES", 100)
    hk.clear_raw_op_prepack_cache()
    T4 = to_dev(orc.make_layer(31340, fin, fout, 1, 16, 8, batch=1, bias=False), torch.float16)
    y3 = hk.code1x16_matmat(T3["x"], T3["codes"], T3["codebooks"], T3["scales"], None)
    one = stats["bytes"]
    monkeypatch.setattr(hk, "RAW_OP_PREPACK_MAX_BYTES", int(one * 1.7))
    hk.code1x16_matmat(T4["x"], T4["codes"], T4["codebooks"], T4["scales"], None)
    assert stats["bytes"] <= int(one * 1.7) and id(T3["codes"]) not in hk._RAW_PACKED and id(T4["codes"]) in hk._RAW_PACKED
    assert torch.equal(hk.code1x16_matmat(T3["x"], T3["codes"], T3["codebooks"], T3["scales"], None), y3)
    assert id(T3["codes"]) in hk._RAW_PACKED and id(T4["codes"]) not in hk._RAW_PACKED
    hk.clear_raw_op_prepack_cache()


@pytest.mark.parametrize("dt", ["float16", "bfloat16"])
@pytest.mark.parametrize("fin,fout", HEADLINE)
def test_packed_kernel_at_headline_shapes_vs_c_oracle(hk, fin, fout, dt):
    dtype = tdtype(dt)
    if dt == "bfloat16" and (fin, fout) not in ((4096, 4096), (4096, 11008), (4096, 14336), (14336, 4096), (8192, 28672)):
        pytest.skip("bf16: the headline pair, the true Llama-3-8B MLP shapes and the 70B layer")
    L = orc.make_layer(4242 + fin + fout, fin, fout, 1, 16, 8, batch=4, bias=True,
                       float_dtype=np.float16 if dt == "float16" else "bfloat16")
    T = to_dev(L, dtype)
    packed = hk.prepack_1x16(T["codes"])
    assert packed is not None
    ref = c_oracle.DequantGemv(L["codebooks"], L["codes"], L["scales"], L["bias"], 16, nthreads=0)
    y1 = hk.code1x16_matmat_packed(T["x"][:1], packed, T["codebooks"], T["scales"], T["bias"])
    check_close(y1[0].float().cpu().numpy(), ref(L["x"][0]).copy(), dtype, f"packed headline {fin}->{fout}")
    y4 = hk.code1x16_matmat_packed(T["x"], packed, T["codebooks"], T["scales"], T["bias"])
    for b in range(4):
        check_close(y4[b].float().cpu().numpy(), ref(L["x"][b]).copy(), dtype, f"packed headline {fin}->{fout} row {b}")
    assert torch.equal(y4[0], y1[0])
    if dt == "bfloat16":
        return
    # the direct (L2-gather) kernel sees the same layer: the two must agree to fp16 rounding
    yd = hk._gemv(T["x"][:1], T["codes"], T["codebooks"], T["scales"], T["bias"], "1x16")
    check_close(y1.float().cpu().numpy(), yd.float().cpu().numpy().astype(np.float64), torch.float16, "packed vs direct")
    # shared-input launch of two such layers == separate launches, bit for bit
    if fout <= 14336:
        L2 = orc.make_layer(99 + fin, fin, 4096, 1, 16, 8, batch=1, bias=False)
        T2 = to_dev(L2, torch.float16)
        pk2 = hk.prepack_1x16(T2["codes"])
        outs = hk.code1x16_matmat_packed_multi(T["x"][:2], [packed, pk2], [T["codebooks"], T2["codebooks"]],
                                               [T["scales"], T2["scales"]], [T["bias"], None])
        assert torch.equal(outs[0], y4[:2])
        assert torch.equal(outs[1], hk.code1x16_matmat_packed(T["x"][:2], pk2, T2["codebooks"], T2["scales"], None))


def test_packed_op_rejects_mismatched_dtype_and_shape(hk):
    L = orc.make_layer(5, 512, 256, 1, 16, 8, batch=1, bias=False)
    T = to_dev(L, torch.float16)
    packed = hk.prepack_1x16(T["codes"])
    with pytest.raises(NotImplementedError):   # bf16 input on an fp16 layer (the direct path and the reference raise too)
        hk.code1x16_matmat_packed(T["x"].bfloat16(), packed, T["codebooks"], T["scales"], None)
    with pytest.raises(NotImplementedError):
        hk.code1x16_matmat_packed(T["x"].float(), packed, T["codebooks"].float(), T["scales"].float(), None)
    with pytest.raises(ValueError):
        hk.code1x16_matmat_packed(T["x"][:, :256], packed, T["codebooks"], T["scales"], None)


def test_quantized_linear_uses_prepacked_path_for_large_layers(hk):
    import aqlm_amd.inference as inf

    fin, fout = 4096, 6144   # 3.1 M codes >= PREPACK_MIN_CODES
    L = orc.make_layer(55, fin, fout, 1, 16, 8, batch=2, bias=True)
    m, T = _module_from(L, 1, 16, 8, fin, fout, torch.float16)
    y1 = m(T["x"][:1])                      # single row -> prepacked kernel
    assert m._packed_codes is not None and fout * fin // 8 >= inf.PREPACK_MIN_CODES
    y2 = m(T["x"])                          # two rows: still "gemv" by the reference's rule (<= 6 rows) -> prepacked kernel
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    check_close(y1.float().cpu().numpy(), y64[:1], torch.float16, "module packed path")
    check_close(y2.float().cpu().numpy(), y64, torch.float16, "module packed path, 2 rows")
    assert torch.equal(y2[0], y1[0])
    assert torch.equal(y2, hk.code1x16_matmat_packed(T["x"], m._packed_codes, m.codebooks, m.scales, m.bias))
    assert "_packed_codes" not in m.state_dict()
    # a bf16 input on the fp16 layer must raise like the direct path and the reference, not read garbage
    with pytest.raises(NotImplementedError):
        m(T["x"][:1].bfloat16())
    # torch.compile traces the module on the packed path (dispatcher op with a fake implementation)
    cm = torch.compile(m, fullgraph=True)
    assert torch.equal(cm(T["x"][:1]), y1)


def test_drop_canonical_codes_keeps_every_path_working(hk):
    """Inference-only footprint switch: the packed buffer is lossless, so `codes` can be freed; state_dict, the
    large-batch op and backward rebuild them on demand."""
    from aqlm.checkpoint import memory_report, prepack_model

    fin, fout = 2048, 1536
    L = orc.make_layer(31, fin, fout, 1, 16, 8, batch=9, bias=True)
    m, T = _module_from(L, 1, 16, 8, fin, fout, torch.float16)
    ref_small, ref_big = m(T["x"][:3]), m(T["x"])
    holder = torch.nn.ModuleDict({"l": m})
    before = prepack_model(holder, min_codes=100_000, drop_canonical=False)
    sd_before = {k: v.clone() for k, v in m.state_dict().items()}
    after = prepack_model(holder, min_codes=100_000, drop_canonical=True)
    assert m._codes_dropped and m.codes.numel() == 0 and after["codes_dropped_layers"] == 1
    assert after["codes"] == 0 and after["code_bits_per_weight"] < before["code_bits_per_weight"]
    assert torch.equal(m(T["x"][:3]), hk.code1x16_matmat_packed(T["x"][:3], m._packed_codes, m.codebooks, m.scales, m.bias))
    check_close(m(T["x"][:3]).float().cpu().numpy(), ref_small.float().cpu().numpy().astype(np.float64), torch.float16, "dropped gemv")
    assert torch.equal(m(T["x"]), ref_big)                      # 9 rows: large-batch op on rebuilt codes
    sd = m.state_dict()
    assert set(sd) == set(sd_before) and all(torch.equal(sd[k], sd_before[k]) for k in sd)
    xg = T["x"][:2].clone().requires_grad_(True)               # backward needs the canonical codes too
    m(xg).sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()
    m2, _ = _module_from(orc.make_layer(32, fin, fout, 1, 16, 8, batch=1, bias=True), 1, 16, 8, fin, fout, torch.float16)
    m.load_state_dict(m2.state_dict())                          # new weights arrive: back to a normal module
    assert not m._codes_dropped and torch.equal(m.codes, m2.codes)
    assert torch.equal(m(T["x"][:1]), m2(T["x"][:1]))


@pytest.mark.parametrize("K,fin,fout,dt,bias", [
    (2, 4096, 4096, "float16", True),
    (2, 11008, 2050, "float16", False),   # 3 iterations per row, ragged last batch of rows
    (1, 4096, 3001, "bfloat16", True),
    (2, 4096, 8192, "bfloat16", True),
    (1, 8192, 2048, "float16", True),
])
def test_gemv_kx8_replicated_kernel(hk, K, fin, fout, dt, bias):
    """Batch-1 K x 8 g8 layers with >= 2048 rows run the replicated-LDS kernel; it must agree with the oracle and,
    closely, with the plain LDS kernel.  (Default routing: >= 4096 rows.)"""
    from aqlm_amd import _native

    dtype = tdtype(dt)
    L = orc.make_layer(8100 + fin + fout + K, fin, fout, K, 8, 8, batch=1, bias=bias,
                       float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    T = to_dev(L, dtype)
    _native.set_tuning("kx8_replicas", 2)   # force the replicated kernel whatever the row count
    try:
        y = hk.codekx8_matmat(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
        _native.set_tuning("kx8_replicas", 0)
        y_plain = hk.codekx8_matmat(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    finally:
        _native.set_tuning("kx8_replicas", 1)
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    check_close(y.float().cpu().numpy(), y64, dtype, f"replicated {K}x8 {fin}->{fout}")
    check_close(y.float().cpu().numpy(), y_plain.float().cpu().numpy().astype(np.float64), dtype, "replicated vs plain")


@pytest.mark.parametrize("g,fin,fout,dt,bias", [
    (32, 4096, 4096, "float16", True),
    (32, 11008, 1000, "float16", False),   # 344 groups: last slab half empty; ragged row ranges
    (32, 4096, 37, "bfloat16", True),
    (8, 1024, 512, "float16", True),
    (16, 2048, 300, "bfloat16", True),
])
def test_gemv_8x8_lut(hk, g, fin, fout, dt, bias):
    """Look-up-table kernel for 8 x 8-bit schemes vs the oracle and vs the direct LDS-gather kernel."""
    dtype = tdtype(dt)
    L = orc.make_layer(4400 + fin + fout + g, fin, fout, 8, 8, g, batch=1, bias=bias,
                       float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    T = to_dev(L, dtype)
    y = hk._gemv_8x8_lut(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    check_close(y.float().cpu().numpy(), y64, dtype, f"lut 8x8g{g} {fin}->{fout}")
    yd = hk._gemv(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"], "kx8")
    check_close(y.float().cpu().numpy(), yd.float().cpu().numpy().astype(np.float64), dtype, "lut vs gather kernel")
    assert torch.equal(y, hk.codekx8_matmat(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"]))  # default route
    if bias:
        yz = hk._gemv_8x8_lut(torch.zeros_like(T["x"]), T["codes"], T["codebooks"], T["scales"], T["bias"])
        assert torch.equal(yz[0], T["bias"])


@pytest.mark.parametrize("g,fin,fout,dt", [(32, 4096, 2048, "float16"), (32, 11008, 1000, "bfloat16"), (8, 2048, 300, "float16")])
def test_gemv_8x8_lut_fused_finalize(hk, g, fin, fout, dt):
    """Single-kernel form of the look-up-table matvec (fixed-point slab sums in zero-at-rest cells, one returning atomic per
    slab and row) vs the oracle and vs the two-kernel form; repeatable bit for bit; clean cells; magnitudes; NaN."""
    dtype = tdtype(dt)
    L = orc.make_layer(5500 + fin + fout + g, fin, fout, 8, 8, g, batch=1, bias=True,
                       float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    T = to_dev(L, dtype)
    run = lambda x: hk._gemv_8x8_lut(x, T["codes"], T["codebooks"], T["scales"], T["bias"])
    assert hk.USE_8X8_LUT_FUSED
    y = run(T["x"])
    cells = hk.accumulator_cells()
    assert cells and all(int(c.abs().max()) == 0 for c in cells), "cells must be zero when the kernel has finished"
    hk.USE_8X8_LUT_FUSED = False
    try:
        y_two = run(T["x"])
    finally:
        hk.USE_8X8_LUT_FUSED = True
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    check_close(y.float().cpu().numpy(), y64, dtype, "lut fused finalize")
    check_close(y.float().cpu().numpy(), y_two.double().cpu().numpy(), dtype, "lut fused vs two-kernel")
    for _ in range(10):
        assert torch.equal(run(T["x"]), y)
    for factor, what in ((2.0 ** 10, "|x| large"), (2.0 ** -12, "|x| small")):
        xs = T["x"] * factor
        ys64 = orc.dequantize_gemm(xs.double().cpu().numpy(), L["codes"], T["codebooks"].double().cpu().numpy(),
                                   (T["scales"] * 2.0 ** -6).double().cpu().numpy(), None)
        ys = hk._gemv_8x8_lut(xs, T["codes"], T["codebooks"], T["scales"] * 2.0 ** -6, None)
        check_close(ys.float().cpu().numpy(), ys64, dtype, f"lut fused finalize, {what}")
    xn = T["x"].clone()
    xn[0, 5] = float("nan")
    assert torch.isnan(run(xn)).all() and all(int(c.abs().max()) == 0 for c in cells)
    assert torch.equal(run(T["x"]), y)
    # hipGraph: a stream that has run the op captures the single-kernel form; a fresh one falls back to two kernels
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(T["x"])
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph, stream=s):
        yg = run(T["x"])
    gph.replay(); gph.replay()
    torch.cuda.synchronize()
    assert torch.equal(yg, y)
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    gph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph2, stream=s2):
        yg2 = run(T["x"])
    gph2.replay()
    torch.cuda.synchronize()
    assert torch.equal(yg2, y_two)


def test_ops_trace_under_torch_compile(hk):
    """torch.ops.aqlm.* carry fake impls (like cuda_kernel.py:20-22), so Dynamo traces a QuantizedLinear without
    graph breaks -- what the reference's CUDA-graph notebook relies on (notebooks/aqlm_cuda_graph.ipynb)."""
    import torch._dynamo as dynamo

    L = orc.make_layer(71, 1024, 256, 1, 16, 8, batch=1, bias=True)
    m, T = _module_from(L, 1, 16, 8, 1024, 256, torch.float16)
    m(T["x"])  # resolve the kernels eagerly first (lazy prepare_matmul_op)
    explanation = dynamo.explain(lambda x: torch.ops.aqlm.code1x16_matmat(x, m.codes, m.codebooks, m.scales, m.bias))(T["x"])
    assert explanation.graph_break_count == 0, explanation.break_reasons
    ops = [str(n.target) for g in explanation.graphs for n in g.graph.nodes if n.op == "call_function"]
    assert any("code1x16_matmat" in o for o in ops), ops
    compiled = torch.compile(lambda x: torch.ops.aqlm.code1x16_matmat(x, m.codes, m.codebooks, m.scales, m.bias) * 2.0,
                             backend="eager", fullgraph=True)
    y = compiled(T["x"])
    ref = torch.ops.aqlm.code1x16_matmat(T["x"], m.codes, m.codebooks, m.scales, m.bias) * 2.0
    assert torch.equal(y, ref)


# ------------------------------------------------------------------ shared-input launches
def _layers_sharing_x(seed, fin, fouts, g, dtype, batch, biases):
    fd = np.float16 if dtype == torch.float16 else "bfloat16"
    Ls = [orc.make_layer(seed + 17 * k, fin, fo, 1, 16, g, batch=batch, bias=b, float_dtype=fd)
          for k, (fo, b) in enumerate(zip(fouts, biases))]
    Ts = [to_dev(L, dtype) for L in Ls]
    return Ls, Ts, Ts[0]["x"]


@pytest.mark.parametrize("fin,fouts,g,dt,batch", [
    (4096, (4096, 1024, 1024), 8, "float16", 1),     # Llama-3-8B q/k/v
    (1024, (37, 512), 8, "bfloat16", 3),             # ragged rows, batch 3
    (2048, (256, 256, 64, 1000), 16, "float16", 8),  # g16, four segments, full batch
    (520, (128, 96), 8, "float16", 2),               # in_groups % 8 != 0 -> per-segment generic launches
    (4096, (512,), 8, "float16", 11),                # one segment, two batch chunks
])
def test_gemv_1x16_multi_is_bit_identical_to_separate_launches(hk, fin, fouts, g, dt, batch):
    dtype = tdtype(dt)
    biases = [k % 2 == 0 for k in range(len(fouts))]
    Ls, Ts, x = _layers_sharing_x(4000 + fin, fin, fouts, g, dtype, batch, biases)
    outs = torch.ops.aqlm.code1x16_matmat_multi(x, [T["codes"] for T in Ts], [T["codebooks"] for T in Ts],
                                                [T["scales"] for T in Ts], [T["bias"] for T in Ts])
    assert len(outs) == len(fouts)
    for L, T, y in zip(Ls, Ts, outs):
        single = hk._gemv(x, T["codes"], T["codebooks"], T["scales"], T["bias"], "1x16")   # the matvec kernel, launch by launch
        assert y.shape == single.shape and torch.equal(y, single)
        y64 = orc.dequantize_gemm(Ls[0]["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
        check_close(y.float().cpu().numpy(), y64, dtype, f"multi 1x16g{g} {fin}->{L['codes'].shape[0]}")
        # the raw op itself: the same kernel up to 6 rows, the MFMA kernel from 7 rows on -- the oracle holds either way
        raw = hk.code1x16_matmat(x, T["codes"], T["codebooks"], T["scales"], T["bias"])
        check_close(raw.float().cpu().numpy(), y64, dtype, f"raw op, {batch} rows")
        if batch < hk.MATMAT_GEMM_MIN_ROWS:
            assert torch.equal(raw, single)


@pytest.mark.parametrize("fin,fouts,dt", [
    (4096, (1024, 1024), "float16"),
    (4096, (300, 2048, 77), "bfloat16"),
    (11008, (256, 512, 128, 64), "float16"),
])
def test_gemv_1x16_packed_multi_is_bit_identical_to_separate_launches(hk, fin, fouts, dt):
    dtype = tdtype(dt)
    biases = [k % 2 == 1 for k in range(len(fouts))]
    Ls, Ts, x = _layers_sharing_x(5000 + fin, fin, fouts, 8, dtype, 1, biases)
    packed = [hk.prepack_1x16(T["codes"]) for T in Ts]
    outs = hk.code1x16_matmat_packed_multi(x, packed, [T["codebooks"] for T in Ts], [T["scales"] for T in Ts],
                                           [T["bias"] for T in Ts])
    for L, T, pk, y, fo in zip(Ls, Ts, packed, outs, fouts):
        single = hk.code1x16_matmat_packed(x, pk, T["codebooks"], T["scales"], T["bias"])
        assert torch.equal(y, single)
        y64 = orc.dequantize_gemm(Ls[0]["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
        check_close(y.float().cpu().numpy(), y64, dtype, f"packed multi {fin}->{fo}")


@pytest.mark.parametrize("K,fin,fouts,dt,batch", [
    (2, 4096, (4096, 1024, 1024), "float16", 1),   # 6144 rows -> replicated-LDS kernel for every segment
    (2, 4096, (11008, 11008), "bfloat16", 1),      # gate/up
    (1, 2048, (300, 37, 4096), "float16", 1),      # ragged segments, 1x8
    (2, 1024, (256, 512), "float16", 1),           # < 4096 rows in total -> plain LDS kernel, one launch
    (2, 4096, (4096, 4096), "float16", 5),         # batch > 1 -> plain LDS kernel, batch chunks 4 + 1
    (1, 12352, (512, 512), "float16", 1),          # 4 unit iterations: replicated kernel refuses, plain kernel runs
    (2, 4096, (4096, 1024, 1024), "float16", 4),   # round 5: 2+ rows -> ONE launch of the X-resident MFMA kernel over all layers
    (2, 4096, (11008, 11008), "bfloat16", 8),      # gate / up, 64 KiB of X next to two layers' codebooks
    (1, 2048, (300, 37, 4096, 16), "float16", 16), # four ragged layers, 1x8, the full first batch tile
    (2, 11008, (4096, 4096), "float16", 3),        # 66 KiB of X: fits next to 32 KiB of codebooks
    (2, 11008, (4096, 4096), "float16", 7),        # 154 KiB of X: does not fit -> one launch per layer (6 rows would; 7 do not)
])
def test_gemv_kx8_multi_matches_separate_launches(hk, K, fin, fouts, dt, batch):
    dtype = tdtype(dt)
    fd = np.float16 if dtype == torch.float16 else "bfloat16"
    Ls = [orc.make_layer(6000 + fin + 13 * k, fin, fo, K, 8, 8, batch=batch, bias=(k % 2 == 0), float_dtype=fd)
          for k, fo in enumerate(fouts)]
    Ts = [to_dev(L, dtype) for L in Ls]
    x = Ts[0]["x"]
    outs = torch.ops.aqlm.codekx8_matmat_multi(x, [T["codes"] for T in Ts], [T["codebooks"] for T in Ts],
                                               [T["scales"] for T in Ts], [T["bias"] for T in Ts])
    for L, T, y in zip(Ls, Ts, outs):
        y64 = orc.dequantize_gemm(Ls[0]["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
        check_close(y.float().cpu().numpy(), y64, dtype, f"multi {K}x8g8 {fin}->{L['codes'].shape[0]}")
        single = hk.codekx8_matmat(x, T["codes"], T["codebooks"], T["scales"], T["bias"])
        check_close(y.float().cpu().numpy(), single.float().cpu().numpy().astype(np.float64), dtype, "multi vs single")
        if batch > 1:   # the fused MFMA kernel serves both forms: a tile's arithmetic does not know about the other layers
            assert torch.equal(y, single), "shared-input launch differs from the layer's own launch"
        if T["bias"] is not None:   # zero input -> exactly the bias
            yz = torch.ops.aqlm.codekx8_matmat_multi(torch.zeros_like(x), [T["codes"]], [T["codebooks"]], [T["scales"]], [T["bias"]])[0]
            assert torch.equal(yz, T["bias"].expand_as(yz))
    # deterministic
    again = torch.ops.aqlm.codekx8_matmat_multi(x, [T["codes"] for T in Ts], [T["codebooks"] for T in Ts],
                                                [T["scales"] for T in Ts], [T["bias"] for T in Ts])
    assert all(torch.equal(a, b) for a, b in zip(outs, again))


@pytest.mark.parametrize("g,fin,fouts,dt", [(32, 4096, (4096, 1024, 1024), "float16"), (32, 1024, (300, 77), "bfloat16"),
                                           (8, 512, (256, 256, 256, 40), "float16"), (16, 2048, (1000, 1000), "float16")])
def test_gemv_8x8_lut_multi_is_bit_identical_to_separate_launches(hk, g, fin, fouts, dt):
    dtype = tdtype(dt)
    fd = np.float16 if dtype == torch.float16 else "bfloat16"
    Ls = [orc.make_layer(6600 + fin + 7 * k, fin, fo, 8, 8, g, batch=1, bias=(k % 2 == 0), float_dtype=fd)
          for k, fo in enumerate(fouts)]
    Ts = [to_dev(L, dtype) for L in Ls]
    x = Ts[0]["x"]
    outs = torch.ops.aqlm.codekx8_matmat_multi(x, [T["codes"] for T in Ts], [T["codebooks"] for T in Ts],
                                               [T["scales"] for T in Ts], [T["bias"] for T in Ts])
    for L, T, y in zip(Ls, Ts, outs):
        assert torch.equal(y, hk.codekx8_matmat(x, T["codes"], T["codebooks"], T["scales"], T["bias"]))
        y64 = orc.dequantize_gemm(Ls[0]["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
        check_close(y.float().cpu().numpy(), y64, dtype, f"lut multi 8x8g{g} {fin}->{L['codes'].shape[0]}")


@pytest.mark.parametrize("g,fin,fout,dt,bias", [
    (32, 4096, 4096, "float16", True),
    (32, 11008, 1000, "float16", False),   # 344 groups: three 128-group slabs per codebook, the last one 88 wide; ragged rows
    (32, 4096, 37, "bfloat16", True),
    (8, 1024, 512, "float16", True),       # 128 groups of 8
    (16, 2080, 300, "bfloat16", True),     # 130 groups: rows of the planes padded to 132 bytes
    (32, 8192, 28672, "float16", False),   # 896 rows per workgroup: the hand-in pass covers several rows per thread
])
def test_gemv_8x8_lut_planar(hk, g, fin, fout, dt, bias):
    """Planar code layout of the 8x8 look-up-table matvec: lossless re-layout, single-kernel and two-kernel forms vs the oracle and
    vs the canonical-layout kernel, both workgroup sizes, repeatable bit for bit, zero input == bias."""
    from aqlm_amd import _native

    dtype = tdtype(dt)
    L = orc.make_layer(4700 + fin + fout + g, fin, fout, 8, 8, g, batch=1, bias=bias,
                       float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    T = to_dev(L, dtype)
    planar = hk.planar_8x8_pack(T["codes"], g, codebooks=T["codebooks"])
    assert planar is not None and planar.numel() == 8 * fout * ((fin // g + 3) // 4 * 4) and planar.codebook_absmax > 0
    assert torch.equal(hk.planar_8x8_unpack(planar), T["codes"])
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    y_canon = hk._gemv_8x8_lut(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    outs = {}
    for waves in (16, 8):
        _native.set_tuning("lut_waves", waves)
        try:
            y = hk.code8x8_matmat_planar(T["x"], planar, T["codebooks"], T["scales"], T["bias"])
            check_close(y.float().cpu().numpy(), y64, dtype, f"planar lut 8x8g{g} {fin}->{fout}, {waves} waves")
            for _ in range(3):
                assert torch.equal(hk.code8x8_matmat_planar(T["x"], planar, T["codebooks"], T["scales"], T["bias"]), y)
            keep, planar.codebook_absmax = planar.codebook_absmax, 0.0   # unknown bound -> two-kernel form
            try:
                planar._range_of = (T["codebooks"].data_ptr(), hk._version(T["codebooks"]))
                y2 = hk.code8x8_matmat_planar(T["x"], planar, T["codebooks"], T["scales"], T["bias"])
            finally:
                planar.codebook_absmax = keep
            check_close(y.float().cpu().numpy(), y2.double().cpu().numpy(), dtype, "planar fused vs two-kernel")
            yc = hk._gemv_8x8_lut(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
            check_close(yc.float().cpu().numpy(), y64, dtype, f"canonical lut, {waves} waves")
            outs[waves] = y
        finally:
            _native.set_tuning("lut_waves", 0)
    check_close(outs[16].float().cpu().numpy(), y_canon.double().cpu().numpy(), dtype, "planar vs canonical layout")
    assert all(int(c.abs().max()) == 0 for c in hk.accumulator_cells()), "cells must be zero when the kernels have finished"
    if bias:
        yz = hk.code8x8_matmat_planar(torch.zeros_like(T["x"]), planar, T["codebooks"], T["scales"], T["bias"])
        assert torch.equal(yz[0], T["bias"])
    xn = T["x"].clone()
    xn[0, 3] = float("inf")
    assert torch.isnan(hk.code8x8_matmat_planar(xn, planar, T["codebooks"], T["scales"], T["bias"])).all()
    assert torch.equal(hk.code8x8_matmat_planar(T["x"], planar, T["codebooks"], T["scales"], T["bias"]), outs[16])


@pytest.mark.parametrize("g,fin,fouts,dt", [(32, 4096, (4096, 1024, 1024), "float16"), (32, 11008, (300, 77), "bfloat16"),
                                           (8, 1024, (256, 256, 256, 40), "float16")])
def test_gemv_8x8_lut_planar_multi_is_bit_identical_to_separate_launches(hk, g, fin, fouts, dt):
    dtype = tdtype(dt)
    fd = np.float16 if dtype == torch.float16 else "bfloat16"
    Ls = [orc.make_layer(6700 + fin + 7 * k, fin, fo, 8, 8, g, batch=1, bias=(k % 2 == 0), float_dtype=fd)
          for k, fo in enumerate(fouts)]
    Ts = [to_dev(L, dtype) for L in Ls]
    x = Ts[0]["x"]
    planars = [hk.planar_8x8_pack(T["codes"], g, codebooks=T["codebooks"]) for T in Ts]
    outs = hk.code8x8_matmat_planar_multi(x, planars, [T["codebooks"] for T in Ts], [T["scales"] for T in Ts], [T["bias"] for T in Ts])
    for L, T, pl, y in zip(Ls, Ts, planars, outs):
        assert torch.equal(y, hk.code8x8_matmat_planar(x, pl, T["codebooks"], T["scales"], T["bias"]))
        y64 = orc.dequantize_gemm(Ls[0]["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
        check_close(y.float().cpu().numpy(), y64, dtype, f"planar lut multi 8x8g{g} {fin}->{L['codes'].shape[0]}")


@pytest.mark.parametrize("g,fin,fout,dt", [(32, 4096, 4096, "float16"), (32, 11008, 1000, "bfloat16"), (8, 1024, 512, "float16"),
                                           (16, 2080, 300, "float16")])
def test_gemv_8x8_lut_rows(hk, g, fin, fout, dt, monkeypatch):
    """2..8 input rows of an 8-codebook scheme in ONE launch of the look-up-table kernel (aqlm_hip_gemv_8x8_lut_batch; the
    reference loops its generic gemv over the rows, triton_kernel.py:161-182): planar and canonical codes vs the fp64 oracle,
    every row bit-identical to the same row launched alone, through the raw op's routing and the shared-input ops; cells zero.
    (The raw op sends 3+ rows of 8x8 g32 to the fused MFMA kernel since round 5 -- test_fused_8x8g32_mfma; switched off here.)"""
    monkeypatch.setattr(hk, "USE_FUSED_8X8_MFMA", False)
    dtype = tdtype(dt)
    L = orc.make_layer(5100 + fin + fout + g, fin, fout, 8, 8, g, batch=8, bias=True,
                       float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    T = to_dev(L, dtype)
    planar = hk.planar_8x8_pack(T["codes"], g, codebooks=T["codebooks"])
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    alone_p = [hk.code8x8_matmat_planar(T["x"][b:b + 1].contiguous(), planar, T["codebooks"], T["scales"], T["bias"]) for b in range(8)]
    alone_c = [hk._gemv_8x8_lut(T["x"][b:b + 1].contiguous(), T["codes"], T["codebooks"], T["scales"], T["bias"]) for b in range(8)]
    for B in (2, 3, 6, 8):
        yp = hk.code8x8_matmat_planar(T["x"][:B], planar, T["codebooks"], T["scales"], T["bias"])
        yc = hk.codekx8_matmat(T["x"][:B], T["codes"], T["codebooks"], T["scales"], T["bias"])
        assert yp.shape == (B, fout) and yc.shape == (B, fout)
        check_close(yp.float().cpu().numpy(), y64[:B], dtype, f"planar lut, {B} rows")
        check_close(yc.float().cpu().numpy(), y64[:B], dtype, f"canonical lut, {B} rows")
        for b in range(B):
            assert torch.equal(yp[b], alone_p[b][0]) and torch.equal(yc[b], alone_c[b][0]), (B, b)
    # strided rows (a slice of a wider activation buffer) and a 3-D batch shape
    wide = torch.zeros(4, 2 * fin, dtype=dtype, device=DEV)
    wide[:, :fin] = T["x"][:4]
    ys = hk.code8x8_matmat_planar(wide[:, :fin], planar, T["codebooks"], T["scales"], T["bias"])
    assert all(torch.equal(ys[b], alone_p[b][0]) for b in range(4))
    y3 = hk.code8x8_matmat_planar(T["x"][:6].reshape(2, 3, fin), planar, T["codebooks"], T["scales"], T["bias"])
    assert y3.shape == (2, 3, fout) and torch.equal(y3.reshape(6, fout)[5], alone_p[5][0])
    # shared-input ops with 2+ rows: one multi-row launch per layer, same bits
    outs = hk.code8x8_matmat_planar_multi(T["x"][:3], [planar, planar], [T["codebooks"]] * 2, [T["scales"]] * 2, [T["bias"], None])
    assert torch.equal(outs[0][2], alone_p[2][0]) and outs[1].shape == (3, fout)
    outs = hk.codekx8_matmat_multi(T["x"][:3], [T["codes"]] * 2, [T["codebooks"]] * 2, [T["scales"]] * 2, [T["bias"], None])
    assert torch.equal(outs[0][2], alone_c[2][0])
    # under hipGraph capture on a stream whose cells exist (one eager call on it): the multi-row launch itself is captured
    sx = T["x"][:4].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        hk.code8x8_matmat_planar(sx, planar, T["codebooks"], T["scales"], T["bias"])
        with torch.cuda.graph(g_, stream=side):
            yg = hk.code8x8_matmat_planar(sx, planar, T["codebooks"], T["scales"], T["bias"])
    g_.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(yg[b], alone_p[b][0]) for b in range(4))
    # ... and on a stream without cells the capture takes the two-kernel form row by row: same values to fp32 rounding
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        yg2 = hk.code8x8_matmat_planar(sx, planar, T["codebooks"], T["scales"], T["bias"])
    g2.replay()
    torch.cuda.synchronize()
    check_close(yg2.float().cpu().numpy(), y64[:4], dtype, "captured without cells")
    assert all(int(c.abs().max()) == 0 for c in hk.accumulator_cells()), "cells must be zero when the kernels have finished"


@pytest.mark.parametrize("planar", [True, False])
def test_gemv_8x8_lut_many_rows_per_workgroup(hk, planar):
    """More rows per workgroup than one staging pass holds (2048): the walk / hand-in loop runs several passes with the staging
    area reused between barriers.  100 000 rows x 128 groups (3125 rows per workgroup in both layouts); checked against the C oracle (streams the rows, no dense W)."""
    fin, fout, g = 4096, 100_000, 32
    L = orc.make_layer(4900, fin, fout, 8, 8, g, batch=1, bias=True)
    T = to_dev(L, torch.float16)
    ref = c_oracle.DequantGemv(L["codebooks"], L["codes"], L["scales"], L["bias"], 8, nthreads=0)
    y64 = ref(L["x"][0]).copy()
    if planar:
        pl = hk.planar_8x8_pack(T["codes"], g, codebooks=T["codebooks"])
        y = hk.code8x8_matmat_planar(T["x"], pl, T["codebooks"], T["scales"], T["bias"])
        assert torch.equal(hk.code8x8_matmat_planar(T["x"], pl, T["codebooks"], T["scales"], T["bias"]), y)
    else:
        y = hk._gemv_8x8_lut(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    check_close(y[0].float().cpu().numpy(), y64, torch.float16, f"lut, 100000 rows, planar={planar}")
    assert all(int(c.abs().max()) == 0 for c in hk.accumulator_cells())


def test_8x8_module_uses_planar_codes_and_can_drop_the_canonical_ones(hk):
    """QuantizedLinear of an 8x8 scheme: single-row calls run on the planar copy, 2..6 rows and backward on the canonical codes;
    dropping the canonical codes keeps 2.0 bits per weight of codes resident and state_dict() lossless; fused q/k/v in one launch."""
    import aqlm
    from aqlm.checkpoint import prepack_model

    fin = 4096
    holder = torch.nn.Module()
    Ls = {}
    for k, (n, fo) in enumerate([("q_proj", 1024), ("k_proj", 256), ("v_proj", 256)]):
        Ls[n] = orc.make_layer(4800 + k, fin, fo, 8, 8, 32, batch=3, bias=(k == 0))
        m, _ = _module_from(Ls[n], 8, 8, 32, fin, fo, torch.float16)
        setattr(holder, n, m)
    T = to_dev(Ls["q_proj"], torch.float16)
    x1, x3 = T["x"][:1].contiguous(), T["x"]
    with torch.no_grad():
        y1 = holder.q_proj(x1)
        assert isinstance(holder.q_proj._packed_codes, hk.PlanarCodes)
        assert torch.equal(y1, hk.code8x8_matmat_planar(x1, holder.q_proj._packed_codes, T["codebooks"], T["scales"], T["bias"]))
        from aqlm_amd import _front
        if _front.available():   # the compiled lane serves single rows on the same kernel, and hands anything else back
            assert holder.q_proj._fast is not None and holder.q_proj._fast.kind == _front.KIND_LUT_PLANAR_8X8
            y2f = holder.q_proj._fast(x3[:2])  # 2 rows: one launch of rows x the single-row workgroups (round 5)
            assert torch.equal(holder.q_proj._fast(x1), y1) and y2f is not None and torch.equal(y2f[:1], y1)
            y3f = holder.q_proj._fast(x3)   # 3+ rows of 8x8 g32: the lane launches the fused MFMA kernel itself (same bits as the op)
            assert y3f is not None and torch.equal(y3f, hk.codekx8_matmat(x3, T["codes"], T["codebooks"], T["scales"], T["bias"]))
            assert torch.equal(holder.q_proj(x1), y1)
        y64 = orc.dequantize_gemm(Ls["q_proj"]["x"], Ls["q_proj"]["codes"], Ls["q_proj"]["codebooks"], Ls["q_proj"]["scales"], Ls["q_proj"]["bias"])
        check_close(y1.float().cpu().numpy(), y64[:1], torch.float16, "8x8 module, one row")
        check_close(holder.q_proj(x3).float().cpu().numpy(), y64, torch.float16, "8x8 module, three rows")
        sd = {k: v.clone() for k, v in holder.q_proj.state_dict().items()}
        rep = prepack_model(holder, min_codes=100_000, drop_canonical=True)
        assert holder.q_proj._codes_dropped and holder.v_proj._codes_dropped and rep["codes"] == 0 and abs(rep["code_bits_per_weight"] - 2.0) < 1e-9
        assert torch.equal(holder.q_proj(x1), y1)
        check_close(holder.q_proj(x3).float().cpu().numpy(), y64, torch.float16, "8x8 module, three rows, codes dropped")
        sd2 = holder.q_proj.state_dict()
        assert all(torch.equal(sd[k], sd2[k]) for k in sd)
        groups = aqlm.fuse_shared_input_linears(holder)
        assert len(groups) == 1
        want = {n: getattr(holder, n)._packed_codes for n in Ls}
        outs = {n: getattr(holder, n)(x1) for n in Ls}
        assert groups[0].launches == 1 and groups[0].served == 2
        for n in Ls:
            Tn = to_dev(Ls[n], torch.float16)
            assert torch.equal(outs[n], hk.code8x8_matmat_planar(x1, want[n], Tn["codebooks"], Tn["scales"], Tn["bias"])), n
    # torch.compile traces a module on planar codes through the dispatcher op (fake implementation), also without canonical codes
    with torch.no_grad():
        solo, _ = _module_from(Ls["q_proj"], 8, 8, 32, fin, 1024, torch.float16)
        y_solo = solo(x1)
        assert isinstance(solo._packed_codes, hk.PlanarCodes)
        solo.drop_canonical_codes()
        cm = torch.compile(solo, fullgraph=True)
        assert torch.equal(cm(x1), y_solo)
    holder.q_proj.restore_canonical_codes()
    xg = x1.clone().requires_grad_(True)
    holder.q_proj(xg).sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


def test_gemv_kx8_multi_other_schemes_fall_back_per_segment(hk):
    Ls = [orc.make_layer(6500 + k, 1024, fo, 4, 8, 16, batch=2, bias=True) for k, fo in enumerate((128, 320))]
    Ts = [to_dev(L, torch.float16) for L in Ls]
    outs = torch.ops.aqlm.codekx8_matmat_multi(Ts[0]["x"], [T["codes"] for T in Ts], [T["codebooks"] for T in Ts],
                                               [T["scales"] for T in Ts], [T["bias"] for T in Ts])
    for T, y in zip(Ts, outs):
        assert torch.equal(y, hk.codekx8_matmat(Ts[0]["x"], T["codes"], T["codebooks"], T["scales"], T["bias"]))


def test_fused_2x8_modules_match_unfused(hk):
    import aqlm

    fin = 2048
    mods, x = {}, None
    for k, (n, fo) in enumerate([("q_proj", 2048), ("k_proj", 2048), ("v_proj", 2048), ("gate_proj", 5504), ("up_proj", 5504)]):
        L = orc.make_layer(980 + k, fin, fo, 2, 8, 8, batch=3, bias=(k == 0))
        mods[n], T = _module_from(L, 2, 8, 8, fin, fo, torch.float16)
        x = T["x"] if x is None else x
    holder = torch.nn.Module()
    for n, m in mods.items():
        setattr(holder, n, m)
    with torch.no_grad():
        for rows in (1, 3):
            h = x[:rows].clone()
            ref = {n: m(h) for n, m in mods.items()}
            groups = aqlm.fuse_shared_input_linears(holder)
            assert [len(g.members) for g in groups] == [3, 2]
            got = {n: getattr(holder, n)(h) for n in mods}
            assert [(g.launches, g.served) for g in groups] == [(1, 2), (1, 1)]
            aqlm.unfuse_shared_input_linears(holder)
            for n in mods:
                a, b = got[n].float().cpu().numpy(), ref[n].float().cpu().numpy().astype(np.float64)
                check_close(a, b, torch.float16, f"fused 2x8 {n} rows={rows}")


class _FakeDecoderBlock(torch.nn.Module):
    """Calls its projections the way Hugging Face's Llama code does: q, k, v on one tensor; gate, up on another."""

    def __init__(self, mods):
        super().__init__()
        for n, m in mods.items():
            setattr(self, n, m)

    def forward(self, h):
        norm = lambda t: torch.nn.functional.normalize(t.float(), dim=-1).to(h.dtype)  # keeps fp16 finite
        q, k, v = self.q_proj(h), self.k_proj(h), self.v_proj(h)
        h2 = norm(q + torch.nn.functional.pad(k + v, (0, q.shape[-1] - k.shape[-1])))
        return self.down_proj(norm(torch.nn.functional.silu(self.gate_proj(h2)).float() * self.up_proj(h2).float())), (q, k, v)


def test_fused_shared_input_modules_match_unfused(hk):
    import aqlm
    import aqlm_amd.inference as inf

    fin, kv, inter = 1024, 256, 3072
    shapes = dict(q_proj=(fin, fin), k_proj=(fin, kv), v_proj=(fin, kv), gate_proj=(fin, inter), up_proj=(fin, inter),
                  down_proj=(inter, fin))
    old = inf.PREPACK_MIN_CODES
    inf.PREPACK_MIN_CODES = 300_000   # gate/up/down (393 216 codes) take the prepacked route, q/k/v the direct one
    try:
        mods = {}
        for k, (n, (fi, fo)) in enumerate(shapes.items()):
            L = orc.make_layer(900 + k, fi, fo, 1, 16, 8, batch=4, bias=(k % 2 == 0))
            mods[n], T = _module_from(L, 1, 16, 8, fi, fo, torch.float16)
            if n == "q_proj":
                x = T["x"]
        block = _FakeDecoderBlock(mods)
        with torch.no_grad():
            for rows in (1, 4):
                h = x[:rows].clone()
                ref, (q0, k0, v0) = block(h)
                groups = aqlm.fuse_shared_input_linears(block)
                assert [len(g.members) for g in groups] == [3, 2]
                assert aqlm.fuse_shared_input_linears(block) == []  # idempotent
                out, (q1, k1, v1) = block(h)
                assert torch.isfinite(ref).all()
                assert torch.equal(out, ref) and torch.equal(q0, q1) and torch.equal(k0, k1) and torch.equal(v0, v1)
                assert [(g.launches, g.served) for g in groups] == [(1, 2), (1, 1)]
                from aqlm_amd import _front
                if _front.available():  # both groups launch through the compiled front end (csrc_front FastGroup): gate / up on
                    assert groups[1]._fast_group is not None and groups[0]._fast_group is not None  # the prepacked, q / k / v on the direct entry
                assert all(g._input is None and not g._pending for g in groups)  # nothing kept alive
                # a different tensor object (even with equal values) is a new launch; a repeated call too
                out2, _ = block(h.clone())
                assert torch.equal(out2, ref) and groups[0].launches == 2
                y_a = block.q_proj(h)
                y_b = block.q_proj(h)          # same member twice: second call must recompute, not starve
                assert torch.equal(y_a, q0) and torch.equal(y_b, q0)
                h_mut = h.clone()
                k_first = block.k_proj(h_mut)  # parks q and v for h_mut ...
                h_mut.mul_(2.0)                # ... but the tensor is modified in place before they are fetched
                assert torch.equal(block.q_proj(h_mut), block.q_proj(h_mut.clone()))
                assert torch.equal(k_first, k0)
                aqlm.unfuse_shared_input_linears(block)
                assert all(m._shared_input_group is None for m in mods.values())
        assert mods["gate_proj"]._packed_codes is not None and mods["q_proj"]._packed_codes is None
        # > GEMV_MAX_ROWS rows and inputs that need grad bypass the group
        groups = aqlm.fuse_shared_input_linears(block)
        big = torch.randn(16, fin, dtype=torch.float16, device=DEV)
        block.q_proj(big)
        xg = x[:2].clone().requires_grad_(True)
        block.q_proj(xg).float().sum().backward()
        assert xg.grad is not None and groups[0].launches == 0
    finally:
        inf.PREPACK_MIN_CODES = old


def test_pipelined_launch_randomized(hk):
    """Seeded random shared-input groups (2..4 segments, mixed sizes, forced wave counts 3..16 -> both DMA modes, deferred and
    immediate hand-shakes, waves without rows) through the pipelined kernel: every output bit-identical to the segment's own
    single-layer launch, repeatedly (accumulator cells zero at rest).  AQLM_TEST_PIPE_CASES raises the case count."""
    import os
    import random

    from aqlm_amd import _native

    rng = random.Random(20260924)
    ncases = int(os.environ.get("AQLM_TEST_PIPE_CASES", "20"))
    for case in range(ncases):
        fin = rng.choice([512, 1024, 2048, 4096, 4096, 8192])
        nseg = rng.randint(2, 4)
        outs = [rng.choice([64, 200, 512, 1024, 1536, 2048, 4096, 6000, 11008]) for _ in range(nseg)]
        waves = rng.choice([0, 0, 0, 3, 6, 9, 13, 14, 15, 16])
        dt = rng.choice(["float16", "float16", "bfloat16"])
        g = rng.choice([8, 8, 16])  # 16-element codebook vectors: the second build of the same kernel
        dtype = tdtype(dt)
        Ls = [orc.make_layer(5000 + 17 * case + i, fin, o, 1, 16, g, batch=1, bias=bool((case + i) % 2),
                             float_dtype=np.float16 if dt == "float16" else "bfloat16") for i, o in enumerate(outs)]
        Ts = [to_dev(L, dtype) for L in Ls]
        x = Ts[0]["x"]
        _native.set_tuning("packed_waves", waves)
        try:
            packed = [hk.prepack_1x16(T["codes"], g, codebooks=T["codebooks"]) for T in Ts]
        finally:
            _native.set_tuning("packed_waves", 0)
        if any(p is None for p in packed):
            continue
        args = (x, packed, [T["codebooks"] for T in Ts], [T["scales"] for T in Ts], [T["bias"] for T in Ts])
        singles = [hk.code1x16_matmat_packed(x, p, T["codebooks"], T["scales"], T["bias"]) for p, T in zip(packed, Ts)]
        for rep in range(3):
            piped = hk.code1x16_matmat_packed_multi(*args)
            for k in range(nseg):
                assert torch.equal(piped[k], singles[k]), f"case {case}: fin {fin} g {g} outs {outs} waves {waves} {dt}, segment {k}, repeat {rep}"
        del Ls, Ts, packed, singles, piped
    torch.cuda.synchronize()


def test_compiled_group_launch_equals_separate_modules(hk):
    """q / k / v of Llama-3-8B size through the compiled group launch (front.cpp FastGroup -> the pipelined kernel): the
    outputs are those of the unfused modules, bit for bit; a parameter written in place sends the call back to the Python
    path, which rebuilds the lanes; 7 rows bypass the group."""
    import aqlm
    from aqlm_amd import _front

    if not _front.available():
        pytest.skip("aqlm_amd/_aqlm_front.so not built")
    fin = 4096
    mods = {}
    for k, (n, fo) in enumerate((("q_proj", 4096), ("k_proj", 1024), ("v_proj", 1024))):
        L = orc.make_layer(1300 + k, fin, fo, 1, 16, 8, batch=3, bias=(k == 1))
        mods[n], T = _module_from(L, 1, 16, 8, fin, fo, torch.float16)
        if k == 0:
            x = T["x"]
    with torch.no_grad():
        ref = {n: m(x) for n, m in mods.items()}          # unfused (each through its own compiled lane)
        holder = torch.nn.Module()
        for n, m in mods.items():
            setattr(holder, n, m)
        groups = aqlm.fuse_shared_input_linears(holder)
        assert len(groups) == 1 and len(groups[0].members) == 3
        for rows in (1, 3):
            h = x[:rows].clone()
            out = {n: m(h) for n, m in mods.items()}
            assert groups[0]._fast_group is not None
            for n in mods:
                assert torch.equal(out[n], ref[n][:rows]), n
        assert groups[0].launches == 2 and groups[0].served == 4
        # an in-place update of one member's scales: the lanes notice, the Python path rebuilds them, results follow
        mods["k_proj"].scales.mul_(2.0)
        h = x[:1].clone()
        out = {n: m(h) for n, m in mods.items()}
        assert torch.equal(out["q_proj"], ref["q_proj"][:1])
        assert torch.equal(out["k_proj"], mods["k_proj"].__class__.forward(mods["k_proj"], h.clone()))
        assert not torch.equal(out["k_proj"], ref["k_proj"][:1])
        aqlm.unfuse_shared_input_linears(holder)


def test_fused_group_inside_hipgraph(hk):
    """Shared-input launches are capturable: replaying the graph on new input data reproduces the eager result."""
    import aqlm

    fin = 1024
    mods = {}
    for k, (n, fo) in enumerate([("q_proj", 1024), ("k_proj", 256), ("v_proj", 256)]):
        L = orc.make_layer(950 + k, fin, fo, 1, 16, 8, batch=1, bias=True)
        mods[n], T = _module_from(L, 1, 16, 8, fin, fo, torch.float16)
    holder = torch.nn.Module()
    for n, m in mods.items():
        setattr(holder, n, m)
    aqlm.fuse_shared_input_linears(holder)
    static_x = torch.randn(1, fin, dtype=torch.float16, device=DEV)
    with torch.no_grad():
        [m(static_x) for m in mods.values()]  # warm-up (lazy kernel resolution) outside capture
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs = [holder.q_proj(static_x), holder.k_proj(static_x), holder.v_proj(static_x)]
        new_x = torch.randn(1, fin, dtype=torch.float16, device=DEV)
        static_x.copy_(new_x)
        graph.replay()
        torch.cuda.synchronize()
        aqlm.unfuse_shared_input_linears(holder)
        for n, y in zip(mods, outs):
            assert torch.equal(y, mods[n](new_x))


def test_prepack_model_runs_the_load_time_repack_eagerly(hk):
    from aqlm.checkpoint import memory_report, prepack_model

    mods = torch.nn.ModuleDict()
    Ls = {}
    for k, (n, fi, fo) in enumerate([("big", 2048, 1536), ("small", 512, 128)]):
        Ls[n] = orc.make_layer(990 + k, fi, fo, 1, 16, 8, batch=1, bias=True)
        mods[n], T = _module_from(Ls[n], 1, 16, 8, fi, fo, torch.float16)
    assert memory_report(mods)["prepacked_layers"] == 0
    rep = prepack_model(mods, min_codes=100_000, drop_canonical=False)      # big: 393 216 codes -> repacked now; small: 8192 codes -> not
    assert rep["quantized_linears"] == 2 and rep["prepacked_layers"] == 1
    assert mods["big"]._packed_codes is not None and mods["small"]._packed_codes is None
    assert 1.5 * 2 * 393216 < rep["prepacked"] < 2.9 * 2 * 393216           # 3-4 B per code + padding of a small layer
    import aqlm_amd.inference as inf
    assert inf.PREPACK_MIN_CODES == 500_000   # the override did not leak
    for n, m in mods.items():
        T = to_dev(Ls[n], torch.float16)
        y64 = orc.dequantize_gemm(Ls[n]["x"], Ls[n]["codes"], Ls[n]["codebooks"], Ls[n]["scales"], Ls[n]["bias"])
        check_close(m(T["x"]).float().cpu().numpy(), y64, torch.float16, f"prepack_model {n}")


# ------------------------------------------------------------------ randomized shapes (seeded): every route, odd sizes
@pytest.mark.parametrize("seed", range(int(os.environ.get("AQLM_FUZZ_SEEDS", "48"))))  # AQLM_FUZZ_SEEDS=600: the long one-off sweep
def test_randomized_layer_against_oracle(hk, seed):
    """Random scheme / shape / batch / dtype / bias per seed, module-level forward (so the gemv rule, the large-batch
    op, the prepacked route -- threshold lowered -- and the generic kernels all get hit) against the fp64 oracle."""
    import aqlm_amd.inference as inf

    rng = np.random.default_rng(1000 + seed)
    K, nbits, g = [(1, 16, 8), (1, 16, 16), (2, 8, 8), (1, 8, 8), (8, 8, 32), (4, 8, 16), (1, 12, 8), (2, 8, 4)][seed % 8]
    unit = g * int(rng.choice([1, 8, 8, 16]))            # in_features: sometimes not a multiple of 8 groups
    fin = unit * int(rng.integers(1, 24))
    fout = int(rng.choice([int(rng.integers(1, 64)), int(rng.integers(64, 700)), 1024]))
    if seed >= 24:                                       # second half: larger layers
        fin = g * 8 * int(rng.integers(8, 8192 // (g * 8) + 1))
        fout = int(rng.integers(700, 5000))
    rows = int(rng.choice([1, 1, 2, 5, 6, 7, 13, 40]))
    dt = "float16" if rng.random() < 0.7 else "bfloat16"
    bias = bool(rng.random() < 0.5)
    dtype = tdtype(dt)
    L = orc.make_layer(2000 + seed, fin, fout, K, nbits, g, batch=rows, bias=bias,
                       float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    old, inf.PREPACK_MIN_CODES = inf.PREPACK_MIN_CODES, 20_000
    try:
        m, T = _module_from(L, K, nbits, g, fin, fout, dtype)
        x_in = T["x"]
        if seed % 3 == 1:      # non-contiguous rows: a column slice of a wider tensor
            wide = torch.zeros(rows, fin + 24, dtype=dtype, device=DEV)
            wide[:, 8:8 + fin] = T["x"]
            x_in = wide[:, 8:8 + fin]
        with torch.no_grad():
            y = m(x_in)
            y_again = m(T["x"])
            if seed % 3 == 2 and rows % 2 == 0:   # leading dimensions are flattened and restored
                y3 = m(T["x"].reshape(2, rows // 2, fin))
                assert y3.shape == (2, rows // 2, fout) and torch.equal(y3.reshape(rows, fout), y)
    finally:
        inf.PREPACK_MIN_CODES = old
    assert y.shape == (rows, fout) and torch.equal(y, y_again)
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    # routes that materialise W in the storage dtype (large batch, several codebooks) round the K-term sum of every
    # weight once -- exactly what the reference's dequant + GEMM path does; with 10^5 outputs the 4.5-sigma tail of that
    # noise needs a slightly wider per-element bound (the mean bound is unchanged)
    wide = 1.0
    fused_kx8 = (K, nbits, g) == (2, 8, 8) and fin % 128 == 0 and fin >= 384   # the fused MFMA kernel: exact products, no rounded W
    if rows > 6 and K > 1 and not fused_kx8:
        wide = 1.6 if dtype == torch.float16 else 3.0   # bf16: the GEMM output itself is rounded to 8 bits before bias
    check_close(y.float().cpu().numpy(), y64, dtype, f"seed {seed}: {K}x{nbits}g{g} {fin}->{fout} rows={rows} {dt} bias={bias}",
                el_scale=wide)
    if seed % 2 == 0:   # backward through the module (every scheme has a route): grad_input = grad_output @ W
        xg = T["x"].clone().requires_grad_(True)
        go = torch.randn(rows, fout, generator=torch.Generator().manual_seed(seed)).to(dtype).to(DEV)
        m(xg).backward(go)
        W64 = orc.dequantize_weight(L["codes_unsigned"], L["codebooks"], L["scales"])
        g64 = go.double().cpu().numpy() @ W64
        got = xg.grad.float().cpu().numpy()
        rel = np.mean(np.abs(got - g64)) / np.mean(np.abs(g64))
        assert rel <= (2e-3 if dtype == torch.float16 else 1.2e-2), f"seed {seed}: grad_input mean-rel {rel:.3e}"


@pytest.mark.parametrize("seed", range(int(os.environ.get("AQLM_FUZZ_GROUP_SEEDS", "10"))))
def test_randomized_shared_input_groups(hk, seed):
    """Random member count / scheme / shapes / rows for shared-input groups; 1x16 outputs must equal the unfused
    modules bit for bit, K x 8 outputs within the parity bound; both against the oracle."""
    import aqlm
    import aqlm_amd.inference as inf

    rng = np.random.default_rng(3000 + seed)
    K, nbits, g = [(1, 16, 8), (2, 8, 8), (1, 16, 16), (1, 8, 8)][seed % 4]
    n = int(rng.integers(2, 5))
    fin = g * 8 * int(rng.integers(2, 40))
    fouts = [int(rng.choice([int(rng.integers(8, 200)), int(rng.integers(200, 3000))])) for _ in range(n)]
    rows = int(rng.choice([1, 1, 2, 6]))
    dt = "float16" if seed % 3 else "bfloat16"
    dtype = tdtype(dt)
    fd = np.float16 if dtype == torch.float16 else "bfloat16"
    old, inf.PREPACK_MIN_CODES = inf.PREPACK_MIN_CODES, (30_000 if seed % 2 else 0)
    try:
        holder = torch.nn.Module()
        Ls, x = [], None
        names = ["q_proj", "k_proj", "v_proj", "gate_proj", "up_proj"]
        pats = [tuple(names[:n])]
        for k in range(n):
            L = orc.make_layer(3100 + 10 * seed + k, fin, fouts[k], K, nbits, g, batch=rows, bias=bool(k % 2), float_dtype=fd)
            m, T = _module_from(L, K, nbits, g, fin, fouts[k], dtype)
            setattr(holder, names[k], m)
            Ls.append(L)
            x = T["x"] if x is None else x
        with torch.no_grad():
            ref = [getattr(holder, names[k])(x) for k in range(n)]
            groups = aqlm.fuse_shared_input_linears(holder, patterns=pats)
            assert len(groups) == 1 and len(groups[0].members) == n
            got = [getattr(holder, names[k])(x) for k in range(n)]
            assert (groups[0].launches, groups[0].served) == (1, n - 1)
            aqlm.unfuse_shared_input_linears(holder)
    finally:
        inf.PREPACK_MIN_CODES = old
    for k in range(n):
        y64 = orc.dequantize_gemm(Ls[0]["x"], Ls[k]["codes"], Ls[k]["codebooks"], Ls[k]["scales"], Ls[k]["bias"])
        check_close(got[k].float().cpu().numpy(), y64, dtype, f"seed {seed} member {k}")
        if nbits == 16:
            assert torch.equal(got[k], ref[k]), f"seed {seed} member {k}: fused 1x16 output differs from the unfused module"


# ------------------------------------------------------------------ sharded layer on the HIP ops
def test_sharded_layer_world1_nccl_and_emulated_split(hk):
    """ShardedQuantizedLinear with the real per-shard kernels: (a) a world_size-1 RCCL process group (the code path the
    8-GPU run takes, minus the wire), (b) an 8-way split emulated on one GPU: in-split partial outputs must add up to
    the unsharded result, out-split slices must concatenate to it bit for bit."""
    import socket

    import torch.distributed as dist

    from aqlm_amd.sharded import ShardedQuantizedLinear, shard_bounds

    fin, fout, g = 8192, 1536, 8
    L = orc.make_layer(515, fin, fout, 1, 16, g, batch=2, bias=True)
    T = to_dev(L, torch.float16)
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    full = hk.code1x16_matmat(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    # (a) world_size 1 over RCCL
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device(DEV))
    try:
        for mode in ("in", "out"):
            m = ShardedQuantizedLinear.from_full(T["codes"], T["codebooks"], T["scales"].reshape(-1, 1, 1, 1), T["bias"], mode=mode)
            y = m(T["x"])
            # 1.5 M codes: the shard takes the prepacked kernel, like a QuantizedLinear of that size would
            assert m._packed is not None
            assert torch.equal(y, hk.code1x16_matmat_packed(T["x"], m._packed, T["codebooks"], T["scales"], T["bias"])), mode
            check_close(y.float().cpu().numpy(), y64, torch.float16, f"sharded world 1 {mode}")
        t = torch.ones(4, device=DEV)
        dist.all_reduce(t)          # the collective itself works on this box
        assert float(t.sum()) == 4.0
    finally:
        if created:
            dist.destroy_process_group()
    # (b) 8-way split emulated on one device
    world = 8
    acc = torch.zeros(2, fout, dtype=torch.float32, device=DEV)
    acc16 = torch.zeros(2, fout, dtype=torch.float16, device=DEV)
    for r in range(world):
        j0, j1 = shard_bounds(fin // g, world, r, multiple=8)
        part = hk.code1x16_matmat(T["x"][:, j0 * g:j1 * g], T["codes"][:, j0:j1].contiguous(), T["codebooks"], T["scales"],
                                  T["bias"] if r == 0 else None)
        acc += part.float()
        acc16 += part           # what an fp16 all-reduce does (sequential here; a ring / tree rounds as often)
    check_close(acc.half().float().cpu().numpy(), y64, torch.float16, "in-split x8, fp32 reduction (the default)")
    # reducing the 8 partials in fp16 stays inside the 1e-3 bar as well, but spends most of it: the default is fp32
    e32 = float(np.abs(acc.half().float().cpu().numpy() - y64).mean() / np.abs(y64).mean())
    e16 = float(np.abs(acc16.float().cpu().numpy() - y64).mean() / np.abs(y64).mean())
    assert e32 < e16 < 1e-3, (e32, e16)
    outs = []
    for r in range(world):
        i0, i1 = shard_bounds(fout, world, r)
        outs.append(hk.code1x16_matmat(T["x"], T["codes"][i0:i1].contiguous(), T["codebooks"], T["scales"][i0:i1].contiguous(),
                                       T["bias"][i0:i1].contiguous()))
    assert torch.equal(torch.cat(outs, dim=-1), full)


def test_derived_state_follows_the_parameters(hk):
    """The prepacked buffer must track `codes`: in-place reloads, .to()/.half() conversions and a first forward inside
    a hipGraph capture (where the repack is postponed) all have to give the right answer."""
    import aqlm_amd.inference as inf

    fin, fout = 2048, 640
    La = orc.make_layer(7001, fin, fout, 1, 16, 8, batch=1, bias=True)
    Lb = orc.make_layer(7002, fin, fout, 1, 16, 8, batch=1, bias=True)
    old, inf.PREPACK_MIN_CODES = inf.PREPACK_MIN_CODES, 100_000
    try:
        m, Ta = _module_from(La, 1, 16, 8, fin, fout, torch.float16)
        Tb = to_dev(Lb, torch.float16)
        ya = m(Ta["x"])
        assert m._packed_codes is not None
        check_close(ya.float().cpu().numpy(), orc.dequantize_gemm(La["x"], La["codes"], La["codebooks"], La["scales"], La["bias"]),
                    torch.float16, "before reload")
        # in-place reload of other weights (what load_state_dict does)
        m.load_state_dict({"codes": Tb["codes"], "codebooks": Tb["codebooks"], "scales": Tb["scales"].reshape(-1, 1, 1, 1),
                           "bias": Tb["bias"]})
        yb = m(Tb["x"])
        check_close(yb.float().cpu().numpy(), orc.dequantize_gemm(Lb["x"], Lb["codes"], Lb["codebooks"], Lb["scales"], Lb["bias"]),
                    torch.float16, "after load_state_dict")
        # rebinding the parameter (accelerate's set_module_tensor_to_device, `module.codes = ...`) must be noticed too: a
        # fresh Parameter starts at the same version counter as the old one
        m.codes = torch.nn.Parameter(Ta["codes"].clone(), requires_grad=False)
        m.codebooks = torch.nn.Parameter(Ta["codebooks"].clone(), requires_grad=False)
        m.scales = torch.nn.Parameter(Ta["scales"].reshape(-1, 1, 1, 1).clone(), requires_grad=False)
        m.bias = torch.nn.Parameter(Ta["bias"].clone(), requires_grad=False)
        assert torch.equal(m(Ta["x"]), ya)
        m.load_state_dict({"codes": Tb["codes"], "codebooks": Tb["codebooks"], "scales": Tb["scales"].reshape(-1, 1, 1, 1),
                           "bias": Tb["bias"]})
        assert torch.equal(m(Tb["x"]), yb)
        # dtype conversion drops and rebuilds the derived state
        m.to(torch.bfloat16)
        assert m._packed_codes is None and m.gemv_op is None
        yc = m(Tb["x"].to(torch.bfloat16))
        assert m._packed_codes is not None and yc.dtype == torch.bfloat16
        rel = (yc.float() - yb.float()).abs().mean() / yb.float().abs().mean()
        assert rel < 2e-2
        # first forward inside a capture: repack postponed, direct kernel captured, result correct on replay
        m2, _ = _module_from(La, 1, 16, 8, fin, fout, torch.float16)
        static_x = Ta["x"].clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(graph, stream=s):
                y_static = m2(static_x)
        assert m2._packed_codes is None and m2._prepack_deferred
        graph.replay()
        torch.cuda.synchronize()
        check_close(y_static.float().cpu().numpy(), ya.float().cpu().numpy().astype(np.float64), torch.float16, "captured first call")
        y_later = m2(Ta["x"])                      # outside the capture: now the repack happens
        assert m2._packed_codes is not None and torch.equal(y_later, ya)
    finally:
        inf.PREPACK_MIN_CODES = old


@pytest.mark.raw_prepack
def test_derived_state_notices_writes_behind_the_version_counter(hk, monkeypatch):
    """`m.codes.data.copy_(...)` / `m.codebooks.data.copy_(...)` change neither identity nor version of the parameter: the
    prepacked buffer, the codebook image and range, the dense W and the raw op's cache would go stale silently (VERDICT r04
    weak #1b; the reference reads the live tensors on every call, cuda_kernel.cpp:148-182).  Every DERIVED_CHECK_EVERY-th forward
    re-takes the parameters' checksums; `invalidate_derived_state()` does it at once."""
    import aqlm_amd.inference as inf

    fin, fout = 2048, 640
    La = orc.make_layer(7101, fin, fout, 1, 16, 8, batch=1, bias=True)
    Lb = orc.make_layer(7102, fin, fout, 1, 16, 8, batch=1, bias=True)
    monkeypatch.setattr(inf, "PREPACK_MIN_CODES", 100_000)
    monkeypatch.setattr(inf, "DERIVED_CHECK_EVERY", 8)
    m, Ta = _module_from(La, 1, 16, 8, fin, fout, torch.float16)
    Tb = to_dev(Lb, torch.float16)
    want_a = orc.dequantize_gemm(La["x"], La["codes"], La["codebooks"], La["scales"], La["bias"])
    want_b = orc.dequantize_gemm(La["x"], Lb["codes"], Lb["codebooks"], La["scales"], La["bias"])
    ya = m(Ta["x"])
    assert m._packed_codes is not None and m._derived_checks is not None and set(m._derived_checks) == {"codes", "codebooks"}
    check_close(ya.float().cpu().numpy(), want_a, torch.float16, "before the write")
    v0 = m.codes._version
    m.codes.data.copy_(Tb["codes"])            # no version bump, same object, same storage
    m.codebooks.data.copy_(Tb["codebooks"])
    assert m.codes._version == v0
    stale = 0
    for _ in range(10):                        # at most DERIVED_CHECK_EVERY calls on the old weights, then the new ones
        y = m(Ta["x"])
        if torch.equal(y, ya):
            stale += 1
    assert stale < 8
    check_close(m(Ta["x"]).float().cpu().numpy(), want_b, torch.float16, "after the periodic check")
    # on demand
    m.codes.data.copy_(Ta["codes"])
    m.codebooks.data.copy_(Ta["codebooks"])
    m.invalidate_derived_state()
    assert torch.equal(m(Ta["x"]), ya)
    # only the codebook changes (PV-tuning style): range + (relabelled buffers) image follow
    m.codebooks.data.mul_(0.5)
    m.invalidate_derived_state()
    check_close((m(Ta["x"]).float().cpu().numpy() - Ta["bias"].float().cpu().numpy()[None, :]) * 2.0,
                want_a - La["bias"][None, :].astype(np.float64), torch.float16, "halved codebook", el_scale=4.0)
    m.codebooks.data.mul_(2.0)
    m.invalidate_derived_state()
    assert torch.equal(m(Ta["x"]), ya)
    # the dense-W escape hatch is derived state too
    m.prefer_dense_below_rows = 64
    x16 = torch.randn(16, fin, dtype=torch.float16, device=DEV)
    y16a = m(x16)
    assert m._dense is not None and "scales" in m._derived_checks
    m.codes.data.copy_(Tb["codes"])
    m.codebooks.data.copy_(Tb["codebooks"])
    for _ in range(9):
        y16b = m(x16)
    assert m._dense is not None and not torch.equal(y16a, y16b)
    W = orc.dequantize_gemm(np.eye(fin, dtype=np.float32)[:8], Lb["codes"], Lb["codebooks"], La["scales"], None)   # 8 columns of W^T
    got = m(torch.eye(fin, dtype=torch.float16, device=DEV)[:8].contiguous().repeat(2, 1))[:8].float().cpu().numpy() - La["bias"][None, :]
    check_close(got, W, torch.float16, "dense W after the write", el_scale=2.0)
    m.prefer_dense_below_rows = 0
    # the raw op's transparent cache: a hit is re-checked every RAW_OP_CHECK_EVERY hits (the compiled op hands that call to Python)
    monkeypatch.setattr(hk, "RAW_OP_PREPACK_MIN_CODES", 100_000)
    monkeypatch.setattr(hk, "RAW_OP_CHECK_EVERY", 8)
    hk.clear_raw_op_prepack_cache()
    codes, cb = Ta["codes"].clone(), Ta["codebooks"].clone()
    sc = Ta["scales"].reshape(-1, 1, 1, 1)
    op = torch.ops.aqlm.code1x16_matmat
    r0 = op(Ta["x"], codes, cb, sc, Ta["bias"])
    r1 = op(Ta["x"], codes, cb, sc, Ta["bias"])
    assert id(codes) in hk._RAW_PACKED and torch.equal(r0, r1)
    check_close(r0.float().cpu().numpy(), want_a, torch.float16, "raw op before the write")
    codes.data.copy_(Tb["codes"])
    cb.data.copy_(Tb["codebooks"])
    for _ in range(20):
        r2 = op(Ta["x"], codes, cb, sc, Ta["bias"])
    check_close(r2.float().cpu().numpy(), want_b, torch.float16, "raw op after the write")
    hk.clear_raw_op_prepack_cache()


# ------------------------------------------------------------------ fused finalize + one-shot all-reduce (xGMI path)
@pytest.mark.parametrize("fused_publish", [True, False])
def test_xgmi_fused_finalize_world1(hk, fused_publish):
    """ShardedQuantizedLinear(collective="xgmi") on a single rank: the published vector is read back through the same
    protocol (flag, system-scope loads) and must give exactly the ordinary prepacked result; repeated calls advance the
    epoch (both halves of the double buffer), also inside a captured hipGraph.  Both forms: the matvec publishes its totals
    itself (2 launches; same bits as the single-kernel finalize) and partials + publish + reduce (3 launches; same bits
    as the two-kernel finalize)."""
    import aqlm_amd.sharded as sharded
    from aqlm_amd.sharded import ShardedQuantizedLinear

    fin, fout = 2048, 1536
    L = orc.make_layer(616, fin, fout, 1, 16, 8, batch=4, bias=True)
    T = to_dev(L, torch.float16)
    m = ShardedQuantizedLinear.from_full(T["codes"], T["codebooks"], T["scales"].reshape(-1, 1, 1, 1), T["bias"], mode="in",
                                        collective="xgmi")
    import aqlm_amd.inference as inf

    old, inf.PREPACK_MIN_CODES = inf.PREPACK_MIN_CODES, 100_000
    old_fp, sharded.XGMI_FUSED_PUBLISH = sharded.XGMI_FUSED_PUBLISH, fused_publish
    try:
        ys = [m(T["x"][:b]) for b in (1, 4, 2, 1, 3)]
        assert m._xgmi is not None and m._xgmi_ok and not m._xgmi.timed_out()
        hk.set_fused_finalize(fused_publish)   # the reference with the same arithmetic
        try:
            for y, b in zip(ys, (1, 4, 2, 1, 3)):
                ref = hk.code1x16_matmat_packed(T["x"][:b], m._packed, T["codebooks"], T["scales"], T["bias"])
                assert torch.equal(y, ref), b
        finally:
            hk.set_fused_finalize(True)
        y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
        check_close(ys[1].float().cpu().numpy(), y64, torch.float16, "xgmi world 1")
        # graph capture: the epoch lives in device memory, so replays keep working
        static_x = T["x"][:1].clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(graph, stream=s):
                y_static = m(static_x)
        hk.set_fused_finalize(fused_publish)
        try:
            for k in range(5):
                static_x.copy_(T["x"][k % 4:k % 4 + 1])
                graph.replay()
                torch.cuda.synchronize()
                assert torch.equal(y_static, hk.code1x16_matmat_packed(T["x"][k % 4:k % 4 + 1], m._packed, T["codebooks"], T["scales"], T["bias"]))
        finally:
            hk.set_fused_finalize(True)
        assert not m._xgmi.timed_out()
    finally:
        inf.PREPACK_MIN_CODES = old
        sharded.XGMI_FUSED_PUBLISH = old_fp


def _xgmi_two_rank_worker(rank, world, port, q):
    import os

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only; both ranks share cuda:0
    try:
        import aqlm_amd.inference as inf
        from aqlm_amd.sharded import ShardedQuantizedLinear

        inf.PREPACK_MIN_CODES = 100_000
        fin, fout = 4096, 2048
        L = orc.make_layer(717, fin, fout, 1, 16, 8, batch=4, bias=True)
        T = to_dev(L, torch.float16)
        m = ShardedQuantizedLinear.from_full(T["codes"], T["codebooks"], T["scales"].reshape(-1, 1, 1, 1), T["bias"], mode="in",
                                            collective="xgmi")
        outs = []
        for it in range(24):
            b = 1 + it % 4
            if (it + rank) % 3 == 0:
                torch.cuda._sleep(200_000 * (1 + rank))   # uneven pace between the ranks
            outs.append(m(T["x"][:b]).clone())
        torch.cuda.synchronize()
        y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
        errs = [float(np.abs(o.float().cpu().numpy() - y64[:o.shape[0]]).mean() / np.abs(y64[:o.shape[0]]).mean()) for o in outs]
        digest = float(torch.cat([o.double().reshape(-1) for o in outs]).sum())
        q.put((rank, bool(m._xgmi_ok), m._xgmi.timed_out(), max(errs), digest))
    finally:
        dist.destroy_process_group()


def test_xgmi_fused_finalize_two_ranks_on_one_gpu(hk):
    """Two processes share the one GPU of this box (gloo as control plane): each maps the other's state through IPC,
    runs its half of an in-split layer and the fused finalize reads both halves' sums through the mapped memory.  Checks
    the real hand-shake (IPC handles, system-scope stores / loads, flags, double buffering under uneven pace): results
    within tolerance of the unsharded oracle and bit-identical on both ranks.  (The wire is not xGMI here -- same-device
    memory -- so this validates the protocol, not the link.)"""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_xgmi_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    for rank, ok, timed_out, err, _ in res:
        assert ok and not timed_out and err < 1e-3, (rank, ok, timed_out, err)
    assert res[0][4] == res[1][4]   # bit-identical replicas of y


# ------------------------------------------------------------------ round 3: compiled fast lane, re-entrancy, inference mode
def _big_packed_module(seed=31, fin=4096, fout=1536, dt=torch.float16, batch=4):
    L = orc.make_layer(seed, fin, fout, 1, 16, 8, batch=batch, bias=True,
                       float_dtype=np.float16 if dt == torch.float16 else "bfloat16")
    m, T = _module_from(L, 1, 16, 8, fin, fout, dt)
    return L, m, T


@pytest.mark.parametrize("K,nbits,g,fin,fout", [(1, 16, 8, 4096, 1536), (1, 16, 8, 1024, 96), (1, 16, 16, 1024, 96),
                                                (2, 8, 8, 1024, 200), (1, 8, 8, 512, 64)])
def test_fast_lane_equals_python_path(hk, K, nbits, g, fin, fout):
    """The compiled fast lane of QuantizedLinear (aqlm_amd/csrc_front/front.cpp) launches the same kernels as the Python
    ops: bit-identical outputs; calls it does not serve (rows, dtype, grad) fall through to the Python path; a parameter
    that is rebound or written in place is noticed."""
    from aqlm_amd import _front

    if not _front.available():
        pytest.skip("aqlm_amd/_aqlm_front.so not built")
    L = orc.make_layer(123, fin, fout, K, nbits, g, batch=6, bias=True)
    m, T = _module_from(L, K, nbits, g, fin, fout, torch.float16)
    with torch.no_grad():
        y = m(T["x"])                       # first call: prepare_matmul_op, builds the lane
        assert m._fast is not None, "no fast lane was built for a tuned scheme"
        y_fast = m(T["x"])
        lane, m._fast = m._fast, None
        y_py = m(T["x"])                    # the Python path alone
        m._fast = lane
        assert torch.equal(y, y_fast) and torch.equal(y_fast, y_py)
        y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
        check_close(y_fast.float().cpu().numpy(), y64, torch.float16, f"fast lane {K}x{nbits}g{g}")
        # shapes are kept; 3-d inputs; one row
        assert m(T["x"].reshape(2, 3, fin)).shape == (2, 3, fout)
        if nbits == 8:   # 3+ rows of 1x8 / 2x8 run on the fused MFMA kernel, single rows on the matvec kernels: close, not equal
            check_close(m(T["x"][1:2])[0].float().cpu().numpy(), y[1].float().cpu().numpy().astype(np.float64), torch.float16, "one row vs batched")
        else:
            assert torch.equal(m(T["x"][1:2])[0], y[1])
        # not the lane's calls
        assert lane(torch.cat([T["x"], T["x"][:1]])) is None                     # 7 rows: the gemm rule of the module
        assert lane(T["x"].float()) is None                                       # another dtype
    xg = T["x"].clone().requires_grad_(True)
    assert lane(xg) is None                                                       # needs grad: autograd op of the Python path
    m(xg).sum().backward()
    assert xg.grad is not None
    # an in-place update of the codebooks and a rebound codes parameter are noticed; results follow the new values
    with torch.no_grad():
        m.codebooks.mul_(2.0)
        y2 = m(T["x"])
        check_close(y2.float().cpu().numpy(), orc.dequantize_gemm(L["x"], L["codes"], 2.0 * L["codebooks"].astype(np.float32), L["scales"], L["bias"]),
                    torch.float16, "after a codebook update")
        assert m._fast is not None and m._fast.is_current()
        L2 = orc.make_layer(124, fin, fout, K, nbits, g, batch=6, bias=True)
        m.codes = torch.nn.Parameter(torch.from_numpy(L2["codes"]).to(DEV), requires_grad=False)
        y3 = m(T["x"])
        check_close(y3.float().cpu().numpy(), orc.dequantize_gemm(L["x"], L2["codes"], 2.0 * L["codebooks"].astype(np.float32), L["scales"], L["bias"]),
                    torch.float16, "after rebinding codes")
        assert m._fast is not None and m._fast.is_current()


@pytest.mark.raw_prepack
def test_compiled_raw_ops_equal_the_python_implementations(hk):
    """torch.ops.aqlm.code1x16_matmat / code2x8_matmat / code1x8_matmat are compiled dispatcher kernels when the front end is
    built (front.cpp; the reference's ops are C++ too, cuda_kernel.cpp:148-182): decode calls are launched from C++ on the same
    kernels as the Python functions (bit-identical), large 1x16 layers through the Python-owned prepack cache, and every call
    outside the lane -- more rows, malformed arguments with their error types, a layer edited in place -- behaves as before."""
    import gc

    from aqlm_amd import _front

    if not (_front.available() and hk._RAW_FAST is not None):
        pytest.skip("compiled raw ops not installed (aqlm_amd/_aqlm_front.so not built)")
    ext = _front.ext
    ops = {(1, 16): (torch.ops.aqlm.code1x16_matmat, hk.code1x16_matmat), (2, 8): (torch.ops.aqlm.code2x8_matmat, hk.code2x8_matmat),
           (1, 8): (torch.ops.aqlm.code1x8_matmat, hk.code1x8_matmat)}
    # small layers (direct kernels), 1 / 3 / 6 rows, with and without bias, both dtypes, a strided input
    for (K, nbits), fin, fout, rows, bias, dt in (((1, 16), 512, 300, 1, True, "float16"), ((1, 16), 1024, 77, 6, False, "bfloat16"),
                                                  ((2, 8), 1024, 200, 3, True, "float16"), ((2, 8), 4096, 333, 1, False, "bfloat16"),
                                                  ((1, 8), 512, 64, 6, True, "float16"), ((1, 16), 2048, 128, 3, True, "float16")):
        dtype = tdtype(dt)
        L = orc.make_layer(777 + fin + fout, fin, fout, K, nbits, 8, batch=rows, bias=bias, float_dtype=np.float16 if dt == "float16" else "bfloat16")
        T = to_dev(L, dtype)
        op, pyfn = ops[(K, nbits)]
        n0 = ext.raw_served()
        y = op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
        assert ext.raw_served() == n0 + 1, "the call was not launched by the compiled op"
        assert torch.equal(y, pyfn(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"]))
        check_close(y.float().cpu().numpy(), orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"]), dtype,
                    f"compiled raw op {K}x{nbits} {fin}->{fout} rows={rows}")
        wide = torch.zeros(rows, fin + 16, dtype=dtype, device=DEV)
        wide[:, 8:8 + fin] = T["x"]
        assert torch.equal(op(wide[:, 8:8 + fin], T["codes"], T["codebooks"], T["scales"], T["bias"]), y)
        assert op(T["x"].reshape(1, rows, fin), T["codes"], T["codebooks"], T["scales"], T["bias"]).shape == (1, rows, fout)
    # a large 1x16 layer: the first call packs (Python), registers; from then on C++ launches the prepacked kernel
    fin, fout = 4096, 2048
    L = orc.make_layer(4242, fin, fout, 1, 16, 8, batch=2, bias=True)
    T = to_dev(L, torch.float16)
    op, pyfn = ops[(1, 16)]
    p0, e0 = hk._RAW_STATS["packs"], ext.raw_entries()
    y0 = op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    assert hk._RAW_STATS["packs"] == p0 + 1 and ext.raw_entries() == e0 + 1
    n0 = ext.raw_served()
    y1 = op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    y1b = op(T["x"][:1], T["codes"], T["codebooks"], T["scales"], T["bias"])
    assert ext.raw_served() == n0 + 2 and torch.equal(y0, y1) and torch.equal(y1b[0], y1[0])
    assert torch.equal(y1, hk.code1x16_matmat_packed(T["x"], hk._RAW_PACKED[id(T["codes"])][2], T["codebooks"], T["scales"], T["bias"]))
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    check_close(y1.float().cpu().numpy(), y64, torch.float16, "compiled raw op on the prepacked kernel")
    # inside a hipGraph (no packing, no allocation of cells: the registered layer is launched as it is)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        yg = op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(yg, y1)
    # not the lane's calls: 9 rows (MFMA kernel through the Python op), wrong dtypes / shapes with the reference's error types
    n0 = ext.raw_served()
    x9 = T["x"][:1].expand(9, fin).contiguous()
    check_close(op(x9, T["codes"], T["codebooks"], T["scales"], T["bias"])[8].float().cpu().numpy(), y64[0], torch.float16, "9 rows")
    with pytest.raises(NotImplementedError, match="only support float16 and bfloat16"):
        op(T["x"].float(), T["codes"], T["codebooks"], T["scales"], T["bias"])
    with pytest.raises(NotImplementedError):
        op(T["x"].bfloat16(), T["codes"], T["codebooks"], T["scales"], T["bias"])
    with pytest.raises(ValueError):
        op(T["x"][:, :fin - 8], T["codes"], T["codebooks"], T["scales"], T["bias"])
    with pytest.raises(NotImplementedError):
        torch.ops.aqlm.code2x8_matmat(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    assert ext.raw_served() == n0
    # the codebooks change in place: the registered range is stale -> Python refreshes it and registers again; values follow
    T["codebooks"].mul_(2.0)
    y2 = op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    check_close(y2.float().cpu().numpy(), orc.dequantize_gemm(L["x"], L["codes"], 2.0 * L["codebooks"].astype(np.float32), L["scales"], L["bias"]),
                torch.float16, "compiled raw op after a codebook update")
    n0 = ext.raw_served()
    assert torch.equal(op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"]), y2) and ext.raw_served() == n0 + 1
    # the codes change in place: packed again, registered again
    L2 = orc.make_layer(4243, fin, fout, 1, 16, 8, batch=2, bias=True)
    T["codes"].copy_(torch.from_numpy(L2["codes"]).to(DEV))
    y3 = op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    check_close(y3.float().cpu().numpy(), orc.dequantize_gemm(L["x"], L2["codes"], 2.0 * L["codebooks"].astype(np.float32), L["scales"], L["bias"]),
                torch.float16, "compiled raw op after an in-place edit of the codes")
    assert hk._RAW_STATS["packs"] == p0 + 2
    # many layers, each called twice (the reference benchmark's pattern): the second calls are hits of the COMPILED op, and the
    # cache must count them -- it once read "packs without a single hit" and switched itself off after 8 layers
    hk.clear_raw_op_prepack_cache()
    many = [to_dev(orc.make_layer(6000 + k, 4096, 1024, 1, 16, 8, batch=1, bias=False), torch.float16) for k in range(12)]
    pk0 = hk._RAW_STATS["packs"]
    for Tm in many:
        for _ in range(2):
            op(Tm["x"], Tm["codes"], Tm["codebooks"], Tm["scales"], None)
    assert hk._RAW_STATS["packs"] == pk0 + 12 and ext.raw_entries() >= 12
    n0 = ext.raw_served()
    for Tm in many:
        op(Tm["x"], Tm["codes"], Tm["codebooks"], Tm["scales"], None)
    assert ext.raw_served() == n0 + 12 and hk._RAW_STATS["packs"] == pk0 + 12
    del many, Tm
    gc.collect()
    assert ext.raw_entries() == 0
    # a dead codes tensor leaves nothing behind on either side; the knobs reach the compiled side through the cache's clear
    op(T["x"], T["codes"], T["codebooks"], T["scales"], T["bias"])
    e1 = ext.raw_entries()
    assert e1 == 1
    del T["codes"]
    gc.collect()
    assert ext.raw_entries() == e1 - 1
    hk.clear_raw_op_prepack_cache()
    assert ext.raw_entries() == 0
    hk.set_fused_finalize(False)   # two-kernel finalize: the compiled ops stand down
    try:
        T5 = to_dev(orc.make_layer(9, 512, 64, 1, 16, 8, batch=1, bias=False), torch.float16)
        n0 = ext.raw_served()
        op(T5["x"], T5["codes"], T5["codebooks"], T5["scales"], None)
        assert ext.raw_served() == n0
    finally:
        hk.set_fused_finalize(True)


def test_packed_layer_runs_on_two_streams_at_once(hk):
    """Re-entrancy of the single-kernel finalize (reference launcher: stateless on the caller's stream,
    cuda_kernel.cu:505-509): one prepacked module -- i.e. one packed buffer -- driven from two streams concurrently, through
    the module (fast lane), the Python op and a shared-input launch.  Every result equals the single-stream result bit for
    bit (the accumulator cells are per stream, the packed buffer is only read)."""
    L, m, T = _big_packed_module(batch=6)
    xs = [torch.roll(T["x"], k, dims=0) * (1.0 + 0.25 * k) for k in range(8)]
    with torch.no_grad():
        ref = [m(x) for x in xs]
        assert m._packed_codes is not None
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        out = {}
        for rep in range(30):               # interleaved issue: both streams have many launches of the same layer in flight
            for k, x in enumerate(xs):
                s = s1 if (k + rep) % 2 == 0 else s2
                with torch.cuda.stream(s):
                    out[(rep, k)] = m(x)
                    out[(rep, k, "op")] = hk.code1x16_matmat_packed(x, m._packed_codes, m.codebooks, m.scales, m.bias)
                    out[(rep, k, "multi")] = hk.code1x16_matmat_packed_multi(x, [m._packed_codes], [m.codebooks], [m.scales], [m.bias])[0]
        torch.cuda.synchronize()
        for key, y in out.items():
            assert torch.equal(y, ref[key[1]]), f"stream-concurrent launch {key} differs from the single-stream result"
        # the cells inside the packed buffer were not used and the per-stream cells are back to zero
        for st in (s1, s2, torch.cuda.current_stream()):
            cells = hk.accumulator_cells(stream=st.cuda_stream)
            assert cells and all(int(c.abs().max()) == 0 for c in cells)


def test_packed_layer_under_inference_mode(hk):
    """Parameters created under ``torch.inference_mode()`` carry no version counter: the prepacked path, the
    raw op's prepack cache and the fast lane must not ask for one."""
    with torch.inference_mode():
        L, m, T = _big_packed_module(seed=77, fout=1024)
        y = m(T["x"])
        assert m._packed_codes is not None
        y2 = m(T["x"])
        y_op = torch.ops.aqlm.code1x16_matmat(T["x"], m.codes, m.codebooks, m.scales, m.bias)
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    check_close(y.float().cpu().numpy(), y64, torch.float16, "inference-mode module")
    assert torch.equal(y, y2)
    check_close(y_op.float().cpu().numpy(), y64, torch.float16, "inference-mode raw op")


def test_sharded_layer_repacks_when_its_codes_change(hk):
    """ShardedQuantizedLinear kept multiplying with a stale prepacked buffer after its codes were replaced."""
    from aqlm_amd.sharded import ShardedQuantizedLinear

    fin, fout = 4096, 1024
    La = orc.make_layer(5, fin, fout, 1, 16, 8, batch=2, bias=True)
    Lb = orc.make_layer(6, fin, fout, 1, 16, 8, batch=2, bias=True)
    Ta, Tb = to_dev(La, torch.float16), to_dev(Lb, torch.float16)
    sh = ShardedQuantizedLinear.from_full(Ta["codes"].clone(), Ta["codebooks"], Ta["scales"], Ta["bias"], mode="in")
    with torch.no_grad():
        ya = sh(Ta["x"])
        assert sh._packed is not None
        sh.codes.copy_(Tb["codes"])                       # load_state_dict / in-place write
        yb = sh(Ta["x"])
        sh.codes = torch.nn.Parameter(Ta["codes"].clone(), requires_grad=False)   # rebind
        ya2 = sh(Ta["x"])
    ref_a = orc.dequantize_gemm(La["x"], La["codes"], La["codebooks"], La["scales"], La["bias"])
    ref_b = orc.dequantize_gemm(La["x"], Lb["codes"], La["codebooks"], La["scales"], La["bias"])
    check_close(ya.float().cpu().numpy(), ref_a, torch.float16, "sharded, first codes")
    check_close(yb.float().cpu().numpy(), ref_b, torch.float16, "sharded, after an in-place write of the codes")
    assert torch.equal(ya, ya2)


@pytest.mark.parametrize("outs,fin,dt,waves", [
    ((4096, 1024, 1024), 4096, "float16", 0),      # Llama-3-8B q / k / v: different wave and step counts per segment
    ((14336, 14336), 4096, "float16", 0),          # gate / up: 14 waves each + 2 DMA waves, 896-row groups (LDS 163.6 of 163.8 KB)
    ((14336, 14336), 4096, "float16", 16),         # the same packed for 16 waves: no DMA waves, every wave requests its share itself
    ((12288, 12288), 8192, "float16", 0),          # 13 steps per wave at 16 waves: packed for 16 by the default rule (self-service mode)
    ((1024, 1024, 1024, 1024), 4096, "bfloat16", 0),
    ((4096, 4096, 4096), 4096, "float16", 0),
    ((1536, 2048), 8192, "float16", 0),            # 8192-wide input: x fills the window to 16400 of 16640 bytes
    ((11008, 4096, 1024), 4096, "bfloat16", 15),   # mixed sizes, 15 waves: self-service mode with waves that have no rows in a segment
])
def test_pipelined_shared_input_launch(hk, outs, fin, dt, waves):
    """The pipelined shared-input kernel (one workgroup per CU walks the segments, codebook slices double-buffered, two DMA
    waves; gemv_1x16_packed_pipe_kernel): bit-identical to the per-segment kernel of the plain multi launch and to separate
    single-layer launches, repeatable (cells zero at rest), within tolerance of the oracle."""
    from aqlm_amd import _native

    dtype = tdtype(dt)
    Ls = [orc.make_layer(900 + i, fin, o, 1, 16, 8, batch=1, bias=(i % 2 == 0),
                         float_dtype=np.float16 if dt == "float16" else "bfloat16") for i, o in enumerate(outs)]
    Ts = [to_dev(L, dtype) for L in Ls]
    x = Ts[0]["x"]
    _native.set_tuning("packed_waves", waves)
    try:
        packed = [hk.prepack_1x16(T["codes"], 8, codebooks=T["codebooks"]) for T in Ts]
    finally:
        _native.set_tuning("packed_waves", 0)
    assert all(p is not None for p in packed)
    if waves:
        assert all(p.desc.waves == waves for p in packed)
    elif outs == (12288, 12288):
        assert all(p.desc.waves == 16 for p in packed)
    elif outs == (14336, 14336):
        assert all(p.desc.waves == 14 for p in packed)
    args = (x, packed, [T["codebooks"] for T in Ts], [T["scales"] for T in Ts], [T["bias"] for T in Ts])
    singles = [hk.code1x16_matmat_packed(x, p, T["codebooks"], T["scales"], T["bias"]) for p, T in zip(packed, Ts)]
    _native.set_tuning("packed_pipe", 0)
    try:
        plain = hk.code1x16_matmat_packed_multi(*args)
    finally:
        _native.set_tuning("packed_pipe", 1)
    for rep in range(5):
        piped = hk.code1x16_matmat_packed_multi(*args)
        for k in range(len(outs)):
            assert torch.equal(piped[k], plain[k]) and torch.equal(piped[k], singles[k]), f"segment {k}, repeat {rep}"
    for k, L in enumerate(Ls):
        y64 = orc.dequantize_gemm(Ls[0]["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
        check_close(piped[k].float().cpu().numpy(), y64, dtype, f"pipelined segment {k}")
    # inside a hipGraph (the decode loop's form): replay equals eager
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        hk.code1x16_matmat_packed_multi(*args)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            captured = hk.code1x16_matmat_packed_multi(*args)
    g.replay(); g.replay()
    torch.cuda.synchronize()
    for k in range(len(outs)):
        assert torch.equal(captured[k], singles[k])


def test_gpu_modules_copy_and_pickle_after_a_forward(hk):
    """a module that has run on the GPU holds a pybind11 fast lane (and a fused group a compiled launch);
    deepcopy / pickle / torch.save must work at any time, the copy must not share derived state with the original, and a
    module whose canonical codes were dropped must hand them to the copy."""
    import copy
    import io
    import pickle

    import aqlm
    from aqlm.checkpoint import prepack_model

    fin = 2048
    holder = torch.nn.Module()
    Ls = {}
    for k, (n, fo) in enumerate([("q_proj", 1536), ("k_proj", 512), ("v_proj", 512)]):
        Ls[n] = orc.make_layer(4300 + k, fin, fo, 1, 16, 8, batch=2, bias=(k == 0))
        m, T = _module_from(Ls[n], 1, 16, 8, fin, fo, torch.float16)
        setattr(holder, n, m)
    x = to_dev(Ls["q_proj"], torch.float16)["x"][:1].contiguous()
    prepack_model(holder, min_codes=100_000, drop_canonical=False)
    aqlm.fuse_shared_input_linears(holder)
    with torch.no_grad():
        want = {n: getattr(holder, n)(x) for n in Ls}
        assert holder.q_proj._packed_codes is not None
        clones = [copy.deepcopy(holder), pickle.loads(pickle.dumps(holder))]
        buf = io.BytesIO()
        torch.save(holder, buf)
        buf.seek(0)
        clones.append(torch.load(buf, weights_only=False))
        for c in clones:
            assert c.q_proj._fast is None and c.q_proj._packed_codes is None and c.q_proj.gemv_op is None
            g = c.q_proj._shared_input_group
            assert g is not None and g.members[0] is c.q_proj and g.members[1] is c.k_proj and g._fast_group is None
            assert c.q_proj.codes.data_ptr() != holder.q_proj.codes.data_ptr()
            for n in Ls:
                assert torch.equal(getattr(c, n)(x), want[n]), n
        # dropped canonical codes travel with the copy
        prepack_model(holder, min_codes=100_000, drop_canonical=True)
        assert holder.q_proj._codes_dropped
        c = copy.deepcopy(holder)
        assert not c.q_proj._codes_dropped and tuple(c.q_proj.codes.shape) == (1536, fin // 8, 1)
        assert torch.equal(c.q_proj.codes, to_dev(Ls["q_proj"], torch.float16)["codes"])
        for n in Ls:
            assert torch.equal(getattr(c, n)(x), want[n]), n
            assert torch.equal(getattr(holder, n)(x), want[n]), n


def test_prefer_dense_below_rows_escape_hatch(hk):
    """Opt-in: 7 .. N - 1 rows run as a dense GEMM on a cached copy of W (the fused MFMA op is slower than dense below ~128 rows);
    1 .. 6 rows and >= N rows keep their kernels; the copy follows the parameters; off by default."""
    from aqlm.checkpoint import enable_dense_below_rows

    fin, fout = 1024, 512
    L = orc.make_layer(515, fin, fout, 1, 16, 8, batch=40, bias=True)
    m, T = _module_from(L, 1, 16, 8, fin, fout, torch.float16)
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    with torch.no_grad():
        assert m.prefer_dense_below_rows == 0
        base = m(T["x"][:20])
        assert m._dense is None
        holder = torch.nn.ModuleDict({"l": m})
        assert enable_dense_below_rows(holder, 33) == 1
        y20 = m(T["x"][:20])
        assert m._dense is not None and tuple(m._dense[1].shape) == (fout, fin)
        check_close(y20.float().cpu().numpy(), y64[:20], torch.float16, "dense hatch, 20 rows")
        check_close(y20.float().cpu().numpy(), base.double().cpu().numpy(), torch.float16, "dense hatch vs fused op")
        assert torch.equal(m(T["x"][:3]), hk.code1x16_matmat(T["x"][:3], T["codes"], T["codebooks"], T["scales"], T["bias"]))  # matvec path
        w_before = m._dense[1]
        m(T["x"][:40])                                   # 40 rows >= 33: the fused op, no new copy
        assert m._dense[1] is w_before
        m.scales.mul_(2.0)                               # parameters change: the copy is rebuilt
        y20b = m(T["x"][:20])
        check_close(y20b.float().cpu().numpy(), 2.0 * (y64[:20] - L["bias"].astype(np.float64)) + L["bias"].astype(np.float64), torch.float16, "dense hatch after a scale update")
        enable_dense_below_rows(holder, 0)
        assert m._dense is None and m.prefer_dense_below_rows == 0
