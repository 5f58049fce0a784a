"""GPU parity of the slice-scan MFMA kernel (aqlm_hip_gemm_1x16_scan, round 6; an opt-in route: measured slower than the L2-gather
kernels at Llama layer sizes, see the kernel's header): the 1x16 g8 scheme at 2+ rows with the codebook slices in LDS -- against the fp64 oracle (tolerance AND the correctly-rounded statement), the direct matvec kernel (another order
of the same exact products), bit-exact repeatability, batch invariance, strided inputs, NaN rows, every K plan (1 .. 3 units per
wave, 1 .. 4 K chunks, ragged chunks), ragged row counts, and the ops / module that route to it.
Replaces cuda_kernel.cpp:165-175 (per-row relaunch) and cuda_kernel.cpp:249-301 (dequantise + cuBLAS)."""
import numpy as np
import pytest

from oracle import aqlm_oracle as orc

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from tests.test_hip_parity import DEV, check_close, check_rounded, tdtype, to_dev  # noqa: E402


@pytest.fixture(scope="module")
def hk():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    from aqlm_amd.inference_kernels import hip_kernel

    return hip_kernel


# (in, out, rows): one chunk x 2 units (4096), 1 unit and idle waves (1024, 256), 3 units per wave / 2 chunks (11008 at <= 16 rows),
# 3 ragged chunks (11008 at 32 rows), 4 chunks (14336), 2 chunks (8192); out not a multiple of 16 / of 4; rows across the pass sizes
CASES = [(4096, 4096, 8), (4096, 1024, 2), (1024, 512, 5), (256, 48, 3), (4096, 11008, 16), (11008, 4096, 7), (11008, 4096, 32),
         (14336, 4096, 4), (8192, 1000, 12), (4096, 4090, 9), (4096, 4096, 40), (2048, 2051, 130), (5120, 640, 17)]


@pytest.mark.parametrize("fin,fout,rows", CASES)
@pytest.mark.parametrize("dt", ["float16", "bfloat16"])
def test_scan_kernel_vs_oracle(hk, fin, fout, rows, dt):
    if dt == "bfloat16" and (fin * fout > 4096 * 4096 or rows > 40):
        pytest.skip("bf16: the small and mid cases cover the second instantiation")
    dtype = tdtype(dt)
    bias = (fin + fout + rows) % 2 == 0
    L = orc.make_layer(6600 + fin % 997 + fout % 991 + rows, fin, fout, 1, 16, 8, batch=rows, bias=bias,
                       float_dtype=np.float16 if dtype == torch.float16 else "bfloat16")
    T = to_dev(L, dtype)
    args = (T["codes"], T["codebooks"], T["scales"], T["bias"])
    y = hk.code1x16_matmat_scan(T["x"], *args)
    assert y is not None
    y64 = orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], L["bias"])
    what = f"scan 1x16g8 {fin}->{fout} rows {rows} {dt}"
    check_close(y.float().cpu().numpy(), y64, dtype, what)
    check_rounded(y.float().cpu().numpy(), y64, dtype, what)
    assert torch.equal(y, hk.code1x16_matmat_scan(T["x"], *args)), "not repeatable bit for bit"
    # batch invariance: a row's bits depend neither on the other rows nor on their number (passes of 16 / 32 rows included)
    for B in (1, 2, 3, 6, 15, 16, 17, 31, 33):
        if B < rows:
            assert torch.equal(hk.code1x16_matmat_scan(T["x"][:B], *args), y[:B]), f"{B}-row call differs"
    x2 = T["x"].clone()
    x2[1:] = torch.flip(x2[1:], dims=(0,))
    y2 = hk.code1x16_matmat_scan(x2, *args)
    assert torch.equal(y2[0], y[0]) and torch.equal(y2[1], y[rows - 1])
    # strided rows
    wide = torch.zeros(rows, fin + 64, dtype=dtype, device=DEV)
    wide[:, 32:32 + fin] = T["x"]
    assert torch.equal(hk.code1x16_matmat_scan(wide[:, 32:32 + fin], *args), y)
    # the direct matvec kernel on the same rows: same exact products, another summation order
    yd = hk.code1x16_matmat(T["x"][:min(rows, 8)], *args)
    check_close(y[:min(rows, 8)].float().cpu().numpy(), yd.double().cpu().numpy(), dtype, "scan vs direct kernel")
    # the large-batch op takes this kernel when asked to (gemm_variant 4); its default routes give the same values within the bound
    from aqlm_amd import _native

    keep = _native.get_tuning("gemm_variant")
    try:
        _native.set_tuning("gemm_variant", 4)
        assert torch.equal(hk.code1x16_matmat_dequant(T["x"], *args), y)
    finally:
        _native.set_tuning("gemm_variant", keep)
    check_close(hk.code1x16_matmat_dequant(T["x"], *args).float().cpu().numpy(), y64, dtype, what + " (large-batch op, default route)")


def test_scan_kernel_nan_rows_zero_input_and_edge_codes(hk):
    fin, fout, rows = 4096, 2048, 6
    L = orc.make_layer(77, fin, fout, 1, 16, 8, batch=rows, bias=True, float_dtype=np.float16)   # edge_codes: 0, 65535, 32767, 32768 present
    T = to_dev(L, torch.float16)
    args = (T["codes"], T["codebooks"], T["scales"], T["bias"])
    y = hk.code1x16_matmat_scan(T["x"], *args)
    xn = T["x"].clone()
    xn[1, 7] = float("nan")
    xn[4, 100] = float("inf")
    yn = hk.code1x16_matmat_scan(xn, *args)
    assert not torch.isfinite(yn[1]).any() and not torch.isfinite(yn[4]).any()
    for r in (0, 2, 3, 5):
        assert torch.equal(yn[r], y[r]), f"row {r} was touched by another row's NaN / Inf"
    y0 = hk.code1x16_matmat_scan(torch.zeros_like(T["x"]), *args)
    assert torch.equal(y0, T["bias"].reshape(1, -1).expand(rows, -1)), "zero input must give the bias exactly"
    # every code value of slice boundaries lands in the right slice: a layer whose codes are k * 8192 + {0, 1, 8191}
    codes = torch.tensor([s * 8192 + o for s in range(8) for o in (0, 1, 8191)], dtype=torch.int32, device=DEV)
    codes = codes.repeat((fout * (fin // 8) + codes.numel() - 1) // codes.numel())[: fout * (fin // 8)].reshape(fout, fin // 8, 1)
    codes = (codes - (codes >= 32768).int() * 65536).to(torch.int16)
    yb = hk.code1x16_matmat_scan(T["x"], codes, *args[1:])
    y64 = orc.dequantize_gemm(L["x"], codes.cpu().numpy(), L["codebooks"], L["scales"], L["bias"])
    check_close(yb.float().cpu().numpy(), y64, torch.float16, "slice boundary codes")


def test_scan_kernel_declines_what_it_does_not_take(hk):
    L = orc.make_layer(5, 4096 + 64, 256, 1, 16, 8, batch=4, bias=False, float_dtype=np.float16)   # in_features % 256 != 0
    T = to_dev(L, torch.float16)
    assert hk.code1x16_matmat_scan(T["x"], T["codes"], T["codebooks"], T["scales"], None) is None
    y = hk.code1x16_matmat_dequant(T["x"], T["codes"], T["codebooks"], T["scales"], None)          # round 5's kernels answer
    check_close(y.float().cpu().numpy(), orc.dequantize_gemm(L["x"], L["codes"], L["codebooks"], L["scales"], None), torch.float16, "fall-through")
    L16 = orc.make_layer(6, 4096, 256, 1, 16, 16, batch=4, bias=False, float_dtype=np.float16)     # 16-element vectors
    T16 = to_dev(L16, torch.float16)
    assert hk.code1x16_matmat_scan(T16["x"], T16["codes"], T16["codebooks"], T16["scales"], None) is None


def test_scan_kernel_under_hipgraph_and_variants(hk):
    """hipGraph capture (no allocation / synchronisation inside the entry); the default routing of the large-batch op (other kernels,
    another summation order) gives the same values within the tolerance."""
    from aqlm_amd import _native

    L = orc.make_layer(91, 4096, 4096, 1, 16, 8, batch=8, bias=True, float_dtype=np.float16)
    T = to_dev(L, torch.float16)
    args = (T["codes"], T["codebooks"], T["scales"], T["bias"])
    y = hk.code1x16_matmat_scan(T["x"], *args)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        hk.code1x16_matmat_scan(T["x"], *args)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            yg = hk.code1x16_matmat_scan(T["x"], *args)
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(yg, y)
    assert _native.get_tuning("scan_max_rows") == 0   # not on a default route
    y5 = hk.code1x16_matmat_dequant(T["x"], *args)
    check_close(y.float().cpu().numpy(), y5.double().cpu().numpy(), torch.float16, "scan vs the default routing")
