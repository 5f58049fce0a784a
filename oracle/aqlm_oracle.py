"""CPU ORACLE for the AQLM dequant-fused matvec/matmul path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the reference algorithm for the one hot path this
repository accelerates.  It exists so that the HIP kernels can
be checked against something that is obviously correct.  It is NOT part of the product:

    only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
    import it, and there only as the checker / the reported CPU baseline.  Nothing under
    ``aqlm_amd/`` or ``aqlm/`` imports ``oracle``; the product path raises when the HIP
    library is missing.

Parity status: PINNED.  Every function below is checked (tests/test_oracle_golden.py) against
outputs of the reference's own Python implementation (``inference_lib/src/aqlm/utils.py`` and
``inference_lib/src/aqlm/inference_kernels/dequantization.py``) executed in the authoring
container by ``oracle/gen_golden.py``; those outputs are committed under ``tests/golden/``.
The reference holds no golden vectors or known-answer tests of its own.
The reference's numba LUT kernel could not be executed (numba is not installed and there is no
network); ``lut_gemv`` restates it from source and is pinned transitively: it must equal
``dequantize_gemm`` (which is pinned) up to fp32 summation order.

Each function cites the reference file:line (paths relative to /root/reference) it follows.
All arithmetic is carried out in ``acc_dtype`` (float64 by default = "truth"; pass the
storage dtype to mimic the reference's own rounding).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

__all__ = [
    "get_int_dtype",
    "pack_int_data",
    "unpack_int_data",
    "dequantize_weight",
    "dequantize_gemm",
    "dequantize_gemm_transposed",
    "permute_codes_for_lut",
    "lut_gemv",
    "lut_gemm",
    "make_layer",
    "algorithmic_bytes",
]


# --------------------------------------------------------------------------------------------
# integer containers: inference_lib/src/aqlm/utils.py:11-31
# --------------------------------------------------------------------------------------------
def get_int_dtype(nbits: int) -> np.dtype:
    """utils.py:11-20 -- smallest signed container that holds ``nbits`` bits."""
    if nbits <= 8:
        return np.dtype(np.int8)
    if nbits <= 16:
        return np.dtype(np.int16)
    if nbits <= 32:
        return np.dtype(np.int32)
    if nbits <= 64:
        return np.dtype(np.int64)
    raise ValueError(f"No dtype available for {nbits}-bit codebooks")


def pack_int_data(data: np.ndarray, nbits: int) -> np.ndarray:
    """utils.py:23-26 -- unsigned index -> two's-complement container (values >= 2**(nbits-1) wrap
    negative).  Unlike the reference this does not mutate its argument."""
    data = np.array(data, dtype=np.int64, copy=True)
    data[data >= 2 ** (nbits - 1)] -= 2**nbits
    return data.astype(get_int_dtype(nbits))


def unpack_int_data(data: np.ndarray, nbits: int) -> np.ndarray:
    """utils.py:29-31 -- container -> unsigned index via ``int64 % 2**nbits`` (python modulo:
    result is non-negative)."""
    return np.asarray(data).astype(np.int64) % (2**nbits)


# --------------------------------------------------------------------------------------------
# the definition of the layer: utils.py:43-70 + dequantization.py:9-21
# --------------------------------------------------------------------------------------------
def dequantize_weight(
    codes_unsigned: np.ndarray,
    codebooks: np.ndarray,
    scales: Optional[np.ndarray] = None,
    acc_dtype=np.float64,
) -> np.ndarray:
    """utils.py:43-70 (``_dequantize_weight``).

    codes_unsigned: [num_out_groups, num_in_groups, num_codebooks] unsigned indices
    codebooks:      [num_codebooks, codebook_size, out_group_size, in_group_size]
    scales:         broadcastable to [num_out_groups, num_in_groups, out_group_size, in_group_size]
    returns W:      [num_out_groups*out_group_size, num_in_groups*in_group_size] in acc_dtype

    The reference sums the selected codebook vectors with ``F.embedding_bag(mode="sum")``
    (utils.py:57-59), multiplies by scales (:64-65) and swaps the two middle axes (:66).
    """
    codes_unsigned = np.asarray(codes_unsigned)
    num_out_groups, num_in_groups, num_codebooks = codes_unsigned.shape[-3:]
    K, codebook_size, out_group_size, in_group_size = codebooks.shape
    assert K == num_codebooks
    cb = np.asarray(codebooks).astype(acc_dtype)
    acc = np.zeros((num_out_groups, num_in_groups, out_group_size, in_group_size), dtype=acc_dtype)
    for c in range(num_codebooks):  # embedding_bag sums bags in codebook order
        acc += cb[c][codes_unsigned[..., c]]
    if scales is not None:
        acc = acc * np.asarray(scales).astype(acc_dtype)
    # [og, ig, o, i] -> [og, o, ig, i] -> [out, in]
    return acc.swapaxes(-3, -2).reshape(num_out_groups * out_group_size, num_in_groups * in_group_size)


def _nbits_of(codebooks: np.ndarray) -> int:
    # dequantization.py:17 -- codebooks.shape[1].bit_length() - 1
    return int(codebooks.shape[1]).bit_length() - 1


def dequantize_gemm(
    x: np.ndarray,
    codes: np.ndarray,
    codebooks: np.ndarray,
    scales: np.ndarray,
    bias: Optional[np.ndarray],
    acc_dtype=np.float64,
) -> np.ndarray:
    """dequantization.py:9-21 -- ``F.linear(x, _dequantize_weight(unpack(codes), codebooks, scales), bias)``.

    ``codes`` are the signed containers exactly as stored in a checkpoint."""
    W = dequantize_weight(unpack_int_data(codes, _nbits_of(codebooks)), codebooks, scales, acc_dtype)
    y = np.asarray(x).astype(acc_dtype) @ W.T
    if bias is not None:
        y = y + np.asarray(bias).astype(acc_dtype)
    return y


def dequantize_gemm_transposed(
    grad_out: np.ndarray,
    codes: np.ndarray,
    codebooks: np.ndarray,
    scales: np.ndarray,
    bias: Optional[np.ndarray],
    acc_dtype=np.float64,
) -> np.ndarray:
    """Backward-pass operator: ``grad_in = (grad_out * scales) @ W_unscaled (+ bias)``.

    Follows the *intended* semantics of ``code1x16_matmat_dequant_transposed``
    (cuda_kernel.cpp:303-354: scale the incoming gradient by the per-output-channel scales, then
    multiply by the transposed dequantized weight), which equals the generic definition in
    kernel_selector.py:145-161.  The reference's 2x8/1x8 variants drop the scales and the 1x16
    variant hard-codes group size 8; those defects are not
    reproduced."""
    W = dequantize_weight(unpack_int_data(codes, _nbits_of(codebooks)), codebooks, scales, acc_dtype)
    y = np.asarray(grad_out).astype(acc_dtype) @ W
    if bias is not None:
        y = y + np.asarray(bias).astype(acc_dtype)
    return y


# --------------------------------------------------------------------------------------------
# the reference's CPU kernel (numba LUT gemv): numba_kernel.py:37-48, inference.py:78-83
# --------------------------------------------------------------------------------------------
def permute_codes_for_lut(codes: np.ndarray) -> np.ndarray:
    """inference.py:78-83 -- the CPU path stores codes as [num_in_groups, num_out_groups, K]."""
    return np.ascontiguousarray(np.transpose(codes, (1, 0, 2)))


def lut_gemv(
    x: np.ndarray,  # [in_features]
    codebooks: np.ndarray,  # [K, codebook_size, 1, g]
    codes_alt_unsigned: np.ndarray,  # [in_groups, out, K] unsigned (numba views int8 as uint8, :59)
    scales: np.ndarray,  # [out,1,1,1]
    acc_dtype=np.float32,
) -> np.ndarray:
    """numba_kernel.py:37-48 / benchmark/matmul_benchmark_cpu.py:100-111.

        lut = x.reshape(-1, g) @ codebooks.reshape(-1, g).T          (:39)
        lut = lut.reshape(-1, K, codebook_size)                       (:40)
        y[i] += lut[j, c, codes_alt[j, i, c]]   for j, i, c           (:43-46)
        y *= scales.flatten()                                         (:47)

    The reference runs this in float32 (numba_kernel.py:30-32); acc_dtype defaults to that."""
    K, codebook_size, out_group_size, g = codebooks.shape
    assert out_group_size == 1
    x = np.asarray(x).astype(acc_dtype)
    lut = x.reshape(-1, g) @ np.asarray(codebooks).astype(acc_dtype).reshape(-1, g).T
    lut = lut.reshape(-1, K, codebook_size)  # [in_groups, K, codebook_size]
    in_groups, out_features, K2 = codes_alt_unsigned.shape
    assert K2 == K and in_groups == lut.shape[0]
    y = np.zeros(out_features, dtype=acc_dtype)
    jj = np.arange(in_groups)[:, None]
    for c in range(K):
        y += lut[jj, c, codes_alt_unsigned[:, :, c]].sum(axis=0, dtype=acc_dtype)
    y *= np.asarray(scales).astype(acc_dtype).reshape(-1)
    return y


def lut_gemm(x, codes, codebooks, scales, bias, acc_dtype=np.float32):
    """numba_kernel.py:10-65 (``numba_gemm_lut``): python loop over the rows of a flattened input
    (:54-62), ``+= bias`` (:63-64).  ``codes`` here are in the *permuted* CPU layout
    [in_groups, out, K] as signed containers."""
    nbits = _nbits_of(codebooks)
    codes_alt = unpack_int_data(codes, nbits)
    x = np.asarray(x)
    flat = x.reshape(-1, x.shape[-1])
    out = np.stack([lut_gemv(row, codebooks, codes_alt, scales, acc_dtype) for row in flat])
    if bias is not None:
        out = out + np.asarray(bias).astype(acc_dtype)
    return out.reshape(x.shape[:-1] + (-1,))


# --------------------------------------------------------------------------------------------
# seeded synthetic layers (mirrors benchmark/matmul_benchmark.py:83-99, but reproducible)
# --------------------------------------------------------------------------------------------
def make_layer(
    seed: int,
    in_features: int,
    out_features: int,
    num_codebooks: int,
    nbits: int,
    in_group_size: int,
    batch: int = 1,
    bias: bool = True,
    out_group_size: int = 1,
    float_dtype=np.float16,
    edge_codes: bool = True,
):
    """Seeded synthetic layer: uniform random codes (the benchmark's worst case for any cache),
    standard-normal codebooks / x / scales / bias (matmul_benchmark.py:83-97 uses ones for scales and
    no bias when timing; parity runs use random ones).  With ``edge_codes`` the first codes of row 0
    are forced to the container edge values {0, 2**(n-1)-1, 2**(n-1), 2**n-1}.

    float_dtype may be np.float16, np.float32, or the string "bfloat16" (values are then rounded to
    bf16 and returned as float32 arrays holding exactly-representable bf16 values)."""
    rng = np.random.default_rng(seed)
    og = out_features // out_group_size
    ig = in_features // in_group_size
    codes_u = rng.integers(0, 2**nbits, size=(og, ig, num_codebooks), dtype=np.int64)
    if edge_codes and ig * num_codebooks >= 4:
        edges = [0, 2 ** (nbits - 1) - 1, 2 ** (nbits - 1), 2**nbits - 1]
        flat = codes_u.reshape(og, -1)
        flat[0, :4] = edges
        flat[-1, -4:] = edges[::-1]
    codes = pack_int_data(codes_u, nbits)
    codebooks = rng.standard_normal((num_codebooks, 2**nbits, out_group_size, in_group_size), dtype=np.float32)
    scales = rng.standard_normal((og, 1, 1, 1), dtype=np.float32)
    x = rng.standard_normal((batch, in_features), dtype=np.float32)
    b = rng.standard_normal((out_features,), dtype=np.float32) if bias else None

    def rnd(a):
        if a is None:
            return None
        if isinstance(float_dtype, str) and float_dtype == "bfloat16":
            return round_to_bf16(a)
        return a.astype(float_dtype)

    return {
        "codes": codes,
        "codes_unsigned": codes_u,
        "codebooks": rnd(codebooks),
        "scales": rnd(scales),
        "x": rnd(x),
        "bias": rnd(b),
        "nbits": nbits,
    }


def round_to_bf16(a: np.ndarray) -> np.ndarray:
    """Round float32 -> bfloat16 (round-to-nearest-even), returned as float32."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    rounded = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return rounded.astype(np.uint32).view(np.float32).reshape(a.shape)


def bf16_bits(a: np.ndarray) -> np.ndarray:
    """float32 array holding bf16-representable values -> uint16 bit patterns."""
    return (np.ascontiguousarray(a, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def bf16_from_bits(u: np.ndarray) -> np.ndarray:
    return (np.asarray(u).astype(np.uint32) << 16).view(np.float32)


def algorithmic_bytes(
    in_features: int,
    out_features: int,
    num_codebooks: int,
    nbits: int,
    in_group_size: int,
    batch: int = 1,
    bias: bool = False,
    elem: int = 2,
) -> int:
    """SURVEY.md section 8(d):
    codes + codebooks(once) + B*in*2 + B*out*2 + out*2 (+ out*2 if bias)."""
    code_bytes = 1 if nbits <= 8 else 2
    n = out_features * (in_features // in_group_size) * num_codebooks * code_bytes
    n += num_codebooks * (2**nbits) * in_group_size * elem
    n += batch * in_features * elem + batch * out_features * elem + out_features * elem
    if bias:
        n += out_features * elem
    return n
