"""Integer-container helpers and the pure-torch weight reconstruction, mirroring the public helpers of the
reference's ``aqlm.utils`` (inference_lib/src/aqlm/utils.py:11-70) that callers import by name
(benchmark/matmul_benchmark.py:7, convert scripts).  These are utilities (fixture generation, checkpoints), not
the hot path: the MI355X path never calls ``_dequantize_weight``.
"""
from __future__ import annotations

from typing import Optional

import torch


def get_int_dtype(nbits: int) -> torch.dtype:
    """Smallest signed torch integer dtype that holds ``nbits`` bits (reference utils.py:11-20)."""
    for limit, dtype in ((8, torch.int8), (16, torch.int16), (32, torch.int32), (64, torch.int64)):
        if nbits <= limit:
            return dtype
    raise ValueError(f"No dtype available for {nbits}-bit codebooks")


@torch.inference_mode()
def pack_int_data(data: torch.Tensor, nbits: int) -> torch.Tensor:
    """Unsigned indices -> two's-complement containers (reference utils.py:23-26).  The reference wraps in place;
    this returns a new tensor and leaves ``data`` untouched."""
    wrapped = torch.where(data >= 2 ** (nbits - 1), data - 2**nbits, data)
    return wrapped.to(get_int_dtype(nbits))


@torch.inference_mode()
def unpack_int_data(data: torch.Tensor, nbits: int) -> torch.Tensor:
    """Containers -> unsigned int64 indices (reference utils.py:29-31)."""
    return data.to(torch.int64) % (2**nbits)


def _dequantize_weight(codes: torch.Tensor, codebooks: torch.Tensor, scales: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Reconstruct [*dims, out_features, in_features] from UNSIGNED codes [*dims, out_groups, in_groups, K],
    codebooks [K, codebook_size, out_group_size, in_group_size] and broadcastable scales (reference
    utils.py:43-70).  Differentiable w.r.t. codebooks and scales.  Implemented with plain indexing."""
    num_out_groups, num_in_groups, num_codebooks = codes.shape[-3:]
    K, codebook_size, out_group_size, in_group_size = codebooks.shape
    assert K == num_codebooks
    acc = None
    for c in range(num_codebooks):
        part = codebooks[c][codes[..., c]]  # [*dims, og, ig, ogs, igs]
        acc = part if acc is None else acc + part
    if scales is not None:
        acc = acc * scales
    out_features = num_out_groups * out_group_size
    in_features = num_in_groups * in_group_size
    return acc.swapaxes(-3, -2).reshape(list(codes.shape[:-3]) + [out_features, in_features])
