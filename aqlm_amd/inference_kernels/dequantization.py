"""``dequantize_gemm``: the definition of correctness of the path, at the reference's import location
(inference_lib/src/aqlm/inference_kernels/dequantization.py:9-21): reconstruct the weight in torch, then ``F.linear``.
Works on any device and for any scheme (incl. out_group_size > 1); it is the fallback of the CPU branch of the
selector, never the MI355X hot path."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from ..utils import _dequantize_weight, unpack_int_data


def dequantize_gemm(
    input: torch.Tensor,       # [..., in_features]
    codes: torch.IntTensor,    # [num_out_groups, num_in_groups, num_codebooks]
    codebooks: torch.Tensor,   # [num_codebooks, codebook_size, out_group_size, in_group_size]
    scales: torch.Tensor,      # [num_out_groups, 1, 1, 1]
    bias: Optional[torch.Tensor],
) -> torch.Tensor:
    dequantized_weight = _dequantize_weight(unpack_int_data(codes, codebooks.shape[1].bit_length() - 1), codebooks, scales)
    return F.linear(input, dequantized_weight, bias)
