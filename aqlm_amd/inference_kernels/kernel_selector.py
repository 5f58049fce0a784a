"""Shape/device dispatch table: which operator serves a given codebook tensor.

Mirror of the reference's ``aqlm.inference_kernels.kernel_selector`` (kernel_selector.py:21-163): same function
names, same arguments, same return contract -- a callable
``(input, codes, codebooks, scales, bias) -> output`` -- with the CUDA / Triton / numba branches replaced by the
MI355X ops of ``hip_kernel``:

    1x16 g8|g16   decode -> aqlm::code1x16_matmat            batch -> aqlm::code1x16_matmat_dequant (fused MFMA)
    2x8  g8       decode -> aqlm::code2x8_matmat             batch -> aqlm::code2x8_matmat_dequant
    1x8  g8       decode -> aqlm::code1x8_matmat             batch -> aqlm::code1x8_matmat_dequant
    Kx8  any g    decode -> aqlm::codekx8_matmat (reference: Triton)     batch -> dequant + GEMM
    anything else (out_group_size == 1) -> aqlm::generic_matmat (reference: Triton)

ROCm reports ``device.type == "cuda"``.  Tensors on the host take the CPU branch (reference kernel_selector.py:95-102):
K x 8-bit codebooks -> native LUT kernel (``cpu_kernel.cpu_gemm_lut``, the numba kernel's replacement; it expects the
codes permuted to [in_groups, out, K] like the reference's), single 16-bit codebooks -> native direct kernel on the
canonical codes, anything else -> the pure-torch ``dequantize_gemm``.  On a GPU box nothing is ever routed to the CPU
branch: a device tensor either reaches an MI355X kernel or raises.
"""
from __future__ import annotations

import warnings
from contextlib import contextmanager
from typing import Callable, Optional

import torch


@contextmanager
def optimize_for_training():
    """Deprecated no-op kept for API compatibility (reference kernel_selector.py:8-18)."""
    warnings.warn("`optimize_for_training` is deprecated. The optimization now happens automatically at runtime.")
    yield


def _torch_forward(input, codes, codebooks, scales, bias):
    from .dequantization import dequantize_gemm

    return dequantize_gemm(input, codes, codebooks, scales, bias)


def _torch_backward(grad_output, codes, codebooks, scales, bias):
    """grad_input = grad_output @ W (reference kernel_selector.py:145-161, which transposes the operands of its forward)."""
    from ..utils import _dequantize_weight, unpack_int_data

    W = _dequantize_weight(unpack_int_data(codes, codebooks.shape[1].bit_length() - 1), codebooks, scales)
    return torch.matmul(grad_output, W.to(grad_output.dtype))


def _cpu_forward_kernel(codebooks: torch.Tensor, optimize_for_training: bool = False):
    """Host tensors (reference kernel_selector.py:95-102).  NOTE the contract of the 8-bit route: like the reference's
    numba kernel it takes ``codes`` permuted to [in_groups, out, K] uint8 -- ``QuantizedLinear.prepare_matmul_op`` keeps
    that copy next to the canonical codes."""
    from . import cpu_kernel

    num_codebooks, codebook_size, out_group_size, in_group_size = codebooks.shape
    if out_group_size == 1 and codebook_size == 256:
        return cpu_kernel.cpu_gemm_lut
    if (not optimize_for_training and out_group_size == 1 and num_codebooks == 1 and in_group_size in (8, 16)
            and codebook_size in (4096, 65536)):
        # decode only: the direct kernel re-gathers the codebook vector per (row, batch column); for many rows the
        # reference's dequantize + GEMM is the faster host path (64 rows: 0.24 s vs 0.43 s at 4096 x 4096, 8 threads)
        return cpu_kernel.cpu_gemv_1xn
    return _torch_forward


def cpu_kernel_takes_permuted_codes(codebooks: torch.Tensor) -> bool:
    return codebooks.device.type == "cpu" and codebooks.shape[2] == 1 and codebooks.shape[1] == 256


def _require_gpu(codebooks: torch.Tensor):
    if codebooks.device.type != "cuda":
        raise NotImplementedError(
            f"aqlm_amd has MI355X (ROCm, device type 'cuda') and host (cpu) kernels; got codebooks on "
            f"'{codebooks.device.type}'."
        )


def get_forward_pass_kernel(
    codebooks: torch.Tensor,
    optimize_for_training: bool,
) -> Callable[[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, Optional[torch.Tensor]], torch.Tensor]:
    """reference kernel_selector.py:21-102."""
    if codebooks.device.type == "cpu":
        return _cpu_forward_kernel(codebooks, optimize_for_training)
    _require_gpu(codebooks)
    from . import hip_kernel  # noqa: F401  (registers torch.ops.aqlm.*; raises if libaqlm_hip.so is missing)

    num_codebooks, codebook_size, out_group_size, in_group_size = codebooks.shape
    if out_group_size != 1:
        return _torch_forward  # no tuned kernel for out_group_size > 1 anywhere (reference: Triton / torch): torch on device
    ops = torch.ops.aqlm
    if (num_codebooks, codebook_size) == (1, 65536) and in_group_size in (8, 16):
        return ops.code1x16_matmat_dequant if optimize_for_training else ops.code1x16_matmat
    if (num_codebooks, codebook_size, in_group_size) == (2, 256, 8):
        return ops.code2x8_matmat_dequant if optimize_for_training else ops.code2x8_matmat
    if (num_codebooks, codebook_size, in_group_size) == (1, 256, 8):
        return ops.code1x8_matmat_dequant if optimize_for_training else ops.code1x8_matmat
    if codebook_size == 256 and in_group_size % 8 == 0 and num_codebooks <= 16:
        return hip_kernel.code2x8_matmat_dequant if optimize_for_training else ops.codekx8_matmat
    if optimize_for_training:
        return ops.generic_matmat_dequant  # generic dequant + library GEMM (the reference's dequantize_gemm)
    return ops.generic_matmat


def get_backward_pass_kernel(
    codebooks: torch.Tensor,
    optimize_for_training: bool,
) -> Callable[[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, Optional[torch.Tensor]], torch.Tensor]:
    """reference kernel_selector.py:105-163: callable(grad_output, codes, codebooks, scales, bias) -> grad_input.
    One implementation serves both modes: dequantise (scales folded in) and multiply."""
    if codebooks.device.type == "cpu":
        return _torch_backward
    _require_gpu(codebooks)
    from . import hip_kernel  # noqa: F401

    num_codebooks, codebook_size, out_group_size, in_group_size = codebooks.shape
    if out_group_size != 1:
        return _torch_backward
    ops = torch.ops.aqlm
    if (num_codebooks, codebook_size) == (1, 65536) and in_group_size in (8, 16):
        kern = ops.code1x16_matmat_dequant_transposed
    elif (num_codebooks, codebook_size, in_group_size) == (2, 256, 8):
        kern = ops.code2x8_matmat_dequant_transposed
    elif (num_codebooks, codebook_size, in_group_size) == (1, 256, 8):
        kern = ops.code1x8_matmat_dequant_transposed
    elif codebook_size == 256 and in_group_size % 8 == 0 and num_codebooks <= 16:
        kern = hip_kernel.code2x8_matmat_dequant_transposed
    else:  # any other scheme: generic dequant + GEMM (the reference transposes the tensors and reuses its forward kernel)
        kern = ops.generic_matmat_dequant_transposed

    def _backward(grad_output, codes, codebooks, scales, bias):
        # the layer's bias does not enter grad_input (reference kernel_selector.py:160 passes None as well)
        return kern(grad_output, codes, codebooks, scales, None)

    return _backward
