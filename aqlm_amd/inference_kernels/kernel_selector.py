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

ROCm reports ``device.type == "cuda"``.  There is no CPU or fallback branch: this package is the GPU path only and
raises for anything it does not implement.
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


def _require_gpu(codebooks: torch.Tensor):
    if codebooks.device.type != "cuda":
        raise NotImplementedError(
            f"aqlm_amd implements the MI355X (ROCm, device type 'cuda') path only; got codebooks on "
            f"'{codebooks.device.type}'. Move the module to the GPU."
        )


def get_forward_pass_kernel(
    codebooks: torch.Tensor,
    optimize_for_training: bool,
) -> Callable[[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, Optional[torch.Tensor]], torch.Tensor]:
    """reference kernel_selector.py:21-102."""
    _require_gpu(codebooks)
    from . import hip_kernel  # noqa: F401  (registers torch.ops.aqlm.*; raises if libaqlm_hip.so is missing)

    num_codebooks, codebook_size, out_group_size, in_group_size = codebooks.shape
    if out_group_size != 1:
        raise NotImplementedError("aqlm_amd kernels require out_group_size == 1 (as every reference GPU kernel does)")
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
    _require_gpu(codebooks)
    from . import hip_kernel  # noqa: F401

    num_codebooks, codebook_size, out_group_size, in_group_size = codebooks.shape
    if out_group_size != 1:
        raise NotImplementedError("aqlm_amd kernels require out_group_size == 1")
    ops = torch.ops.aqlm
    if (num_codebooks, codebook_size) == (1, 65536) and in_group_size in (8, 16):
        kern = ops.code1x16_matmat_dequant_transposed
    elif (num_codebooks, codebook_size, in_group_size) == (2, 256, 8):
        kern = ops.code2x8_matmat_dequant_transposed
    elif (num_codebooks, codebook_size, in_group_size) == (1, 256, 8):
        kern = ops.code1x8_matmat_dequant_transposed
    elif codebook_size == 256 and in_group_size % 8 == 0:
        kern = hip_kernel.code2x8_matmat_dequant_transposed
    else:  # any other scheme: generic dequant + GEMM (the reference transposes the tensors and reuses its forward kernel)
        kern = ops.generic_matmat_dequant_transposed

    def _backward(grad_output, codes, codebooks, scales, bias):
        # the layer's bias does not enter grad_input (reference kernel_selector.py:160 passes None as well)
        return kern(grad_output, codes, codebooks, scales, None)

    return _backward
