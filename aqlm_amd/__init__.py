"""aqlm_amd -- MI355X-native implementation of the AQLM QuantizedLinear inference path.

Drop-in for the reference's ``aqlm`` package on that one path (the top-level ``aqlm`` package in this repository
re-exports these names so that ``from aqlm import QuantizedLinear`` -- what Hugging Face does -- resolves here).
"""
from . import checkpoint, inference_kernels
from .fusion import SharedInputGroup, fuse_shared_input_linears, unfuse_shared_input_linears
from .inference import QuantizedLinear
from .inference_kernels import get_backward_pass_kernel, get_forward_pass_kernel, optimize_for_training

__version__ = "1.1.7"

__all__ = [
    "QuantizedLinear",
    "get_backward_pass_kernel",
    "get_forward_pass_kernel",
    "optimize_for_training",
    "inference_kernels",
    "checkpoint",
    "SharedInputGroup",
    "fuse_shared_input_linears",
    "unfuse_shared_input_linears",
]
