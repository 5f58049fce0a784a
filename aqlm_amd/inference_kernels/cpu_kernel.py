"""CPU kernels of the QuantizedLinear path, backed by libaqlm_cpu.so (include/aqlm_cpu.h).

Host-side mirror of the reference's CPU branch (inference_lib/src/aqlm/inference_kernels/kernel_selector.py:95-102):
the numba LUT kernel for 8-bit codebooks is replaced by a native OpenMP / AVX2 kernel (no numba dependency), 16-bit
single-codebook schemes get a direct kernel instead of a full dequantisation, and everything else (out_group_size > 1,
odd schemes) takes the pure-torch ``dequantize_gemm`` like the reference.  This is the package's fallback for tensors
that live on the host -- the MI355X kernels are never routed through it.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_HERE, "libaqlm_cpu.so")
ABI_VERSION = 2
_lib = None


def lib():
    """libaqlm_cpu.so, loaded on first use (built by ``make -C aqlm_amd/csrc_cpu`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: run `make -C aqlm_amd/csrc_cpu` (or __graft_entry__.build())")
        L = ctypes.CDLL(LIB_PATH)
        vp, ci, cl, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_size_t
        L.aqlm_cpu_abi_version.restype = ci
        L.aqlm_cpu_max_threads.restype = ci
        L.aqlm_cpu_lut_scratch_floats.restype = sz
        L.aqlm_cpu_lut_scratch_floats.argtypes = [ci, ci, ci]
        L.aqlm_cpu_gemv_lut_kx8.restype = ci
        L.aqlm_cpu_gemv_lut_kx8.argtypes = [vp, vp, vp, vp, vp, vp, ci, cl, cl, ci, ci, ci, ci, vp, ci]
        L.aqlm_cpu_gemv_1xn.restype = ci
        L.aqlm_cpu_gemv_1xn.argtypes = [vp, vp, vp, ci, vp, vp, vp, ci, cl, cl, ci, ci, ci, ci, ci]
        L.aqlm_cpu_gemv_1xn_f16.restype = ci
        L.aqlm_cpu_gemv_1xn_f16.argtypes = [vp, vp, vp, ci, vp, vp, vp, ci, cl, cl, ci, ci, ci, ci, ci]
        if L.aqlm_cpu_abi_version() != ABI_VERSION:
            raise ImportError(f"{LIB_PATH}: ABI version {L.aqlm_cpu_abi_version()}, expected {ABI_VERSION}; rebuild it")
        _lib = L
    return _lib


def _f32(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.detach().to(torch.float32).contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _threads(nthreads: int) -> int:
    """0 = the process's torch thread setting (the knob the reference's CPU benchmark turns together with numba's,
    matmul_benchmark_cpu.py:77-88) -- not OpenMP's default, which is every logical CPU the host shows even when the
    container may only use a few of them (measured on the 256-thread GPU host: 250 ms instead of 3 ms per call)."""
    return int(nthreads) if nthreads and nthreads > 0 else max(1, int(torch.get_num_threads()))


def permute_codes_for_lut(codes: torch.Tensor) -> torch.Tensor:
    """[out, in_groups, K] -> [in_groups, out, K] uint8: the layout of the LUT kernel (what the reference's
    ``prepare_matmul_op`` does to ``codes`` IN PLACE for CPU modules, inference.py:78-83; here a derived copy)."""
    return codes.permute(1, 0, 2).contiguous().view(torch.uint8)


# (torch.compiler.disable: these functions hand raw pointers to libaqlm_cpu.so through ctypes -- Dynamo must call them, not trace them;
# traced, the resumed frame passed pointers of tensors it no longer owned and the kernel wrote through them)
@torch.compiler.disable
def cpu_gemm_lut(input: torch.Tensor, codes_alt: torch.Tensor, codebooks: torch.Tensor, scales: torch.Tensor,
                 bias: Optional[torch.Tensor], nthreads: int = 0) -> torch.Tensor:
    """K x 8-bit schemes through per-row look-up tables (reference: ``numba_gemm_lut``, numba_kernel.py:10-65, same
    contract: ``codes_alt`` is [in_groups, out, K] uint8).  Any floating input dtype; computed in fp32."""
    K, cbsize, ogs, g = codebooks.shape
    if cbsize != 256 or ogs != 1:
        raise NotImplementedError("cpu_gemm_lut needs codebooks [K, 256, 1, g]")
    in_groups, out_features, k2 = codes_alt.shape
    if k2 != K or codes_alt.dtype != torch.uint8:
        raise ValueError("codes_alt must be uint8 [in_groups, out_features, num_codebooks]")
    in_features = in_groups * g
    if input.shape[-1] != in_features:
        raise ValueError(f"input has {input.shape[-1]} features, layer expects {in_features}")
    x = _f32(input.reshape(-1, in_features))
    cb, sc, bi = _f32(codebooks), _f32(scales.reshape(-1)), _f32(bias)
    y = torch.empty((x.shape[0], out_features), dtype=torch.float32)
    scratch = torch.empty((lib().aqlm_cpu_lut_scratch_floats(in_features, K, g),), dtype=torch.float32)
    rc = lib().aqlm_cpu_gemv_lut_kx8(x.data_ptr(), cb.data_ptr(), codes_alt.contiguous().data_ptr(), sc.data_ptr(), _ptr(bi),
                                     y.data_ptr(), x.shape[0], x.stride(0), out_features, in_features, out_features, K, g,
                                     scratch.data_ptr(), _threads(nthreads))
    if rc:
        raise RuntimeError(f"aqlm_cpu_gemv_lut_kx8 failed with {rc}")
    return y.to(input.dtype).reshape(input.shape[:-1] + (out_features,))


# fp16 view of the codebooks for the direct kernel (half the bytes the 65536-entry gathers touch).
#   * an fp16 parameter (what checkpoints store) is used AS IS: no copy, nothing to go stale;
#   * any other dtype gets a converted copy, kept per codebook tensor (keyed by id with a weak reference that drops the entry
#     when the tensor dies -- a WeakKeyDictionary would compare tensors with ==) and used only while it is provably current:
#     same storage, same version counter AND the same checksum of the source, recomputed on every call (two reductions over
#     <= 2 MiB, ~2 % of the kernel's time) -- writes through `.data` (older optimizer / loader code: `p.data.copy_()`) do not
#     bump the version counter, the checksum sees them.  `invalidate_half_tables()` drops every copy.
#   None = the values are not fp16-representable (or the CPU lacks F16C): fp32 table.
HALF_TABLE = True
_HALF_TABLES = {}


def invalidate_half_tables() -> None:
    """Forget every cached fp16 codebook copy (they are rebuilt from the live tensors at the next call)."""
    _HALF_TABLES.clear()


def _checksum(t: torch.Tensor):
    f = t.detach().reshape(-1).to(torch.float32)
    return (float(f.sum()), float(f.abs().sum()))


def _half_table(codebooks: torch.Tensor) -> Optional[torch.Tensor]:
    if not HALF_TABLE:
        return None
    if codebooks.dtype == torch.float16:
        return codebooks.detach().contiguous()  # the live storage itself (contiguous parameters: no copy)
    try:
        version = codebooks._version
    except RuntimeError:  # inference tensors carry no version counter: do not cache what cannot be invalidated
        return None
    key = (codebooks.data_ptr(), version, codebooks.dtype, tuple(codebooks.shape), _checksum(codebooks))
    ident = id(codebooks)
    hit = _HALF_TABLES.get(ident)
    if hit is not None and hit[0]() is codebooks and hit[1] == key:
        return hit[2]
    half = codebooks.detach().to(torch.float16).contiguous()
    if not torch.equal(half.to(torch.float32), codebooks.detach().to(torch.float32)):
        half = None  # a retrained fp32 codebook, bf16 values outside fp16's range, ...
    _HALF_TABLES[ident] = (weakref.ref(codebooks, lambda _, ident=ident: _HALF_TABLES.pop(ident, None)), key, half)
    return half


@torch.compiler.disable
def cpu_gemv_1xn(input: torch.Tensor, codes: torch.Tensor, codebooks: torch.Tensor, scales: torch.Tensor,
                 bias: Optional[torch.Tensor], nthreads: int = 0) -> torch.Tensor:
    """One codebook of up to 65536 entries, g = 8 | 16, canonical codes [out, in_groups, 1] (int8 / int16 containers)."""
    K, cbsize, ogs, g = codebooks.shape
    nbits = int(cbsize).bit_length() - 1
    if K != 1 or ogs != 1 or g not in (8, 16) or 2**nbits != cbsize or codes.dtype not in (torch.int8, torch.int16):
        raise NotImplementedError("cpu_gemv_1xn needs codebooks [1, 2**n, 1, 8 | 16] and 8- / 16-bit code containers")
    out_features, in_groups = codes.shape[0], codes.shape[1]
    in_features = in_groups * g
    if input.shape[-1] != in_features:
        raise ValueError(f"input has {input.shape[-1]} features, layer expects {in_features}")
    x = _f32(input.reshape(-1, in_features))
    sc, bi = _f32(scales.reshape(-1)), _f32(bias)
    c = codes.contiguous()
    y = torch.empty((x.shape[0], out_features), dtype=torch.float32)
    rc = -2
    half = _half_table(codebooks)
    if half is not None:  # same values from half the bytes (exactness checked when the copy was made); -2: no F16C here
        rc = lib().aqlm_cpu_gemv_1xn_f16(x.data_ptr(), half.data_ptr(), c.data_ptr(), c.element_size(), sc.data_ptr(), _ptr(bi),
                                         y.data_ptr(), x.shape[0], x.stride(0), out_features, in_features, out_features, nbits, g,
                                         _threads(nthreads))
    if rc == -2:
        cb = _f32(codebooks)
        rc = lib().aqlm_cpu_gemv_1xn(x.data_ptr(), cb.data_ptr(), c.data_ptr(), c.element_size(), sc.data_ptr(), _ptr(bi),
                                     y.data_ptr(), x.shape[0], x.stride(0), out_features, in_features, out_features, nbits, g,
                                     _threads(nthreads))
    if rc:
        raise RuntimeError(f"aqlm_cpu_gemv_1xn failed with {rc}")
    return y.to(input.dtype).reshape(input.shape[:-1] + (out_features,))
