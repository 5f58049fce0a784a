"""ctypes front-end for oracle/libaqlm_oracle.so (the C restatement of the reference CPU path).

TEST / BASELINE INFRASTRUCTURE ONLY -- see the header of oracle/aqlm_oracle.c.  Imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libaqlm_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "aqlm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libaqlm_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        vp, ci, fp = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
        L.aqlm_oracle_lut_bytes.restype = ctypes.c_size_t
        L.aqlm_oracle_lut_bytes.argtypes = [ci, ci, ci, ci]
        L.aqlm_oracle_max_threads.restype = ci
        L.aqlm_oracle_lut_gemv_f32.restype = None
        L.aqlm_oracle_lut_gemv_f32.argtypes = [fp, fp, vp, ci, fp, fp, ci, ci, ci, ci, ci, fp, ci]
        L.aqlm_oracle_dequant_gemv_f32.restype = None
        L.aqlm_oracle_dequant_gemv_f32.argtypes = [fp, fp, vp, ci, fp, fp, fp, ci, ci, ci, ci, ci, ci]
        L.aqlm_oracle_dequant_weight_f32.restype = None
        L.aqlm_oracle_dequant_weight_f32.argtypes = [fp, vp, ci, fp, fp, ci, ci, ci, ci, ci]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _codes_unsigned_view(codes: np.ndarray):
    codes = np.ascontiguousarray(codes)
    if codes.dtype.itemsize == 1:
        return codes.view(np.uint8), 1
    if codes.dtype.itemsize == 2:
        return codes.view(np.uint16), 2
    raise ValueError("C oracle handles 8- and 16-bit code containers")


def max_threads() -> int:
    return int(lib().aqlm_oracle_max_threads())


class LutGemv:
    """Holds the converted operands + LUT scratch so that repeated calls time only the kernel
    (benchmark/matmul_benchmark_cpu.py:136-146 converts once, then times the call)."""

    def __init__(self, codebooks, codes_alt, scales, nbits, nthreads=1):
        self.codebooks = _f32(codebooks)
        self.codes_alt, self.code_bytes = _codes_unsigned_view(codes_alt)
        self.scales = _f32(scales).reshape(-1)
        self.K, cbsize, ogs, self.g = self.codebooks.shape
        assert ogs == 1 and cbsize == 2**nbits
        self.nbits = nbits
        self.in_groups, self.out_features, k2 = self.codes_alt.shape
        assert k2 == self.K
        self.in_features = self.in_groups * self.g
        nbytes = lib().aqlm_oracle_lut_bytes(self.in_features, self.K, nbits, self.g)
        self.lut = np.empty(nbytes // 4, dtype=np.float32)
        self.y = np.empty(self.out_features, dtype=np.float32)
        self.nthreads = nthreads

    def __call__(self, x):
        x = _f32(x).reshape(-1)
        assert x.size == self.in_features
        lib().aqlm_oracle_lut_gemv_f32(
            _p(x), _p(self.codebooks), _p(self.codes_alt), self.code_bytes, _p(self.scales), _p(self.y),
            self.in_features, self.out_features, self.K, self.nbits, self.g, _p(self.lut), self.nthreads,
        )
        return self.y


class DequantGemv:
    def __init__(self, codebooks, codes, scales, bias, nbits, nthreads=1):
        self.codebooks = _f32(codebooks)
        self.codes, self.code_bytes = _codes_unsigned_view(codes)
        self.scales = _f32(scales).reshape(-1)
        self.bias = _f32(bias)
        self.K, cbsize, ogs, self.g = self.codebooks.shape
        assert ogs == 1 and cbsize == 2**nbits
        self.nbits = nbits
        self.out_features, self.in_groups, k2 = self.codes.shape
        assert k2 == self.K
        self.in_features = self.in_groups * self.g
        self.y = np.empty(self.out_features, dtype=np.float32)
        self.nthreads = nthreads

    def __call__(self, x):
        x = _f32(x).reshape(-1)
        assert x.size == self.in_features
        lib().aqlm_oracle_dequant_gemv_f32(
            _p(x), _p(self.codebooks), _p(self.codes), self.code_bytes, _p(self.scales), _p(self.bias), _p(self.y),
            self.in_features, self.out_features, self.K, self.nbits, self.g, self.nthreads,
        )
        return self.y


def dequant_weight(codebooks, codes, scales, nbits):
    cb = _f32(codebooks)
    cu, cbytes = _codes_unsigned_view(codes)
    K, cbsize, ogs, g = cb.shape
    assert ogs == 1
    out_features, in_groups, _ = cu.shape
    W = np.empty((out_features, in_groups * g), dtype=np.float32)
    s = None if scales is None else _f32(scales).reshape(-1)
    lib().aqlm_oracle_dequant_weight_f32(_p(cb), _p(cu), cbytes, _p(s), _p(W), in_groups * g, out_features, K, nbits, g)
    return W
