"""Loader of ``_aqlm_front.so``: the compiled host glue of the decode path (aqlm_amd/csrc_front/front.cpp).

The reference's ops are C++ (inference_lib/src/aqlm/inference_kernels/cuda_kernel.cpp:148-182, pybind :686-699); here the
general path is Python + ctypes (hip_kernel.py) and this extension is the fast lane of a ``QuantizedLinear`` decode call.
Optional: without it every call takes the Python path (same kernels, ~12 us more host time per call).
"""
from __future__ import annotations

import importlib.util
import os
import warnings

from . import _native  # noqa: F401  (loads libaqlm_hip.so first: the extension links against the same file)

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_aqlm_front.so")
KIND_PACKED_1X16, KIND_GEMV_1X16, KIND_GEMV_KX8, KIND_LUT_PLANAR_8X8 = 0, 1, 2, 3

ext = None
# AQLM_AMD_HIP_LIB points the ctypes side at ANOTHER build of the library; the extension is linked against the in-tree one
# (-laqlm_hip, rpath $ORIGIN), so with that variable set two builds would be live at once (fast-lane calls on one, tuning knobs and
# Python-path calls on the other).  The extension is therefore switched off whenever the variable is set.
if os.path.exists(_PATH) and os.environ.get("AQLM_AMD_NO_FRONT", "0") != "1" and not os.environ.get("AQLM_AMD_HIP_LIB"):
    try:
        _spec = importlib.util.spec_from_file_location("aqlm_amd._aqlm_front", _PATH)
        ext = importlib.util.module_from_spec(_spec)
        _spec.loader.exec_module(ext)
        _loaded = int(_native.lib.aqlm_hip_abi_version())  # the library that was actually loaded, not a constant
        if ext.ABI_VERSION != _loaded:
            raise ImportError(f"built against ABI {ext.ABI_VERSION}, the loaded library is {_loaded}; rebuild it (make -C aqlm_amd/csrc_front)")
    except Exception as e:  # a stale or foreign build must not take the package down: the Python path serves every call
        warnings.warn(f"aqlm_amd: {_PATH} could not be loaded ({e}); decode calls take the Python path")
        ext = None


def available() -> bool:
    return ext is not None
