"""ctypes binding of libaqlm_hip.so (include/aqlm_hip.h).

The library is built ahead of time by ``aqlm_amd/csrc/Makefile`` (``__graft_entry__.build()``) and lives
in-tree next to this file.  There is NO fallback: if the shared object is missing or a symbol is absent,
importing this module raises, and every op that needs it raises with it.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AQLM_AMD_HIP_LIB: another build of the same library (same-box A/B runs of two commits, tools/gpu/r3_ab*.sh); default: in-tree
LIB_PATH = os.environ.get("AQLM_AMD_HIP_LIB") or os.path.join(_HERE, "libaqlm_hip.so")
ABI_VERSION = 9

F16, BF16 = 0, 1
E_INVALID, E_UNSUPPORTED = -1, -2
MAX_GEMV_BATCH = 8
OP_GEMM_1X16_MFMA = 1
OP_GEMV_1X16_PACKED = 3
OP_GEMV_8X8_LUT = 4
OP_GEMM_KX8_MFMA = 6
OP_GEMV_1X16_G16_PACKED = 5

MAX_SEGMENTS = 4

_vp, _ci, _cl, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_size_t


class Segment(ctypes.Structure):
    """aqlm_hip_segment (include/aqlm_hip.h): one layer of a shared-input launch."""

    _fields_ = [("codes", _vp), ("codebook", _vp), ("scales", _vp), ("bias", _vp), ("y", _vp),
                ("y_row_stride", _cl), ("out_features", _ci), ("reserved", _ci)]


_segp = ctypes.POINTER(Segment)


PACKED_RELABELLED, PACKED_VARGEOM, PACKED_HAS_CODEBOOK = 1, 2, 4   # aqlm_hip_packed_desc.flags
PREPACK_NO_RELABEL, PREPACK_UNIFORM_ONLY = 1, 2                   # aqlm_hip_prepack_1x16_ex flags


class PackedDesc(ctypes.Structure):
    """aqlm_hip_packed_desc (include/aqlm_hip.h): what the kernels need to know about a prepacked 1x16 buffer."""

    _fields_ = [("magic", ctypes.c_uint32), ("version", ctypes.c_uint32), ("out_features", ctypes.c_int32),
                ("in_features", ctypes.c_int32), ("slices_log2", ctypes.c_int32), ("waves", ctypes.c_int32),
                ("steps", ctypes.c_int32), ("entry_bytes", ctypes.c_int32), ("used_bytes", ctypes.c_uint64),
                ("x_copies", ctypes.c_uint32), ("codebook_absmax", ctypes.c_float), ("flags", ctypes.c_uint32),
                ("rows_per_group", ctypes.c_int32), ("slice_groups", ctypes.c_uint8 * 32)]

    def as_ints(self):
        """The descriptor as a list of ints (what the registered torch op takes); the float travels as its bit pattern, the 32
        row-group counts as four 64-bit words."""
        import struct

        groups = bytes(self.slice_groups)
        return [int(self.magic), int(self.version), int(self.out_features), int(self.in_features),
                int(self.slices_log2), int(self.waves), int(self.steps), int(self.entry_bytes), int(self.used_bytes),
                int(self.x_copies), struct.unpack("<I", struct.pack("<f", float(self.codebook_absmax)))[0],
                int(self.flags), int(self.rows_per_group)] + [int.from_bytes(groups[8 * i:8 * i + 8], "little", signed=True) for i in range(4)]

    @classmethod
    def from_ints(cls, v):
        import struct

        absmax = struct.unpack("<f", struct.pack("<I", int(v[10]) & 0xFFFFFFFF))[0] if len(v) > 10 else 0.0
        groups = b"".join(int(x).to_bytes(8, "little", signed=True) for x in v[13:17]) if len(v) >= 17 else bytes(32)
        return cls(*[int(x) for x in v[:10]], absmax, int(v[11]) if len(v) > 11 else 0, int(v[12]) if len(v) > 12 else 0,
                   (ctypes.c_uint8 * 32)(*groups))

    @property
    def relabelled(self) -> bool:
        return bool(self.flags & PACKED_RELABELLED)

    @property
    def variable_geometry(self) -> bool:
        return bool(self.flags & PACKED_VARGEOM)


_descp = ctypes.POINTER(PackedDesc)
_descpp = ctypes.POINTER(_descp)


class Xgmi(ctypes.Structure):
    """aqlm_hip_xgmi (include/aqlm_hip.h): one rank's view of the one-shot all-reduce state."""

    _fields_ = [("peer_pub", _vp), ("peer_flag", _vp), ("epoch", _vp), ("status", _vp), ("rank", _ci), ("world", _ci),
                ("max_elems", _ci), ("spin_limit", ctypes.c_uint32)]


_xgp = ctypes.POINTER(Xgmi)

# name -> (restype, argtypes); mirrors include/aqlm_hip.h one to one
SIGNATURES = {
    "aqlm_hip_abi_version": (_ci, []),
    "aqlm_hip_last_error": (ctypes.c_char_p, []),
    "aqlm_hip_gemv_1x16": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _cl, _cl, _ci, _vp]),
    "aqlm_hip_gemv_1x16_multi": (_ci, [_segp, _ci, _vp, _ci, _ci, _ci, _cl, _ci, _vp]),
    "aqlm_hip_gemv_1x16_packed_multi": (_ci, [_segp, _descpp, _ci, _vp, _ci, _ci, _cl, _ci, _vp, _sz, _vp]),
    "aqlm_hip_gemv_1x16_packed_multi_cells": (_ci, [_segp, _descpp, _ci, _vp, _ci, _ci, _cl, _ci, _vp, _sz, _vp]),
    "aqlm_hip_gemv_kx8_multi": (_ci, [_segp, _ci, _vp, _ci, _ci, _ci, _ci, _cl, _ci, _vp]),
    "aqlm_hip_gemv_kx8": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _cl, _cl, _ci, _vp]),
    "aqlm_hip_prepack_1x16_bytes": (_sz, [_ci, _ci, _ci]),
    "aqlm_hip_prepack_1x16": (_ci, [_vp, _ci, _ci, _ci, _vp, _sz, _descp, _vp]),
    "aqlm_hip_prepack_1x16_ex": (_ci, [_vp, _ci, _ci, _ci, _vp, _sz, _descp, _ci, _vp]),
    "aqlm_hip_packed_set_codebook": (_ci, [_descp, _vp, _vp, _vp]),
    "aqlm_hip_packed_plan_relabel": (_ci, [_vp, _ci, _vp]),
    "aqlm_hip_packed_plan_relabel_ex": (_ci, [_vp, _ci, _ci, _vp]),
    "aqlm_hip_packed_plan_geometry": (_ci, [_vp, _ci, _ci, _ci, _vp]),
    "aqlm_hip_packed_desc_read": (_ci, [_vp, _sz, _descp]),
    "aqlm_hip_unpack_1x16": (_ci, [_descp, _vp, _vp, _vp]),
    "aqlm_hip_gemv_1x16_packed": (_ci, [_descp, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _cl, _cl, _ci, _vp, _sz, _vp]),
    "aqlm_hip_gemv_1x16_packed_cells": (_ci, [_descp, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _cl, _cl, _ci, _vp, _sz, _vp]),
    "aqlm_hip_gemv_1x16_packed_chain": (_ci, [_descp, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _cl, _cl, _ci, _vp, _sz, _descp, _vp, _vp, _vp]),
    "aqlm_hip_gemv_1x16_packed_partials": (_ci, [_descp, _vp, _vp, _vp, _ci, _cl, _ci, _vp, _sz, _vp]),
    "aqlm_hip_gemv_1x16_packed_publish": (_ci, [_descp, _vp, _vp, _vp, _ci, _cl, _ci, _xgp, _vp, _vp, _vp]),
    "aqlm_hip_xgmi_state_bytes": (_sz, [_ci]),
    "aqlm_hip_xgmi_finalize": (_ci, [_xgp, _vp, _vp, _vp, _vp, _ci, _ci, _cl, _ci, _vp]),
    "aqlm_hip_gemv_8x8_lut": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp, _sz, _vp]),
    "aqlm_hip_gemv_8x8_lut_multi": (_ci, [_segp, _ci, _vp, _ci, _ci, _ci, _vp, _sz, _vp]),
    "aqlm_hip_gemv_8x8_lut_fused": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp, _sz, _vp]),
    "aqlm_hip_gemv_8x8_lut_multi_fused": (_ci, [_segp, _ci, _vp, _ci, _ci, _ci, _vp, _sz, _vp]),
    "aqlm_hip_8x8_planar_bytes": (_sz, [_ci, _ci, _ci]),
    "aqlm_hip_8x8_planar_pack": (_ci, [_vp, _ci, _ci, _ci, _vp, _sz, _vp]),
    "aqlm_hip_8x8_planar_unpack": (_ci, [_vp, _ci, _ci, _ci, _vp, _vp]),
    "aqlm_hip_gemv_8x8_lut_planar": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, ctypes.c_float, _vp, _sz, _ci, _vp]),
    "aqlm_hip_gemv_8x8_lut_batch": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _cl, _cl, _ci, _ci, ctypes.c_float, _vp, _sz, _vp]),
    "aqlm_hip_gemv_8x8_lut_planar_multi": (_ci, [_segp, ctypes.POINTER(ctypes.c_float), _ci, _vp, _ci, _ci, _ci, _vp, _sz, _ci, _vp]),
    "aqlm_hip_gemv_generic": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _cl, _cl, _ci, _vp]),
    "aqlm_hip_dequant_1x16": (_ci, [_vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp]),
    "aqlm_hip_dequant_kx8": (_ci, [_vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _vp]),
    "aqlm_hip_dequant_generic": (_ci, [_vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _vp]),
    "aqlm_hip_gemm_1x16_mfma": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _cl, _cl, _ci, _vp, _sz, _vp]),
    "aqlm_hip_gemm_1x16_scan": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _cl, _cl, _ci, _vp, _sz, _vp]),
    "aqlm_hip_gemm_1x16_scan_workspace_bytes": (_sz, [_ci, _ci, _ci]),
    "aqlm_hip_gemm_kx8_mfma": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _cl, _cl, _ci, _vp]),
    "aqlm_hip_gemm_kx8_mfma_ws": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _cl, _cl, _ci, _vp, _sz, _vp]),
    "aqlm_hip_gemm_8x8_mfma": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _cl, _cl, _ci, _vp]),
    "aqlm_hip_workspace_bytes": (_sz, [_ci, _ci, _ci, _ci]),
    "aqlm_hip_checksum": (_ci, [_vp, _sz, _vp, _vp]),
    "aqlm_hip_set_tuning": (_ci, [ctypes.c_char_p, _ci]),
    "aqlm_hip_get_tuning": (_ci, [ctypes.c_char_p, ctypes.POINTER(_ci)]),
}


class AqlmHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the MI355X HIP library has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C aqlm_amd/csrc`. "
            "aqlm_amd has no fallback path."
        )
    # torch bundles its own libamdhip64.so (same SONAME libamdhip64.so.7); importing torch first makes the
    # dynamic loader resolve our NEEDED entry to the runtime torch already loaded, so stream handles and device
    # pointers are shared with torch.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - the C ABI is usable without torch
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    v = lib.aqlm_hip_abi_version()
    if v != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {v}, expected {ABI_VERSION}; rebuild it")
    return lib


lib = _load()


def last_error() -> str:
    return (lib.aqlm_hip_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    """Map a C-ABI return code to the reference's exception types (cuda_kernel.cpp:9-25, 136-145)."""
    if rc == 0:
        return
    msg = last_error() or what
    if rc == E_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == E_INVALID:
        raise ValueError(msg)
    raise AqlmHipError(f"HIP error {rc}: {msg}")


def set_tuning(key: str, value: int) -> None:
    check(lib.aqlm_hip_set_tuning(key.encode(), int(value)), "set_tuning")


def get_tuning(key: str) -> int:
    v = _ci(0)
    check(lib.aqlm_hip_get_tuning(key.encode(), ctypes.byref(v)), "get_tuning")
    return v.value
