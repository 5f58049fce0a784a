"""CPU-only checks of the C-ABI boundary: the library loads (no GPU needed), exports every symbol that
include/aqlm_hip.h declares, and rejects bad arguments before touching the device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    from aqlm_amd import _native

    return _native


def header_symbols():
    text = open(os.path.join(ROOT, "include", "aqlm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aqlm_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(native):
    syms = header_symbols()
    assert len(syms) >= 11
    assert sorted(native.SIGNATURES) == syms
    raw = ctypes.CDLL(native.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"libaqlm_hip.so does not export {s}"


def test_abi_version_and_error_string(native):
    assert native.lib.aqlm_hip_abi_version() == native.ABI_VERSION == 1
    assert isinstance(native.last_error(), str)


def test_argument_validation_without_gpu(native):
    L = native.lib
    # null pointers
    rc = L.aqlm_hip_gemv_1x16(None, None, None, None, None, None, 64, 512, 8, 1, 512, 64, native.F16, None)
    assert rc == native.E_INVALID and "null pointer" in native.last_error()
    buf = ctypes.create_string_buffer(4096)
    p = ctypes.addressof(buf)
    # sizes
    rc = L.aqlm_hip_gemv_1x16(p, p, p, None, p, p, 64, 500, 8, 1, 512, 64, native.F16, None)
    assert rc == native.E_INVALID and "multiple of in_group_size" in native.last_error()
    # batch too large for the gemv entry point
    rc = L.aqlm_hip_gemv_1x16(p, p, p, None, p, p, 64, 512, 8, 9, 512, 64, native.F16, None)
    assert rc == native.E_UNSUPPORTED
    # dtype: mirrors check_use_bfloat16 (reference cuda_kernel.cpp:9-25)
    rc = L.aqlm_hip_gemv_kx8(p, p, p, None, p, p, 64, 512, 2, 8, 1, 512, 64, 7, None)
    assert rc == native.E_UNSUPPORTED and "float16 and bfloat16" in native.last_error()
    # 1x16 group size: mirrors cuda_kernel.cpp:136-145
    rc = L.aqlm_hip_gemv_1x16(p, p, p, None, p, p, 64, 512, 4, 1, 512, 64, native.F16, None)
    assert rc == native.E_UNSUPPORTED and "8 or 16" in native.last_error()
    rc = L.aqlm_hip_dequant_1x16(p, p, None, p, 64, 512, 32, native.F16, None)
    assert rc == native.E_UNSUPPORTED
    rc = L.aqlm_hip_dequant_kx8(None, p, None, p, 64, 512, 2, 8, native.F16, None)
    assert rc == native.E_INVALID
    rc = L.aqlm_hip_gemm_1x16_mfma(p, p, p, None, p, p, 128, 4096, 4096, 8, 4096, 4096, native.F16, None, 0, None)
    assert rc == native.E_INVALID and "workspace" in native.last_error()
    with pytest.raises(NotImplementedError):
        native.check(native.E_UNSUPPORTED)
    with pytest.raises(ValueError):
        native.check(native.E_INVALID)


def test_segment_struct_and_multi_validation(native):
    # aqlm_hip_segment: 5 pointers + long + 2 ints, no padding surprises
    assert ctypes.sizeof(native.Segment) == 56
    assert native.Segment.y_row_stride.offset == 40 and native.Segment.out_features.offset == 48
    L = native.lib
    buf = ctypes.create_string_buffer(4096)
    p = ctypes.addressof(buf)
    segs = (native.Segment * 2)()
    rc = L.aqlm_hip_gemv_1x16_multi(segs, 0, p, 512, 8, 1, 512, native.F16, None)
    assert rc == native.E_INVALID and "segments" in native.last_error()
    rc = L.aqlm_hip_gemv_1x16_multi(segs, native.MAX_SEGMENTS + 1, p, 512, 8, 1, 512, native.F16, None)
    assert rc == native.E_INVALID
    rc = L.aqlm_hip_gemv_1x16_multi(segs, 2, p, 512, 8, 1, 512, native.F16, None)  # null segment pointers
    assert rc == native.E_INVALID and "null pointer" in native.last_error()
    for s in segs:
        s.codes = s.codebook = s.scales = s.y = p
        s.out_features, s.y_row_stride = 64, 64
    rc = L.aqlm_hip_gemv_1x16_multi(segs, 2, p, 512, 4, 1, 512, native.F16, None)
    assert rc == native.E_UNSUPPORTED and "8 or 16" in native.last_error()
    rc = L.aqlm_hip_gemv_1x16_multi(segs, 2, p, 512, 8, 1, 512, 5, None)
    assert rc == native.E_UNSUPPORTED and "float16 and bfloat16" in native.last_error()
    rc = L.aqlm_hip_gemv_1x16_packed_multi(segs, 2, p, 512, 16, native.F16, p, 1 << 20, None)  # g16 not packable
    assert rc == native.E_UNSUPPORTED
    rc = L.aqlm_hip_gemv_1x16_packed_multi(segs, 2, p, 512, 8, native.F16, None, 0, None)
    assert rc == native.E_INVALID and "workspace" in native.last_error()


def test_workspace_bytes(native):
    L = native.lib
    n = L.aqlm_hip_workspace_bytes(native.OP_GEMM_1X16_MFMA, 128, 4096, 4096)
    assert n > 0 and n % (4096 * 128 * 4) == 0
    assert L.aqlm_hip_workspace_bytes(99, 128, 4096, 4096) == 0
    assert L.aqlm_hip_workspace_bytes(native.OP_GEMM_1X16_MFMA, 7, 4096, 4096) < n


def test_tuning_knobs(native):
    assert native.get_tuning("gemv_rows_per_wave") == 0
    native.set_tuning("gemv_rows_per_wave", 3)
    assert native.get_tuning("gemv_rows_per_wave") == 3
    native.set_tuning("gemv_rows_per_wave", 0)
    with pytest.raises(ValueError):
        native.set_tuning("no_such_knob", 1)
