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
    assert native.lib.aqlm_hip_abi_version() == native.ABI_VERSION == 9
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
    # prepacked entry points: descriptors are validated before anything is launched
    assert ctypes.sizeof(native.PackedDesc) == 88
    assert L.aqlm_hip_prepack_1x16_bytes(4096, 4096, 8) > 2 * 4096 * 512 * 2
    assert L.aqlm_hip_prepack_1x16_bytes(4096, 4096, 16) > 2 * 4096 * 256 * 2   # 16-element vectors: the second instantiation
    assert L.aqlm_hip_prepack_1x16_bytes(4096, 4096, 32) == 0          # other group sizes are not packable
    assert L.aqlm_hip_prepack_1x16_bytes(64, 8 * 4095, 8) == 0         # in/8 > 4094: j does not fit 12 bits
    bad = native.PackedDesc()
    descs = (native._descp * 2)(ctypes.pointer(bad), ctypes.pointer(bad))
    rc = L.aqlm_hip_gemv_1x16_packed_multi(segs, descs, 2, p, 512, 1, 512, native.F16, p, 1 << 20, None)
    assert rc == native.E_INVALID and "descriptor" in native.last_error()
    rc = L.aqlm_hip_gemv_1x16_packed(ctypes.byref(bad), p, p, p, None, p, p, 1, 512, 64, native.F16, p, 1 << 20, None)
    assert rc == native.E_INVALID and "descriptor" in native.last_error()
    # 64 rows -> 4 per row group; winfo (sized for 16 waves) + row starts (256 x 5 u32) + the accumulator cells of the
    # fused finalize (8 x 64 u64) end at 74 KiB, then 256 x 4 x 1 KiB of entries
    uniform = (ctypes.c_uint8 * 32)(*([16] * 16))
    good = native.PackedDesc(0x37505141, 7, 64, 512, 4, 4, 1, 4, 1024 * (74 + 1024), 4, 1.5, 0, 4, uniform)   # ... 4 copies of x, |codebook| <= 1.5
    back = native.PackedDesc.from_ints(good.as_ints())
    assert bytes(back) == bytes(good)
    rc = L.aqlm_hip_gemv_1x16_packed(ctypes.byref(good), p, p, p, None, p, p, 9, 512, 64, native.F16, p, 1 << 20, None)
    assert rc == native.E_INVALID and "batch" in native.last_error()
    rc = L.aqlm_hip_gemv_1x16_packed(ctypes.byref(good), p, p, p, None, p, p, 1, 512, 64, 7, p, 1 << 20, None)
    assert rc == native.E_UNSUPPORTED
    hdr = ctypes.create_string_buffer(bytes(good), 128)
    out = native.PackedDesc()
    assert L.aqlm_hip_packed_desc_read(ctypes.addressof(hdr), 128, ctypes.byref(out)) == 0 and bytes(out) == bytes(good)
    assert L.aqlm_hip_packed_desc_read(ctypes.addressof(hdr), 64, ctypes.byref(out)) == native.E_INVALID   # shorter than the descriptor
    assert L.aqlm_hip_packed_desc_read(ctypes.addressof(buf), 128, ctypes.byref(out)) == native.E_INVALID
    # format v7: a relabelled buffer refuses to run before its codebook image has been written
    relab = native.PackedDesc.from_ints(good.as_ints())
    relab.flags = native.PACKED_RELABELLED
    relab.used_bytes = 1024 * (74 + 1024) + 65536 * 2 + 65536 * 16
    rc = L.aqlm_hip_gemv_1x16_packed(ctypes.byref(relab), p, p, p, None, p, p, 1, 512, 64, native.F16, p, 1 << 20, None)
    assert rc == native.E_INVALID and "aqlm_hip_packed_set_codebook" in native.last_error()
    # ... and a descriptor whose row groups do not add up to the 256 workgroups is not a descriptor
    odd = native.PackedDesc.from_ints(good.as_ints())
    odd.slice_groups[3] = 17
    rc = L.aqlm_hip_gemv_1x16_packed(ctypes.byref(odd), p, p, p, None, p, p, 1, 512, 64, native.F16, p, 1 << 20, None)
    assert rc == native.E_INVALID and "descriptor" in native.last_error()


def test_workspace_bytes(native):
    L = native.lib
    n = L.aqlm_hip_workspace_bytes(native.OP_GEMM_1X16_MFMA, 128, 4096, 4096)
    assert n > 0 and n % (4096 * 128 * 4) == 0
    assert L.aqlm_hip_workspace_bytes(99, 128, 4096, 4096) == 0
    assert L.aqlm_hip_workspace_bytes(native.OP_GEMM_1X16_MFMA, 7, 4096, 4096) < n
    assert L.aqlm_hip_workspace_bytes(native.OP_GEMV_1X16_PACKED, 1, 4096, 4096) == 16 * 4096 * 4
    assert L.aqlm_hip_workspace_bytes(native.OP_GEMV_1X16_PACKED, 4, 11008, 4096) == 16 * 4 * 11008 * 4
    assert L.aqlm_hip_workspace_bytes(native.OP_GEMV_1X16_G16_PACKED, 4, 11008, 4096) == 32 * 4 * 11008 * 4


def test_tuning_knobs(native):
    assert native.get_tuning("gemv_rows_per_wave") == 0
    native.set_tuning("gemv_rows_per_wave", 3)
    assert native.get_tuning("gemv_rows_per_wave") == 3
    native.set_tuning("gemv_rows_per_wave", 0)
    with pytest.raises(ValueError):
        native.set_tuning("no_such_knob", 1)


def test_scan_entry_validation_and_plan_without_gpu(native):
    """aqlm_hip_gemm_1x16_scan (round 6): argument checks and the plan behind the workspace size -- 8 codebook slices x the K chunks
    the x fragments' registers ask for (<= 3 units of 256 features per wave at <= 16 rows, <= 2 at 17+ rows)."""
    L = native.lib
    buf = ctypes.create_string_buffer(4096)
    p = (ctypes.addressof(buf) + 15) & ~15
    ws = L.aqlm_hip_gemm_1x16_scan_workspace_bytes
    assert ws(8, 4096, 4096) == 8 * 1 * 8 * 4096 * 4          # one chunk: 16 units over 8 waves x 2
    assert ws(16, 11008, 4096) == 8 * 1 * 16 * 11008 * 4
    assert ws(8, 4096, 11008) == 8 * 3 * 8 * 4096 * 4         # 43 units: 3 chunks of 15 / 14 / 14 units (8 waves x 2 units each)
    assert ws(32, 4096, 11008) == 8 * 3 * 32 * 4096 * 4       # the K chunking does not depend on the row count (batch invariance)
    assert ws(8, 4096, 14336) == 8 * 4 * 8 * 4096 * 4
    assert ws(8, 4096, 4096 + 64) == 0 and ws(0, 4096, 4096) == 0   # in_features % 256 != 0: no plan
    assert L.aqlm_hip_workspace_bytes(native.OP_GEMM_1X16_MFMA, 16, 11008, 4096) >= ws(16, 11008, 4096)
    rc = L.aqlm_hip_gemm_1x16_scan(None, p, p, None, p, p, 8, 4096, 4096, 4096, 4096, native.F16, p, 1 << 30, None)
    assert rc == native.E_INVALID and "null pointer" in native.last_error()
    rc = L.aqlm_hip_gemm_1x16_scan(p, p, p, None, p, p, 8, 4096, 4160, 4160, 4096, native.F16, p, 1 << 30, None)
    assert rc == native.E_UNSUPPORTED and "256" in native.last_error()
    rc = L.aqlm_hip_gemm_1x16_scan(p, p, p, None, p, p, 8, 4096, 4096, 4096, 4096, 7, p, 1 << 30, None)
    assert rc == native.E_UNSUPPORTED and "float16 and bfloat16" in native.last_error()
    rc = L.aqlm_hip_gemm_1x16_scan(p, p, p, None, p, p, 8, 4096, 4096, 4096, 4096, native.F16, p, 64, None)
    assert rc == native.E_INVALID and "workspace" in native.last_error()
    assert native.get_tuning("scan_max_rows") == 0              # not on a default route (measured slower: gemm_1x16_scan.hip)
    native.set_tuning("scan_max_rows", 64)
    assert native.get_tuning("scan_max_rows") == 64
    native.set_tuning("scan_max_rows", 0)
