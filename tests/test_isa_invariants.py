"""Static checks on the gfx950 ISA of the prepacked 1x16 kernels (no GPU needed: hipcc cross-compiles).

The kernels wait for their LDS fill with a hand-written ``s_waitcnt vmcnt(N)``: N must equal the number of loads every
wave issues BEHIND its last LDS-DMA instruction (the entry ring + the epilogue's scale / bias), or the barrier lets waves
read a slice that is still in flight.  hipcc may drop or move loads, so the count is verified on the generated code of
every template instance (both builds: 8- and 16-element codebook vectors)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=fast", "-mllvm",
         "-amdgpu-kernarg-preload-count=14", "--cuda-device-only", "-S"]


def check_asm(text):
    """-> (kernels checked, [(kernel, loads behind the last LDS-DMA, count of the first wait)] that disagree)"""
    name, loads, seen_dma, done, n, bad = None, 0, False, False, 0, []
    for line in text.split("\n"):
        m = re.match(r"^(_ZN4aqlm\w+gemv_1x16_packed(?:_multi|_vg)?_kernel\w+):", line)
        if m:
            name, loads, seen_dma, done = m.group(1), 0, False, False
            continue
        if name is None or done:
            if "s_endpgm" in line:
                name = None
            continue
        if "s_endpgm" in line:
            name = None
        elif "global_load_lds" in line:
            loads, seen_dma = 0, True
        elif seen_dma and re.search(r"\b(global_load|buffer_load)_", line):
            loads += 1
        else:
            w = re.search(r"s_waitcnt vmcnt\((\d+)\)", line)
            if w and seen_dma:
                n += 1
                done = True
                if int(w.group(1)) != loads:
                    bad.append((name, loads, int(w.group(1))))
    return n, bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", ["gemv_packed.hip", "gemv_packed_g16.hip"])
def test_fill_wait_counts_the_loads_behind_the_last_lds_dma(src, tmp_path):
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC] + FLAGS + [os.path.join(ROOT, "aqlm_amd", "csrc", src), "-o", str(out)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    n, bad = check_asm(out.read_text())
    assert n >= 60, f"{src}: only {n} packed kernels found in the ISA"
    assert not bad, f"{src}: fill wait does not match the loads issued behind the LDS-DMA: {bad[:5]}"


def dma_loops_with_full_drains(text, kernel_pattern):
    """-> (loops found, [(kernel, label)] of LDS-DMA loops that contain an ``s_waitcnt vmcnt(0)``).

    The producer waves of the MFMA kernels (gemm_mfma.hip) keep several steps of LDS-DMA in flight and wait with a COUNTED
    ``s_waitcnt vmcnt(N)``, N > 0, once per step.  A ``vmcnt(0)`` inside such a loop (hipcc adds one when it believes an LDS read
    may alias the DMA's destination) would drain the ring every step: correct results, a fraction of the speed -- exactly the kind
    of regression no parity test can see.  Loops are taken from the compiler's own annotation ("Loop Header" on the label line)
    up to the branch back to that label."""
    loops, bad = 0, []
    name, header, body = None, None, []
    for line in text.split("\n"):
        m = re.match(r"^(_ZN4aqlm\w+):", line)
        if m:
            name = m.group(1) if re.search(kernel_pattern, m.group(1)) else None
            header, body = None, []
            continue
        if name is None:
            continue
        lm = re.match(r"^(\.LBB\d+_\d+):.*Loop Header", line)
        if lm:
            header, body = lm.group(1), []
            continue
        if header is None:
            continue
        body.append(line)
        if re.search(r"s_cbranch_\w+\s+" + re.escape(header) + r"\b", line):
            if any("global_load_lds" in x for x in body) and any("s_barrier" in x for x in body):
                loops += 1
                if any(re.search(r"s_waitcnt[^\n]*vmcnt\(0\)", x) for x in body):
                    bad.append((name, header))
            header, body = None, []
    return loops, bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_mfma_kernels_dma_loops_never_drain_their_rings(tmp_path):
    out = tmp_path / "gemm_mfma.s"
    subprocess.run([HIPCC] + FLAGS + [os.path.join(ROOT, "aqlm_amd", "csrc", "gemm_mfma.hip"), "-o", str(out)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    text = out.read_text()
    for pattern, least in (("gemm_1x16_rows16_kernel", 16), ("gemm_kx8_rows16_kernel", 16), ("gemm_1x16_glds_kernel", 8)):
        loops, bad = dma_loops_with_full_drains(text, pattern)
        assert loops >= least, f"{pattern}: only {loops} LDS-DMA loops found in the ISA"
        assert not bad, f"{pattern}: vmcnt(0) inside an LDS-DMA loop: {bad[:5]}"
