"""The shipped gfx950 code objects must not contain packed fp32 VALU instructions (CPU test: disassembles the in-tree .so).

MI355X measurement behind the rule (tools/probes/pk_fma_vs_mfma_probe.cpp): v_pk_fma_f32 / v_pk_mul_f32 whose LOW result
takes a source from the HIGH half of a register pair (op_sel:[0,1,0], the compiler's way to broadcast a weight) return a wrong low
result while a wave of another kernel on the same SIMD executes v_mfma_f32_16x16x32_{bf16,f16}.  lattice_net_amd/build_ext.py
compiles with -target-feature -packed-fp32-ops; this test keeps a future flag change from silently re-enabling them."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
EM_AMDGPU = 224


def _code_objects(blob: bytes):
    pos = 0
    while True:
        i = blob.find(b"\x7fELF", pos)
        if i < 0:
            return
        pos = i + 4
        if i == 0 or blob[i + 4] != 2 or struct.unpack_from("<H", blob, i + 18)[0] != EM_AMDGPU:
            continue
        shoff = struct.unpack_from("<Q", blob, i + 40)[0]
        shentsize, shnum = struct.unpack_from("<HH", blob, i + 58)
        yield blob[i:i + shoff + shentsize * shnum]


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_no_packed_fp32_instructions_in_the_library():
    from lattice_net_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the library first (__graft_entry__.build())"
    blob = open(_lib.LIB_PATH, "rb").read()
    objects = list(_code_objects(blob))
    assert len(objects) >= 8  # one per .hip source
    instructions = 0
    for co in objects:
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
        try:
            text = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(f.name)
        instructions += text.count("\n")
        found = sorted(set(re.findall(r"v_pk_(?:fma|mul|add)_f32|v_pk_mov_b32", text)))
        assert not found, f"packed fp32 instructions in the device code: {found}"
    assert instructions > 100000  # the disassembly really happened


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"), reason="llvm-readelf of the ROCm toolchain not found")
def test_no_kernel_of_the_library_spills_registers():
    """Code-object metadata of every kernel: .vgpr_spill_count is 0 (vector registers spilled to scratch memory: such a hot loop runs
    2-3x slower, and the spill comes and goes with unrelated edits — round 4 shipped k_conv_mfma_b3<256, 2, *, 1> with 5-6 spilled
    registers).  Scalar registers parked in vector-register lanes (.sgpr_spill_count, k_bucket_rows: 26) cost no memory traffic and
    are not counted."""
    from lattice_net_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    kernels, spilled = 0, []
    for co in _code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
        try:
            meta = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(f.name)
        # one YAML map per kernel under amdhsa.kernels; the keys of a map are sorted, so .name precedes the spill counts
        for m in re.finditer(r"\.name:\s+(\S+)(.*?)\.vgpr_spill_count:\s+(\d+)", meta, re.S):
            name, between, vspill = m.group(1), m.group(2), int(m.group(3))
            if ".name:" in between:
                continue  # (an argument's .name, not the kernel's)
            sspill = re.search(r"\.sgpr_spill_count:\s+(\d+)", between)
            kernels += 1
            if vspill:
                spilled.append((name, vspill, int(sspill.group(1)) if sspill else 0))
    assert kernels > 300, kernels
    assert not spilled, spilled
