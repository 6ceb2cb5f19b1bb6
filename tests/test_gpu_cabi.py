"""The C ABI driven from plain C++ (tests/cabi/cabi_smoke.cpp: HIP runtime only, no torch, no Python in the data path) —
the boundary a compiled host such as the reference's src/Lattice.cu would bind."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_from_plain_cpp(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    pkg = os.path.join(ROOT, "lattice_net_amd")
    exe = str(tmp_path / "cabi_smoke")
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cabi", "cabi_smoke.cpp"), "-I", os.path.join(ROOT, "include"),
           "-L", pkg, "-llatticenet_hip", f"-Wl,-rpath,{pkg}", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "CABI OK" in r.stdout
