"""Builds the gfx950 C-ABI library in-tree: lattice_net_amd/liblatticenet_hip.so.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with gpurun snapshots.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(PKG, "liblatticenet_hip.so")
OBJ_DIR = os.path.join(ROOT, "build", "obj")
SOURCES = ["ln_table.hip", "ln_rows.hip", "ln_conv.hip", "ln_csr.hip", "ln_norm.hip", "ln_conv_f16.hip", "ln_mlp.hip", "ln_centre.hip"]
import glob
import hashlib

ABI_HEADER = os.path.join(ROOT, "include", "latticenet_hip.h")
# every header under csrc/ is a dependency of every object (editing ln_neighbours.h must rebuild ln_table.o and ln_csr.o)
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [ABI_HEADER]


def abi_hash() -> str:
    """Short hash of the C-ABI header.  Compiled into the library (ln_abi_hash) and compared by _lib.load(): a stale
    git-ignored .so whose argument lists no longer match the ctypes SIGNATURES raises instead of corrupting memory."""
    with open(ABI_HEADER, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]
# -ffp-contract=off: lattice keys and barycentric weights must be bit-identical to the oracle
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
         "-I" + CSRC]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    flags = FLAGS + ['-DLN_ABI_HASH="%s"' % abi_hash()]
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([hipcc] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(OUT, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
