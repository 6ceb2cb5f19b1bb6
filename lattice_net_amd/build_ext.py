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
SOURCES = ["ln_table.hip", "ln_rows.hip", "ln_conv.hip", "ln_csr.hip", "ln_norm.hip", "ln_conv_f16.hip", "ln_mlp.hip", "ln_centre.hip", "ln_classify.hip", "ln_glue.hip"]
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
# -target-feature -packed-fp32-ops: no v_pk_{fma,mul,add}_f32 / v_pk_mov_b32 in the device code.  Measured on MI355X
# (tools/probes/pk_fma_vs_mfma_probe.cpp): a packed fp32 instruction whose LOW result takes a source from the HIGH
# half of a register pair (op_sel:[0,1,0], what the compiler emits to broadcast a weight) returns a wrong low result while a wave of
# ANOTHER kernel on the same SIMD executes v_mfma_f32_16x16x32_{bf16,f16} — e.g. a segment reduce on one stream beside a bf16x3
# convolution on another.  The kernels here are bound by memory and latency, not by the fp32 vector rate, so nothing is lost.
# (The host half of the compilation prints "not a recognized feature for this target (ignoring feature)".)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Xclang", "-target-feature", "-Xclang",
         "-packed-fp32-ops", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


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


def build(force: bool = False, verbose: bool = False, variant: str = "", extra_flags=()) -> str:
    """`variant` / `extra_flags` (tools only): an experimental build liblatticenet_hip_<variant>.so with extra -D flags,
    loaded instead of the product library when LATTICE_NET_LIB points at it."""
    obj_dir = OBJ_DIR + ("_" + variant if variant else "")
    out = OUT if not variant else OUT.replace(".so", "_" + variant + ".so")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    flags = FLAGS + ['-DLN_ABI_HASH="%s"' % abi_hash()] + list(extra_flags)
    stamp = os.path.join(obj_dir, "flags.txt")  # objects compiled with other flags are stale whatever their age
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s.replace(".hip", ".o"))
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
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    if force or jobs or _stale(out, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    # python build_ext.py [--force] [--variant NAME -DX=1 -DY=2 ...]
    argv = sys.argv[1:]
    variant = argv[argv.index("--variant") + 1] if "--variant" in argv else ""
    print(build(force="--force" in argv, verbose=True, variant=variant, extra_flags=[a for a in argv if a.startswith("-D")]))
