#!/usr/bin/env python3
"""Disassembles one kernel of the in-tree library: python tools/disasm.py <kernel-name-substring> [grep-regex]."""
import os, re, struct, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_device_isa import _code_objects, OBJDUMP
from lattice_net_amd import _lib
want = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
blob = open(os.environ.get("LATTICE_NET_LIB", _lib.LIB_PATH), "rb").read()
for co in _code_objects(blob):
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(co)
    text = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True).stdout
    meta = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    os.unlink(f.name)
    for blk in re.split(r"\n(?=[0-9a-f]+ <)", text):
        head = blk.split("\n", 1)[0]
        if want in head:
            lines = blk.split("\n")
            print(head, f"({len(lines)} lines)")
            for ln in lines[1:]:
                if pat is None or pat.search(ln):
                    print(ln)
    for m in re.finditer(r"\.name:\s+(\S*%s\S*)" % re.escape(want), meta):
        i = m.start()
        seg = meta[max(0, i - 1500): i + 600]
        vg = re.findall(r"\.(vgpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|agpr_count):\s+(\d+)", seg)
        print(m.group(1), vg)
