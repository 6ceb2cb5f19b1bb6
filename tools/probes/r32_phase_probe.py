#!/usr/bin/env python3
"""Where a wave of k_conv_rows32_b3 spends its cycles, and where the workgroups run (needs the -DLN_STAMPS build):

    python lattice_net_amd/build_ext.py --variant stamps -DLN_STAMPS=1
    LATTICE_NET_LIB=lattice_net_amd/liblatticenet_hip_stamps.so python tools/probes/r32_phase_probe.py [VxF ...]
"""
import ctypes as C
import os
import sys
from collections import Counter

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lattice_net_amd as L  # noqa: E402
from lattice_net_amd import synthetic  # noqa: E402

lib = C.CDLL(L.LIB_PATH)
lib.ln_debug_set_stamps_conv.argtypes = [C.c_void_p]
dev = torch.device("cuda", 0)
pos = torch.from_numpy(synthetic.lidar_cloud(120000, 0)).to(dev)
lat = L.Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev)
lat.begin_splat()
dl, _, _, _ = lat.distribute(pos, torch.zeros((120000, 1), device=dev))
m = dl.nr_lattice_vertices()
names = ["loop tail", "wait for the chunks (vmcnt)", "barrier", "LDS reads issued (fragments, next rows)", "held-back products issued", "requests (DMA, ids)", "second fragments + products issued", "next rows split"]
for shp in (sys.argv[1:] or ["128x128", "64x64"]):
    v, f = (int(x) for x in shp.split("x"))
    vals = torch.randn((m, v), device=dev)
    bank = torch.randn((9 * v, f), device=dev) * 0.05
    dl.set_values(vals)
    st = torch.zeros((4096 * 4, 12), dtype=torch.int64, device=dev)
    for _ in range(3):
        dl.convolve_im2row_standalone(bank, 1, dl, False)
    torch.cuda.synchronize()
    lib.ln_debug_set_stamps_conv(st.data_ptr())
    dl.convolve_im2row_standalone(bank, 1, dl, False)
    torch.cuda.synchronize()
    lib.ln_debug_set_stamps_conv(None)
    s = st.cpu().numpy()
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    start, end = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0
    hw, xcc = s[:, 2] & 0xFFFFFFFF, s[:, 2] >> 32
    wave_id = hw & 15
    cu, sh, se, simd = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 4) & 3
    where = Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_cu = Counter(where.values())
    ph = s[:, 3:11].astype(np.float64)
    iters = s[0, 11]
    print(f"V {v} -> F {f}: {len(s)} waves on {len(where)} CUs (waves per CU -> CUs: {dict(sorted(per_cu.items()))}); "
          f"{iters} chunks per wave")
    print(f"  wave start {start.min():.1f} .. {start.max():.1f} us, end {end.min():.1f} .. {end.max():.1f} us, wave lifetime mean {np.mean(end - start):.1f} us")
    tot = ph.sum(1)
    for k, nm in enumerate(names):
        print(f"  {nm:42s} {ph[:, k].mean() / iters:8.0f} cycles per chunk  ({100 * ph[:, k].sum() / tot.sum():5.1f} %)")
    print(f"  {'sum':42s} {tot.mean() / iters:8.0f} cycles per chunk; {tot.mean():.0f} per wave = {tot.mean() / np.mean(end - start) / 1e3:.2f} GHz")
    for nwg in sorted(set(where.values())):
        sel = np.array([where[k] == nwg for k in zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())])
        print(f"  waves on CUs holding {nwg} waves: {sel.sum():5d}, lifetime {np.mean((end - start)[sel]):6.1f} us, "
              + ", ".join(f"{ph[sel, k].mean() / iters:.0f}" for k in range(8)))
    for sd in range(4):
        sel = simd == sd
        order = np.argsort(start[sel])
        print(f"  SIMD {sd}: {sel.sum()} waves; barrier wait per chunk by hardware wave slot: "
              + ", ".join(f"slot {w}: {ph[sel & (wave_id == w), 2].mean() / iters:.0f}" for w in sorted(set(wave_id[sel].tolist()))))
