import ctypes as C, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lattice_net_amd as L
from lattice_net_amd import synthetic
from lattice_net_amd.lattice_funcs import ConvIm2RowLattice
lib = L.load_library(); dev = torch.device("cuda", 0)
pos = torch.from_numpy(synthetic.lidar_cloud(120000, 0)).to(dev)
lat = L.Lattice(sigmas=[0.9]*3, capacity=100000, device=dev); lat.begin_splat(); lat.just_create_verts(pos, False); m = lat.nr_lattice_vertices()
for v, f in ((64, 64), (128, 128), (96, 96), (128, 64), (64, 32)):
    lv = torch.randn((m, v), device=dev, requires_grad=True); fb = (torch.randn((9*v, f), device=dev)*0.05).requires_grad_(True); g = torch.randn((m, f), device=dev)
    def step():
        lv.grad = fb.grad = None
        y, _ = ConvIm2RowLattice.apply(lv, lat, fb, 1); y.backward(g)
    for _ in range(3): step()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < float(os.environ.get("LN_OPS_PREWARM_MS", "30")):  # the clocks of a busy GPU
        for _ in range(4): step()
        torch.cuda.synchronize()
    out = []
    for name in (b"k_grad_filter_mfma", b"k_reduce_slabs", b"k_conv_split_bank"):
        lib.ln_profile_begin(name, 64)
        for _ in range(10): step()
        torch.cuda.synchronize()
        ms, cnt = C.c_double(0), C.c_int(0); lib.ln_profile_end(C.byref(ms), C.byref(cnt))
        out.append(ms.value / max(cnt.value, 1) * 1e3)
    ref = torch.zeros_like(fb)
    nb = lat.neighbours(lat, 1, False).long()
    with torch.no_grad():
        for e in range(9):
            ok = nb[:, e] >= 0
            ref[e * v:(e + 1) * v] = (lv.detach()[nb[ok, e]].double().T @ g[ok].double()).float()
    err = float((fb.grad - ref).abs().max() / ref.abs().max())
    # (the slab sum rides in the bank-split launch of the value-gradient convolution since round 5: a separate k_reduce_slabs launch only
    #  where that convolution has no bank to split; the split launches are two per step: forward bank, value-gradient bank + slab sum)
    print(f"V {v} F {f}: grad filter {out[0]:.1f} us + slab sum {out[1]:.1f} us (bank split launches, one of them carrying the sum: {out[2]:.1f} us each)   rel err {err:.1e}")
