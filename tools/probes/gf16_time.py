"""Per-launch time of the fp16 filter gradient (k_grad_filter_mfma_f16 + slab sum) at C5's shape.  LN_GF16_EG1=1: one slot per workgroup."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import lattice_net_amd as L  # noqa: E402
lib = L.load_library(); dev = torch.device("cuda", 0)
cfg = bench.WORKLOADS["C5"]
pos = torch.from_numpy(bench.make_cloud(cfg["gen"], cfg["n"], 0)).to(dev)
lat = L.Lattice(sigmas=[0.9] * 3, capacity=cfg["capacity"], device=dev); lat.begin_splat(); lat.just_create_verts(pos, False)
m = lat.nr_lattice_vertices()
lv = torch.randn((m, 64), device=dev).half().requires_grad_(True)
fb = (torch.randn((9 * 64, 64), device=dev) * 0.05).requires_grad_(True)
g = torch.randn((m, 64), device=dev).half()
def step():
    lv.grad = fb.grad = None
    y, _ = L.ConvIm2RowLattice.apply(lv, lat, fb, 1); y.backward(g)
for _ in range(60): step()
torch.cuda.synchronize()
for name in (b"k_grad_filter_mfma_f16", b"k_reduce_slabs", b"k_conv_mfma_f16"):
    lib.ln_profile_begin(name, 256)
    for _ in range(20): step()
    torch.cuda.synchronize()
    ms, cnt = C.c_double(0), C.c_int(0); lib.ln_profile_end(C.byref(ms), C.byref(cnt))
    print(f"{name.decode():26s} {ms.value / max(cnt.value, 1) * 1e3:7.1f} us x {cnt.value / 20:g} per step   (m = {m})")
