"""Per-launch times of the fused PointNet vertex reduction (k_csr_segment_max + decode, token-major backward) on the C3 cloud.
    LATTICE_NET_LIB=lattice_net_amd/liblatticenet_hip_<variant>.so python tools/probes/pointnet_reduce_time.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from ops_roofline import _profile  # noqa: E402
import lattice_net_amd as L  # noqa: E402
from lattice_net_amd import synthetic  # noqa: E402
from lattice_net_amd.lattice_modules import DistributeLatticeModule, PointNetReduceFunction  # noqa: E402
dev = torch.device("cuda", 0); lib = L.load_library()
pos = torch.from_numpy(synthetic.lidar_cloud(120000, 0)).to(dev)
lat = L.Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev)
dl, rows, idx, _ = DistributeLatticeModule()(lat, pos, torch.zeros((120000, 1), device=dev))
for c in (32, 64):
    feat = torch.randn((rows.shape[0], c), device=dev, requires_grad=True)
    g = torch.randn((dl.nr_lattice_vertices(), 2 * c), device=dev)
    def both():
        feat.grad = None
        PointNetReduceFunction.apply(feat, rows, dl, idx).backward(g)
    k = _profile(lib, both, 20)
    print(f"C = {c}: " + ", ".join(f'{x["kernel"]} {x["avg_us"]:.1f} us' for x in k))
