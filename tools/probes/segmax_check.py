"""k_csr_segment_max (run-combining form) against the NumPy oracle on the full C3 cloud, with ties; and the loss of two identical
short trainings (is the whole-network step reproducible run to run?)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lattice_oracle as O
import lattice_net_amd as L
from lattice_net_amd import synthetic, ScatterMaxLattice
from lattice_net_amd.lattice_modules import DistributeLatticeModule
from lattice_net_amd import lattice as _lat
_lat.set_row_order("canonical")  # rows numbered in first-occurrence order, as the oracle (and a serial run of the reference) does
dev = torch.device("cuda", 0)
pos_np = synthetic.lidar_cloud(120000, 0)
lat = L.Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev)
dl, rows, idx, _ = DistributeLatticeModule()(lat, torch.from_numpy(pos_np).to(dev), torch.zeros((120000, 1), device=dev))
m = dl.nr_lattice_vertices()
tab = O.OracleHashTable(100000, 3)
oidx, _ = O.build_splat(tab, O.scale_positions(pos_np, np.full((3,), 0.9, np.float32)))
assert np.array_equal(oidx.ravel(), idx.cpu().numpy().ravel())
rng = np.random.default_rng(1)
for c in (32, 64, 7):
    f = rng.standard_normal((480000, c)).astype(np.float32)
    f[::3] = np.round(f[::3])
    vmax, arg = ScatterMaxLattice.apply(torch.from_numpy(f).to(dev), dl, idx)
    omax, oarg = O.scatter_max(f, oidx, m)
    print(c, "max equal", np.array_equal(vmax.cpu().numpy(), omax), "arg equal", np.array_equal(arg.cpu().numpy(), oarg),
          "counts equal", np.array_equal(dl.vertex_point_counts(idx).cpu().numpy(), O.vertex_point_counts(oidx, m)))
