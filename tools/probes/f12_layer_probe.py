#!/usr/bin/env python3
"""Layer by layer: the SemanticKITTI model on a lidar cloud of n points, GPU float32 (canonical row order) against float64 over the
oracle lattice; the first tensor-valued output of every module, in execution order, with its error."""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from tests.test_model_assembly import KITTI_CFG
from tests.oracle_lattice import OracleLattice
from make_reference_network_fixture import seeded_parameter
import lattice_net_amd as L
from lattice_net_amd import Lattice, ModelParams
from lattice_net_amd.models import LNN
from lattice_net_amd.synthetic import lidar_cloud
dev = torch.device("cuda", 0)
L.set_row_order("canonical")
n = int(sys.argv[1])
with tempfile.NamedTemporaryFile("w", suffix=".cfg") as f:
    f.write(KITTI_CFG); f.flush()
    mp = ModelParams.create(f.name)
    glat = Lattice.create(f.name, "lattice")
pos = torch.from_numpy(lidar_cloud(n, 0))
rec = {}
for name, device, dtype in (("cpu", "cpu", torch.float64), ("gpu", dev, torch.float32)):
    net = LNN(20, mp, device=device).to(dtype)
    sd = net.state_dict()
    for i, k in enumerate(sd.keys()):
        sd[k].copy_(torch.from_numpy(seeded_parameter(i, k, sd[k].shape, 5001)).to(dtype))
    log = []
    def hook(mod_name):
        def fn(mod, inp, out):
            t = out[0] if isinstance(out, (tuple, list)) else out
            if torch.is_tensor(t) and t.dtype.is_floating_point:
                log.append((mod_name, t.detach().cpu().double().numpy()))
        return fn
    for mn, m in net.named_modules():
        if mn:
            m.register_forward_hook(hook(mn))
    lat = OracleLattice([0.9] * 3, 100000) if name == "cpu" else glat
    with torch.no_grad():
        net(lat, pos.to(device), torch.zeros((n, 1), dtype=dtype, device=device))
    rec[name] = log
print(f"n {n}: {len(rec['cpu'])} / {len(rec['gpu'])} recorded outputs")
def keyed(log):
    seen, out = {}, {}
    for k, t in log:
        seen[k] = seen.get(k, 0) + 1
        out[(k, seen[k])] = t
    return out
g = keyed(rec["gpu"])
for key, b in keyed(rec["cpu"]).items():
    if key not in g:
        continue
    a = g[key]
    if a.shape != b.shape:
        print(f"  {key[0]:60s} SHAPES {a.shape} vs {b.shape}")
        continue
    e = np.abs(a - b) / max(np.abs(b).max(), 1e-30)
    flag = "  <<<<" if e.max() > 1e-4 else ""
    print(f"  {key[0]:60s} {str(a.shape):16s} max {e.max():.2e} median {np.median(e):.2e} rows above 1e-4: {(e.reshape(e.shape[0], -1).max(1) > 1e-4).sum()}{flag}")
