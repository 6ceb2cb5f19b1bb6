#!/usr/bin/env python3
"""The SemanticKITTI model on the GPU under the two vertex numberings (slot order = the shipped default, canonical = the reference's
serial numbering): the first tensor-valued output of every module, rows matched through the lattice keys."""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from tests.test_model_assembly import KITTI_CFG
from make_reference_network_fixture import seeded_parameter
import lattice_net_amd as L
from lattice_net_amd import Lattice, ModelParams
from lattice_net_amd.models import LNN
from lattice_net_amd.synthetic import lidar_cloud
dev = torch.device("cuda", 0)
n = int(sys.argv[1])
with tempfile.NamedTemporaryFile("w", suffix=".cfg") as f:
    f.write(KITTI_CFG); f.flush()
    mp = ModelParams.create(f.name)
    glat = Lattice.create(f.name, "lattice")
pos = torch.from_numpy(lidar_cloud(n, 0)).to(dev)
rec = {}
for order in ("canonical", "slot"):
    L.set_row_order(order)
    net = LNN(20, mp, device=dev)
    sd = net.state_dict()
    for i, k in enumerate(sd.keys()):
        sd[k].copy_(torch.from_numpy(seeded_parameter(i, k, sd[k].shape, 5001)).float())
    log = []
    def hook(mod_name):
        def fn(mod, inp, out):
            t = out[0] if isinstance(out, (tuple, list)) else out
            ls = out[1] if isinstance(out, (tuple, list)) and len(out) > 1 else None
            if torch.is_tensor(t) and t.dtype.is_floating_point:
                keys = None
                lat = getattr(ls, "lattice", ls)
                try:
                    m = lat.nr_lattice_vertices()
                    if m == t.shape[0]:
                        keys = lat.hash_table().m_keys_tensor[:m].cpu().numpy()
                except Exception:
                    keys = None
                log.append((mod_name, t.detach().cpu().double().numpy(), keys))
        return fn
    for mn, m in net.named_modules():
        if mn:
            m.register_forward_hook(hook(mn))
    with torch.no_grad():
        net(glat, pos, torch.zeros((n, 1), device=dev))
    rec[order] = log
print(f"n {n}: {len(rec['canonical'])} recorded outputs")
for (ka, a, keys_a), (kb, b, keys_b) in zip(rec["slot"], rec["canonical"]):
    assert ka == kb
    if keys_a is not None and keys_b is not None:
        pa, pb = np.lexsort(keys_a.T[::-1]), np.lexsort(keys_b.T[::-1])
        assert np.array_equal(keys_a[pa], keys_b[pb]), "the two numberings hold different vertex sets"
        a, b = a[pa], b[pb]
        how = "keys"
    else:
        how = "rows" if a.shape[0] == n else "NO KEYS"
    e = np.abs(a - b) / max(np.abs(b).max(), 1e-30)
    flag = "  <<<<" if e.max() > 1e-5 else ""
    print(f"  {ka:60s} {str(a.shape):16s} ({how}) max {e.max():.2e} median {np.median(e):.2e} rows above 1e-5: {(e.reshape(e.shape[0], -1).max(1) > 1e-5).sum()}{flag}")
