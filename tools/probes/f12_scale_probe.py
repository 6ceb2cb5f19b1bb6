#!/usr/bin/env python3
"""Where does the GPU network leave the float64 oracle-lattice network as the cloud grows?  The SemanticKITTI model (F12's cfg and seeded
parameters) on lidar clouds of n points: GPU float32 against this package's definition in float64 over tests/oracle_lattice (CPU).
usage: f12_scale_probe.py n [n ...]   (environment toggles select kernels: LN_CONV_EXACT_F32=1, LN_CONV_ROWS32=0, LN_GFB_WIDE=0, ...)"""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from tests.test_model_assembly import KITTI_CFG
from tests.oracle_lattice import OracleLattice
from make_reference_network_fixture import seeded_parameter
from lattice_net_amd import Lattice, ModelParams
from lattice_net_amd.models import LNN
from lattice_net_amd.synthetic import lidar_cloud
dev = torch.device("cuda", 0)
import lattice_net_amd as L
L.set_row_order(os.environ.get("F12_ROW_ORDER", "canonical"))
with tempfile.NamedTemporaryFile("w", suffix=".cfg") as f:
    f.write(KITTI_CFG); f.flush()
    mp = ModelParams.create(f.name)
    glat = Lattice.create(f.name, "lattice")
SEEDS = [int(x) for x in os.environ.get("F12_SEEDS", "5001").split(",")]
for n, seed in [(int(x), sd_) for x in sys.argv[1:] for sd_ in SEEDS]:
    pos = torch.from_numpy(lidar_cloud(n, 0))
    target = torch.from_numpy(np.random.default_rng(0).integers(0, 20, n))
    nets = {}
    for name, device, dtype in (("cpu", "cpu", torch.float64), ("gpu", dev, torch.float32)):
        net = LNN(20, mp, device=device).to(dtype)
        sd = net.state_dict()
        for i, k in enumerate(sd.keys()):
            sd[k].copy_(torch.from_numpy(seeded_parameter(i, k, sd[k].shape, seed)).to(dtype))
        lat = OracleLattice([0.9] * 3, 100000) if name == "cpu" else glat
        ls, logits = net(lat, pos.to(device), torch.zeros((n, 1), dtype=dtype, device=device))
        loss = torch.nn.functional.nll_loss(ls, target.to(device))
        loss.backward()
        nets[name] = (net, logits.detach().cpu().double().numpy(), float(loss))
    a, b = nets["gpu"][1], nets["cpu"][1]
    err = np.abs(a - b) / np.abs(b).max()
    g64 = dict(nets["cpu"][0].named_parameters())
    gmax = max(float(p.grad.norm()) for p in g64.values())
    l2, nrm = [], []
    for k, p in nets["gpu"][0].named_parameters():
        g, r = p.grad.detach().cpu().double(), g64[k].grad
        l2.append((float((g - r).norm() / max(float(r.norm()), 1e-3 * gmax)), k))
        nrm.append((abs(float(g.norm()) - float(r.norm())) / max(float(r.norm()), 1e-3 * gmax), k))
    rel_l2 = float(np.linalg.norm(a - b) / np.linalg.norm(b))
    print(f"n {n:7d} seed {seed}: logits max {err.max():.2e} median {np.median(err):.2e} rel L2 {rel_l2:.2e} within 1e-4: {100.0 * (err.max(1) <= 1e-4).mean():5.1f} %; "
          f"loss {nets['gpu'][2]:.7f} vs {nets['cpu'][2]:.7f}; gradients: worst rel L2 {max(l2)[0]:.2e} ({max(l2)[1]}), worst norm error {max(nrm)[0]:.2e}", flush=True)
