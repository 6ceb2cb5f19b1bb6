#!/usr/bin/env python3
"""F12 (SemanticKITTI network at 120 000 points) on the GPU against the fixture: error statistics of the logits and of every gradient.
LN_CONV_EXACT_F32=1 / LN_CONV_ROWS32=0 / LN_GFB_WIDE=0 select other kernels for an A/B."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from tests.test_model_assembly import kitti_fixture_case
from make_reference_network_fixture import gradient_sample_index, logits_sample_index
dev = torch.device("cuda", 0)
import lattice_net_amd as L
L.set_row_order(os.environ.get("F12_ROW_ORDER", "canonical"))  # the fixture's "vertex 0" (zeroed by PointNet, lattice_modules.py:711) is the serial numbering's
fx, net, lattice, pos, target = kitti_fixture_case(dev, torch.float32)
n = pos.shape[0]
ls, logits = net(lattice, pos.to(dev), torch.zeros((n, 1), device=dev))
loss = torch.nn.functional.nll_loss(ls, target.to(dev))
loss.backward()
torch.cuda.synchronize()
lg = logits.detach().cpu().double().numpy()[logits_sample_index(n, fx["logits"].shape[0])]
err = np.abs(lg - fx["logits"]) / np.abs(fx["logits"]).max()
print(f"logits: max rel err {err.max():.3e}, 99.9 % {np.quantile(err, 0.999):.3e}, median {np.median(err):.3e}; points above 1e-4: {(err.max(1) > 1e-4).sum()} of {err.shape[0]}; "
      f"loss {float(loss):.8f} vs {float(fx['loss']):.8f}")
named = dict(net.named_parameters())
gmax = float(np.nanmax(fx["grad_norms"]))
rows = []
for i, k in enumerate(str(k) for k in fx["keys"]):
    if k not in named:
        continue
    g = named[k].grad.detach().cpu().double().numpy().reshape(-1)
    norm = float(np.linalg.norm(g))
    ref = fx[f"grad_full/{i}"] if f"grad_full/{i}" in fx else fx[f"grad_sample/{i}"]
    if f"grad_full/{i}" not in fx:
        g = g[gradient_sample_index(g.size)]
    e = float(np.abs(g - ref).max()) / max(float(np.abs(ref).max()), 1e-3 * gmax / np.sqrt(max(ref.size, 1)))
    rows.append((e, abs(norm - float(fx["grad_norms"][i])) / max(float(fx["grad_norms"][i]), 1e-3 * gmax), k))
rows.sort(reverse=True)
print("worst gradients (max-entry error relative to the tensor's largest entry, norm error):")
for e, en, k in rows[:12]:
    print(f"  {e:.3e}  {en:.3e}  {k}")
print(f"gradients above 1e-4: {sum(1 for r in rows if r[0] > 1e-4)} of {len(rows)}")
