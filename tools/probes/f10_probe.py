import numpy as np, torch, tempfile, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests/golden")
from make_reference_network_fixture import seeded_parameter, gradient_sample_index
from tests.test_oracle_network import CFG
from lattice_net_amd import ModelParams, Lattice
from lattice_net_amd.models import LNN
from lattice_net_amd.synthetic import box_surface_cloud
from lattice_net_amd import lattice as LT
LT.set_row_order(os.environ.get("ROW_ORDER", "canonical"))
dev = torch.device("cuda", 0)
with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
    f.write(CFG); path = f.name
mp = ModelParams.create(path); lat = Lattice.create(path, "lattice")
fx = np.load(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/F10_reference_lnn.npz")
SEED = int(fx["param_seed"])
DEV = os.environ.get("F10_DEVICE", "cuda")
ref = [str(k) for k in fx["keys"]]
n=int(fx["n_points"])
pos = torch.from_numpy(box_surface_cloud(n, 0)).to(dev)
target = torch.from_numpy(np.random.default_rng(0).integers(0, 6, n)).to(dev)
net = LNN(6, mp)
sd = net.state_dict()
for i,k in enumerate(ref): sd[k].copy_(torch.from_numpy(seeded_parameter(i,k,sd[k].shape,SEED)).float())
ls, logits = net(lat, pos, torch.zeros((n,1), device=dev))
loss = torch.nn.functional.nll_loss(ls, target); loss.backward()
print("logits rel", np.abs(logits.detach().cpu().numpy()-fx["logits"]).max()/np.abs(fx["logits"]).max(), "loss", float(loss), float(fx["loss"]))
named = dict(net.named_parameters())
rows=[]
for i,k in enumerate(ref):
    if k not in named: continue
    g = named[k].grad.cpu().numpy().astype(np.float64).reshape(-1)
    if f"grad_full/{i}" in fx: r = fx[f"grad_full/{i}"]
    else: r = fx[f"grad_sample/{i}"]; g = g[gradient_sample_index(g.size)]
    rows.append((np.abs(g-r).max()/max(np.abs(r).max(),1e-30), np.abs(g-r).max(), np.abs(r).max(), k))
rows.sort(reverse=True)
print("SEED", SEED, "worst gradient", "%.2e" % rows[0][0], rows[0][3])
for r in rows[:int(os.environ.get("F10_TOP", "3"))]: print("%.2e  abs %.2e  max|ref| %.2e  %s" % r)
