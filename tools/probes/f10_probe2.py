"""Where does the GPU backward depart from the float64 oracle network?  Same seeded parameters (F10), per-module output gradients."""
import numpy as np, torch, tempfile, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests/golden")
from make_reference_network_fixture import seeded_parameter
from tests.test_oracle_network import CFG, make_oracle_case
from lattice_net_amd import ModelParams, Lattice
from lattice_net_amd.models import LNN
from lattice_net_amd.synthetic import box_surface_cloud
from lattice_net_amd import lattice as LT
LT.set_row_order("canonical")
dev = torch.device("cuda", 0)
fx = np.load("tests/golden/F10_reference_lnn.npz")
ref = [str(k) for k in fx["keys"]]
n = int(fx["n_points"])
pos = torch.from_numpy(box_surface_cloud(n, 0))
target = torch.from_numpy(np.random.default_rng(0).integers(0, 6, n))
net64, olat, _, _, _ = make_oracle_case(n=8)
with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
    f.write(CFG); path = f.name
mp = ModelParams.create(path); lat = Lattice.create(path, "lattice")
net = LNN(6, mp)
for nn_, dt in ((net64, torch.float64), (net, torch.float32)):
    sd = nn_.state_dict()
    for i, k in enumerate(ref): sd[k].copy_(torch.from_numpy(seeded_parameter(i, k, sd[k].shape)).to(dt))
def instrument(nn_, store):
    for name, mod in nn_.named_modules():
        if name == "" : continue
        def hook(m, inp, out, name=name):
            t = out[0] if isinstance(out, tuple) else out
            if torch.is_tensor(t) and t.requires_grad:
                t.retain_grad(); store[name] = t
        mod.register_forward_hook(hook)
s64, s32 = {}, {}
instrument(net64, s64); instrument(net, s32)
l64, _ = net64(olat, pos, torch.zeros((n, 1), dtype=torch.float64))
torch.nn.functional.nll_loss(l64, target).backward()
l32, _ = net(lat, pos.to(dev), torch.zeros((n, 1), device=dev))
torch.nn.functional.nll_loss(l32, target.to(dev)).backward()
for name in s64:
    if name not in s32 or s64[name].grad is None or s32[name].grad is None: continue
    a, b = s32[name].grad.cpu().double().numpy(), s64[name].grad.numpy()
    fa, fb = s32[name].detach().cpu().double().numpy(), s64[name].detach().numpy()
    if a.shape != b.shape: print(name, "shape", a.shape, b.shape); continue
    print("%-60s fwd %.1e  grad %.1e  (max|g| %.1e)" % (name, np.abs(fa - fb).max() / max(np.abs(fb).max(), 1e-30), np.abs(a - b).max() / max(np.abs(b).max(), 1e-30), np.abs(b).max()))
name = "resnet_blocks_per_up_lvl_list.0.0.conv.conv"
a, b = s32[name].grad.cpu().double().numpy(), s64[name].grad.numpy()
err = np.abs(a - b) / np.abs(b).max()
print("shape", a.shape, "elements > 1e-4:", int((err > 1e-4).sum()), "per channel max:", np.round(err.max(0) * 1e3, 2))
x64 = s64[name].detach().numpy()
print("channel mean/std of the norm's input:", np.round(np.abs(x64.mean(0)) / (x64.std(0) + 1e-30), 1))
ch = int(err.max(0).argmax())
print("worst channel", ch, "rows > 1e-4:", np.nonzero(err[:, ch] > 1e-4)[0][:20], "x:", x64[np.nonzero(err[:, ch] > 1e-4)[0][:5], ch])
