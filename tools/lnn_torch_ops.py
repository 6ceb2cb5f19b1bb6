#!/usr/bin/env python3
"""Which torch operators (not this library's kernels) still run in the LNN training step: torch.profiler table by GPU time,
with input shapes.  Usage: python tools/lnn_torch_ops.py [kitti|shapenet|scannet]"""
import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_lnn import PRESETS  # noqa: E402
from lattice_net_amd import Lattice, ModelParams, synthetic  # noqa: E402
from lattice_net_amd.losses import nll_loss_gather  # noqa: E402
from lattice_net_amd.models import LNN  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "kitti"
preset = PRESETS[name]
dev = torch.device("cuda", 0)
with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
    f.write(preset["cfg"])
torch.manual_seed(0)
mp = ModelParams.create(f.name)
lattice = Lattice.create(f.name, "lattice")
os.unlink(f.name)  # the readers are done with the temporary cfg
net = LNN(preset["classes"], mp)
n = preset["n"]
gen = {"lidar": synthetic.lidar_cloud, "box": synthetic.box_surface_cloud, "planes": synthetic.planes_cloud}[preset["cloud"]]
pos = torch.from_numpy(gen(n, 0)).to(dev)
vals = torch.zeros((n, 1), device=dev) if preset["values"] == 1 else torch.rand((n, preset["values"]), device=dev)
target = torch.from_numpy(np.random.default_rng(0).integers(0, preset["classes"], n)).to(dev)
opt = None
def step():
    global opt
    ls, _ = net(lattice, pos, vals)
    loss = nll_loss_gather(ls, target)
    if opt is None:
        opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-4, amsgrad=True, fused=True)
    opt.zero_grad()
    loss.backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
steps = 3
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith(("aten::", "Optimizer", "_fused"))]
def gpu_us(e):
    return getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0.0)
rows.sort(key=gpu_us, reverse=True)
total = sum(gpu_us(e) for e in rows)
print(f"torch operators: {total / steps:.0f} us of GPU time per step")
for e in rows[:30]:
    print(f"{e.key[:34]:34s} {e.count / steps:6.1f}/step {gpu_us(e) / steps:8.1f} us/step   {str(e.input_shapes)[:110]}")

fills = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::new_zeros", "aten::masked_fill_")]
print("fill-like operators by input shape (all of them; aten::zeros / zero_ call fill_):")
for e in sorted(fills, key=lambda e: -e.count):
    print(f"  {e.key:18s} {e.count / steps:6.1f}/step {gpu_us(e) / steps:8.1f} us/step   {str(e.input_shapes)[:110]}")

# the operator chain above every fill (the autograd engine and C++ operators create zeros no Python hook sees)
import collections as _c
chains = _c.Counter()
for e in prof.events():
    if e.name == "aten::fill_" and getattr(e, "device_time_total", 0) > 0:
        names, p_ = [], e.cpu_parent
        while p_ is not None and len(names) < 4:
            names.append(p_.name[:48])
            p_ = p_.cpu_parent
        chains[(str(e.input_shapes)[:40], " < ".join(names))] += 1
print("fills that launched a kernel, by (shape, parents):")
for (shape, chain), c in chains.most_common(40):
    print(f"  {c / steps:5.1f}/step  {shape:40s} {chain}")

# ---- where the fill / zero launches come from: Python call sites of the zero-filling constructors and in-place fills (CUDA tensors)
if os.environ.get("LNN_FILL_SITES", "1") == "1":
    import collections
    import traceback
    sites = collections.Counter()

    def site():
        for fr in reversed(traceback.extract_stack()[:-2]):
            if "lattice_net_amd" in fr.filename or fr.filename.endswith(("bench_lnn.py", "lnn_torch_ops.py")):
                return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line.strip()[:90]}"
        return "torch internals"

    def wrap_fn(mod, name):
        orig = getattr(mod, name)

        def fn(*a, **k):
            out = orig(*a, **k)
            t = out if torch.is_tensor(out) else (a[0] if a and torch.is_tensor(a[0]) else None)
            if t is not None and t.is_cuda:
                sites[(name, site())] += 1
            return out
        setattr(mod, name, fn)
        return orig

    saved = [(torch, n_, wrap_fn(torch, n_)) for n_ in ("zeros", "zeros_like", "full", "ones", "ones_like", "full_like")]
    saved += [(torch.Tensor, n_, wrap_fn(torch.Tensor, n_)) for n_ in ("zero_", "fill_", "new_zeros", "new_full", "masked_fill", "masked_fill_")]
    step()
    torch.cuda.synchronize()
    for mod, n_, orig in saved:
        setattr(mod, n_, orig)
    print(f"zero-filling calls of one step by Python call site ({sum(sites.values())} in all; autograd's own zero gradients are not seen here):")
    for (fn_name, where), c in sites.most_common(40):
        print(f"  {c:3d} x {fn_name:12s} {where}")
