#!/usr/bin/env python3
"""Runs the SemanticKITTI-size network (fixture F12 case) several times in default and deterministic mode: are logits / gradients bitwise
identical run to run, and does the deterministic run land on the no-flip side of the reference fixture?"""
import hashlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from lattice_net_amd import lattice as LT
from tests.test_model_assembly import kitti_fixture_case
from make_reference_network_fixture import logits_sample_index
dev = torch.device("cuda", 0)
torch.autograd.set_multithreading_enabled(False)
LT.set_row_order("canonical")
fx, net, lattice, pos, target = kitti_fixture_case(dev, torch.float32)
n = pos.shape[0]
pos, target = pos.to(dev), target.to(dev)
vals = torch.zeros((n, 1), device=dev)
def digest(t):
    return hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]
for mode in (False, True):
    LT.set_deterministic(mode)
    seen = []
    for run in range(int(os.environ.get("RUNS", "5"))):
        net.zero_grad(set_to_none=True)
        logsoftmax, logits = net(lattice, pos, vals)
        loss = torch.nn.functional.nll_loss(logsoftmax, target)
        loss.backward()
        torch.cuda.synchronize()
        grads = hashlib.sha1(b"".join(p.grad.detach().cpu().numpy().tobytes() for p in net.parameters() if p.grad is not None)).hexdigest()[:12]
        lg = logits.detach().cpu().double().numpy()[logits_sample_index(n, fx["logits"].shape[0])]
        err = np.abs(lg - fx["logits"]) / np.abs(fx["logits"]).max()
        seen.append((digest(logits), grads))
        print(f"deterministic={mode} run {run}: logits {seen[-1][0]} grads {seen[-1][1]} max logit err {err.max():.2e} median {np.median(err):.2e} within1e-4 {(err.max(1) <= 1e-4).mean():.3f}", flush=True)
    print(f"deterministic={mode}: distinct logits {len(set(s[0] for s in seen))} distinct grads {len(set(s[1] for s in seen))}")
