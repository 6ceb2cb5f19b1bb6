#!/usr/bin/env python3
"""One 40-step training run with the captured whole-network step (exit code 0 = completed, loss curve sane).  Looped by hand to
count how often replays alternating with eager AdamW steps abort on this stack.  argv[1]: "main" (replay on the current stream),
"side" (replay on the capture stream, the main stream waits), "sync" (as main + device synchronize after every optimizer step)."""
import pathlib
import sys
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_gpu_lnn_static as T  # noqa: E402
from lattice_net_amd import CapturedNetworkStep  # noqa: E402
from lattice_net_amd.losses import nll_loss_gather  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "main"
torch.autograd.set_multithreading_enabled(False)
tmp = pathlib.Path("/tmp/flake")
tmp.mkdir(exist_ok=True)
import os
T.CFG = T.CFG.replace("hash_table_capacity: 60000", "hash_table_capacity: " + os.environ.get("FLAKE_CAP", "60000"))
net, lattice, pos, vals, target = T._setup(tmp, n=int(os.environ.get("FLAKE_N", 20000)))
opt = torch.optim.AdamW(net.parameters(), lr=2e-3, weight_decay=1e-4, amsgrad=True, fused=(mode != "foreach"))


def one():
    logsoftmax, _ = net(lattice, pos, vals)
    loss = nll_loss_gather(logsoftmax, target)
    loss.backward()
    return loss.detach()


for _ in range(int(os.environ.get("FLAKE_PRE", 0))):  # eager training steps before the capture
    opt.zero_grad()
    one()
    opt.step()
torch.cuda.synchronize()
for p in net.parameters():
    p.grad = None
cap = CapturedNetworkStep(one, lattice, net.parameters(), stream=torch.cuda.Stream() if mode == "side" else None)
main = torch.cuda.current_stream()
losses = []
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    if mode == "side":
        cap.stream.wait_stream(main)
    loss = cap.launch()
    if mode == "side":
        main.wait_stream(cap.stream)
    losses.append(loss.clone())
    cap.bind_gradients()
    opt.step()
    if mode == "sync":
        torch.cuda.synchronize()
torch.cuda.synchronize()
ls = [float(x) for x in losses]
assert ls[-1] < ls[0], ls
print("ok", mode, ls[0], ls[-1])
