"""Debug aid: every torch.empty / empty_like / new_empty buffer is filled with NaN (floats) or a large sentinel (ints) before
use, so that a kernel reading memory it never wrote shows up as NaN instead of as run-to-run noise."""
import os, sys, tempfile, pathlib
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))

_empty, _empty_like = torch.empty, torch.empty_like
def _poison(t):
    if t.is_cuda and t.numel():
        if t.dtype in (torch.float32, torch.float64, torch.float16):
            t.fill_(float("nan"))
        elif t.dtype in (torch.int32, torch.int64):
            t.fill_(0x3FFFFFF)
    return t
torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))

import test_gpu_lnn as T
tmp = pathlib.Path(tempfile.mkdtemp())
net, lattice, pos_a, val_a, target_a = T.make_case(tmp, n=4000)
for it in range(3):
    net.zero_grad()
    ls, _ = net(lattice, pos_a, val_a)
    print("forward finite:", bool(torch.isfinite(ls).all()))
    torch.nn.functional.nll_loss(ls, target_a).backward()
    bad = [k for k, p in net.named_parameters() if not torch.isfinite(p.grad).all()]
    print("iteration", it, "non-finite grads:", bad[:8], len(bad))
