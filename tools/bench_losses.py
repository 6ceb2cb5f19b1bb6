#!/usr/bin/env python3
"""Time of the training losses on a [120k, 20] prediction: NLL (gather form), Lovasz-Softmax, soft Dice; forward + backward."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lattice_net_amd.losses import GeneralizedSoftDiceLoss, LovaszSoftmax, nll_loss_gather
dev = torch.device("cuda", 0)
n, c = 120000, 20
logits = torch.randn((n, c), device=dev, requires_grad=True)
target = torch.randint(0, c, (n,), device=dev)
lov, dice = LovaszSoftmax(ignore_index=0), GeneralizedSoftDiceLoss(ignore_index=0)
def run(fn):
    for _ in range(3):
        logits.grad = None
        fn(torch.log_softmax(logits, 1)).backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        logits.grad = None
        fn(torch.log_softmax(logits, 1)).backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3
print(f"log_softmax + nll   {run(lambda lp: nll_loss_gather(lp, target, 0)):6.3f} ms")
print(f"log_softmax + lovasz {run(lambda lp: lov(lp, target)):6.3f} ms")
print(f"log_softmax + dice  {run(lambda lp: dice(lp, target)):6.3f} ms")
