#!/usr/bin/env python3
"""Do several LARGE hipGraphs replaying concurrently on their own streams survive hundreds of rounds on this stack?  Pure PyTorch
(an MLP with many small layers, forward + backward = ~600 graph nodes), no lattice kernels.  (GPU box.)"""
import os
import sys
import torch
HOST_JOIN = bool(os.environ.get("PROBE_HOST_JOIN"))  # device-level waits instead of event joins between the streams

torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda", 0)
K, ROUNDS, LAYERS = int(sys.argv[1]) if len(sys.argv) > 1 else 3, 600, 60
net = torch.nn.Sequential(*[m for _ in range(LAYERS) for m in (torch.nn.Linear(64, 64), torch.nn.GroupNorm(8, 64), torch.nn.ReLU())]).to(dev)
params = list(net.parameters())
opt = torch.optim.AdamW(params, lr=1e-4, fused=True)
caps = []
for k in range(K):
    x = torch.randn((50000, 64), device=dev)
    s = torch.cuda.Stream()

    def step(x=x):
        loss = net(x).square().mean()
        loss.backward()
        return loss.detach()

    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            for p in params:
                p.grad = None
            step()
    torch.cuda.synchronize()
    for p in params:
        p.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            loss = step()
    torch.cuda.synchronize()
    caps.append((g, s, loss, [p.grad for p in params]))
main = torch.cuda.current_stream()
bufs = [torch.empty_like(p) for p in params]
for it in range(ROUNDS):
    if HOST_JOIN:
        torch.cuda.synchronize()
    for g, s, _, _ in caps:
        if not HOST_JOIN:
            s.wait_stream(main)
        with torch.cuda.stream(s):
            g.replay()
    if HOST_JOIN:
        torch.cuda.synchronize()
    else:
        for g, s, _, _ in caps:
            main.wait_stream(s)
    torch._foreach_copy_(bufs, caps[0][3])
    for c in caps[1:]:
        torch._foreach_add_(bufs, c[3])
    for p, b in zip(params, bufs):
        p.grad = b
    opt.step()
    if it % 8 == 0:
        torch.cuda.synchronize()
    if it % 100 == 0:
        print("round", it, float(caps[0][2]), flush=True)
torch.cuda.synchronize()
print("done", K, "graphs x", ROUNDS, "rounds")
