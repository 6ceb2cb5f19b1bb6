#!/usr/bin/env python3
"""Speed of light of random 128-byte row gathers on this chip, measured with the slice-forward kernel (GPU box):
out[p] = sum of 4 rows values[idx[p, r]] * w[p, r], 120k points x 4 rows of V = 32 floats, for value tables of different sizes
and index patterns.  Answers: is a gather kernel that takes X us for 480k row gathers at the hardware's limit?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lattice_net_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
N, V = 120000, 32
st = _lib.stream_ptr(dev)

def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

rng = np.random.default_rng(0)
w = torch.rand((N * 4,), device=dev)
out = torch.empty((N, V), device=dev)
for rows in (4096, 46538, 120000, 480000, 4000000):
    vals = torch.randn((rows, V), device=dev)
    for name, idx_np in (("random", rng.integers(0, rows, N * 4)), ("sequential", (np.arange(N * 4) // 4) % rows),
                         ("4 consecutive rows", ((rng.integers(0, rows - 4, N)[:, None] + np.arange(4)[None]).reshape(-1)))):
        idx = torch.from_numpy(idx_np.astype(np.int32)).to(dev)
        t = timed(lambda: _lib.check(lib.ln_slice_forward(_lib.ptr(vals), _lib.ptr(idx), _lib.ptr(w), N, 3, V, _lib.ptr(out), st)))
        print(f"table {rows:8d} rows ({rows * V * 4 / 1e6:7.1f} MB)  {name:20s} {t:7.1f} us   {N * 4 * V * 4 / t / 1e6:6.2f} TB/s of gathered rows")
