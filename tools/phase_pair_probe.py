#!/usr/bin/env python3
"""Which phases of the step saturate the chip on their own?  (GPU box.)
Every phase of the C3 step — build, accumulate + neighbour list, convolution, slice, slice backward, convolution backward — is
captured into its own hipGraph for K independent scans (own lattice and tensors each) and replayed (a) one scan at a time on one
stream, (b) K scans concurrently on K streams.  ratio = time per scan concurrent / solo: ~1/K = pure latency (K copies fit
side by side), ~1 = the phase already saturates some unit of the chip."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattice_net_amd as L  # noqa: E402
from lattice_net_amd import synthetic  # noqa: E402

torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda", 0)
n, v, f, sigma, cap = 120000, 32, 32, 0.9, 100000
K = int(os.environ.get("PROBE_K", 4))
ROWS = 49152
R = int(os.environ.get("PROBE_R", 8))


class Scan:
    def __init__(self, k):
        rng = np.random.default_rng(k)
        self.pos = torch.from_numpy(synthetic.lidar_cloud(n, k)).to(dev)
        self.vals = torch.from_numpy(rng.standard_normal((n, v)).astype(np.float32)).to(dev)
        self.G = torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).to(dev)
        self.W = (torch.rand((9 * v, f), device=dev) - 0.5)
        self.lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev)
        self.lat.set_static_rows(ROWS)
        self.stream = torch.cuda.Stream()
        # state produced phase by phase (eagerly once, then each phase is captured reading the previous phase's tensors)
        self.idx, self.w = None, None

    # phases (each leaves its outputs in attributes with stable storage)
    def p_build(self):
        self.lat.begin_splat()
        self.idx, self.w = self.lat.just_create_verts(self.pos, True)

    def p_accumulate(self):
        lat = self.lat
        self.lv = torch.zeros((ROWS, v), device=dev)
        lat._accumulate_and_prefetch(self.vals, self.idx, self.w, self.lv, v, 4, n * 4)

    def p_conv(self):
        self.lat.set_values(self.lv)
        self.cl = self.lat.convolve_im2row_standalone(self.W, 1, self.lat, False)
        self.cv = self.cl.values()

    def p_slice(self):
        self.out = self.cl.slice_standalone_with_precomputation(self.pos, self.idx, self.w)

    def p_slice_bwd(self):
        self.cl.slice_backwards_standalone_with_precomputation_no_homogeneous(self.pos, self.G, self.idx, self.w)
        self.gcv = self.cl.values()

    def p_conv_bwd(self):
        self.lat.set_values(self.lv)
        self.gv, self.gw = self.lat.convolve_im2row_backward(self.gcv, self.W, 1, self.lat, self.lat)


PHASES = ["p_build", "p_accumulate", "p_conv", "p_slice", "p_slice_bwd", "p_conv_bwd"]
scans = [Scan(k) for k in range(K)]
graphs = {}
for s in scans:  # every workspace / cache reaches its final size before anything is captured
    with torch.cuda.stream(s.stream):
        for _ in range(2):
            for ph in PHASES:
                getattr(s, ph)()
torch.cuda.synchronize()
for ph in PHASES:
    for s in scans:
        with torch.cuda.stream(s.stream):
            for _ in range(2):
                getattr(s, ph)()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s.stream):
            with torch.cuda.graph(g, stream=s.stream):
                for _ in range(R):  # R copies per replay: the host's ~10 us per replay stays off the measurement
                    getattr(s, ph)()
        torch.cuda.synchronize()
        with torch.cuda.stream(s.stream):
            g.replay()  # a capture executes nothing: the next phase's warm-up needs this phase's outputs to be real
        torch.cuda.synchronize()
        graphs[(ph, id(s))] = g


def replay(ph, which, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for s in which:
            with torch.cuda.stream(s.stream):
                graphs[(ph, id(s))].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / len(which) / R * 1e6


print(f"{'phase':14s} {'solo us':>9s} " + " ".join(f"{'x%d us/scan' % k:>12s} {'ratio':>6s}" for k in (2, K)))
tot = {1: 0.0, 2: 0.0, K: 0.0}
tot_add = lambda k, t: tot.__setitem__(k, tot[k] + t)
for ph in PHASES:
    replay(ph, scans, 5)
    solo = replay(ph, scans[:1], 40)
    row = f"{ph:14s} {solo:9.1f} "
    tot[1] += solo
    for k in (2, K):
        t = replay(ph, scans[:k], 40)
        if not (k == 2 and K == 2 and False):
            tot[k] += t
        row += f"{t:12.1f} {t / solo:6.2f} "
    print(row)
print(f"{'sum':14s} {tot[1]:9.1f} " + " ".join(f"{tot[k]:12.1f} {tot[k] / tot[1]:6.2f}" for k in (2, K)))
