"""Upper bound of what row locality can give the CSR segment reduce: the same token walk (the splat adjacency of a cloud) over
 (a) the real point rows (token t reads row t / 4), (b) a handful of rows (every read an L2 hit), (c) rows in a random permutation
 (no locality even in DRAM pages).  python tools/probes/reduce_locality_probe.py [C3|C5]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import lattice_net_amd as L  # noqa: E402
wl = sys.argv[1] if len(sys.argv) > 1 else "C5"
cfg = bench.WORKLOADS[wl]
n, v, half = cfg["n"], int(os.environ.get("V", cfg["v"])), bool(cfg.get("half"))  # V=96: the row width of the slice_classify scatter
dev = torch.device("cuda", 0)
pos = torch.from_numpy(bench.make_cloud(cfg["gen"], n, 0)).to(dev)
vals = torch.randn((n, v), device=dev)
vals = vals.half() if half else vals
lat = L.Lattice(sigmas=[cfg["sigma"]] * 3, capacity=cfg["capacity"], device=dev)
lat.begin_splat()
idx, w = lat.splat_standalone(pos, vals)
m = lat.nr_lattice_vertices()
T = idx.numel()


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3


dst = torch.zeros((m, v), dtype=torch.float32, device=dev)
real = lambda: lat._scatter_rows(vals, idx, w, dst, v, 4, v)                      # token t -> row t / 4
hot = lambda: lat._scatter_rows(vals, idx, w, dst, v, 1 << 14, v)                  # token t -> row t / 16384: ~117 distinct rows
perm = torch.randperm(n, device=dev)
vals_perm = vals[perm].contiguous()                                               # (same bytes, other places: results differ, timing only)
scattered = lambda: lat._scatter_rows(vals_perm, idx, w, dst, v, 4, v)
tok_rows = torch.randn((T, v), device=dev)
tok_rows = tok_rows.half() if half else tok_rows
once = lambda: lat._scatter_rows(tok_rows, idx, w, dst, v, 1, v)                   # every token its own row (read once: no reuse possible)
print(f"{wl}: {n} points, {m} vertices, {T} tokens, rows of {v * (2 if half else 4)} bytes")
for name, fn in (("real rows (t / 4)", real), ("L2-hot rows", hot), ("point rows permuted", scattered), ("one row per token", once)):
    print(f"  {name:24s} {timed(fn):7.1f} us")
