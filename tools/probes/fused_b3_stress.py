#!/usr/bin/env python3
"""Stress of the fused convolution backward under concurrency: K captured C3-like steps replayed concurrently, the filter gradient of
every replay compared with the eager one (1e-5).  argv: n (points), K (scans), rounds."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import lattice_net_amd as L
from lattice_net_amd.synthetic import lidar_cloud

torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 100
v = f = 32
cap = 100000 if n > 20000 else 30000
rng = np.random.default_rng(9)
W = torch.from_numpy((rng.standard_normal((9 * v, f)) / np.sqrt(9 * v)).astype(np.float32)).to(dev).requires_grad_(True)
scans = []
for k in range(K):
    sc = {"st": {}, "pos": torch.from_numpy(lidar_cloud(n, 40 + k)).to(dev), "vals": torch.randn((n, v), device=dev), "G": torch.randn((n, f), device=dev),
          "lat": L.Lattice(sigmas=[0.9] * 3, capacity=cap, device=dev)}

    def step(sc=sc):
        W.grad = None
        lv, _, idx, w = L.SplatLattice.apply(sc["lat"], sc["pos"], sc["vals"])
        lv = lv[:sc["lat"].nr_lattice_vertices()].requires_grad_(True)
        cv, cw = L.ConvIm2RowLattice.apply(lv, sc["lat"], W, 1)
        cv.retain_grad()  # the gradient rows the fused backward gathers stay readable after the replay
        out = L.SliceLattice.apply(cv, cw.lattice, sc["pos"], idx, w)
        out.backward(sc["G"])
        sc["st"].update(idx=idx, gw=W.grad, gv=lv.grad, cv=cv)

    step()
    torch.cuda.synchronize()
    sc["gw_ref"] = sc["st"]["gw"].detach().clone()
    sc["gv_ref"] = sc["st"]["gv"].detach().clone()
    sc["G_ref"] = sc["st"]["cv"].grad.detach().clone()
    sc["cap"] = L.CapturedStep(step, [sc["lat"]], region_indices=lambda sc=sc: sc["st"]["idx"], stream=torch.cuda.Stream(), before_capture=sc["st"].clear)
    scans.append(sc)
torch.cuda.synchronize()


def dump_check():
    """-DLN_TR_CHECK builds: element-wise re-read of the transposed fragments inside the kernel."""
    import ctypes as C
    lib = L._lib.load()
    if not hasattr(lib, "ln_debug_dump"):
        return
    buf = (C.c_int * 132)()
    lib.ln_debug_dump(buf, 132)
    if buf[0] or buf[1]:
        print("   in-kernel check: transposed-read mismatches", buf[0], " own-row read-back mismatches", buf[1])
        for sl in range(min(buf[0], 16)):
            r = [buf[4 + sl * 8 + j] & 0xffffffff for j in range(8)]
            print(f"     wg {r[0]} lane {r[1] & 255} wave {(r[1] >> 8) & 255} k {(r[1] >> 16) & 255} part {(r[1] >> 24) & 15} st {r[1] >> 28}"
                  f" elems {r[2]:08b} lds_alloc {r[3]:#x} hw_id {r[4]:#x} xcc {r[5]} expected/got {r[6] & 0xffff:#x}/{r[6] >> 16:#x} stage_off {r[7]}", flush=True)


bad = 0
for rnd in range(rounds):
    for sc in scans:
        sc["cap"].launch()
    torch.cuda.synchronize()
    for k, sc in enumerate(scans):
        rel = float((sc["st"]["gw"] - sc["gw_ref"]).abs().max() / sc["gw_ref"].abs().max())
        if rnd in (0, 3) and "G_ref" in sc:  # control: the gathered rows after a good replay
            mm = sc["gv_ref"].shape[0]
            Gd = (sc["st"]["cv"].grad[:mm] - sc["G_ref"]).abs()
            print("round", rnd, "scan", k, "rel", rel, "gradient rows off (control):", int((Gd.max(1).values > 1e-4).sum()), "max", float(Gd.max()), flush=True)
        if rel > 1e-5:
            bad += 1
            print("round", rnd, "scan", k, "rel", rel, flush=True)
            if bad <= 6:  # anatomy of the damage: which (slot, v, f) entries, and is the value gradient hit as well
                d = (sc["st"]["gw"] - sc["gw_ref"]).reshape(9, v, f)
                wrong = d.abs() > 1e-5 * float(sc["gw_ref"].abs().max())
                print("   wrong entries", int(wrong.sum()), "by f%4", [int(wrong[:, :, j::4].sum()) for j in range(4)],
                      "by f", wrong.sum((0, 1)).tolist(), "by slot", wrong.sum((1, 2)).tolist(), "by v", wrong.sum((0, 2)).tolist())
                mm = sc["gv_ref"].shape[0]
                gvd = (sc["st"]["gv"][:mm] - sc["gv_ref"]).abs()
                rows = (gvd.max(1).values > 1e-5 * float(sc["gv_ref"].abs().max())).nonzero().flatten()
                print("   value-gradient rows off:", rows.numel(), rows[:16].tolist(), "max", float(gvd.max()), flush=True)
                Gd = (sc["st"]["cv"].grad[:mm] - sc["G_ref"]).abs()
                print("   gathered gradient rows (input of the kernel) off after the replay:", int((Gd.max(1).values > 1e-4).sum()), "max", float(Gd.max()), flush=True)
                G_now = sc["st"]["cv"].grad[:mm]
                cnt = torch.bincount(sc["st"]["idx"].flatten().clamp(min=0).long(), minlength=mm)
                offrows = (Gd.max(1).values > 1e-4).nonzero().flatten()
                print("   tokens per vertex: median", int(cnt.median()), " of the wrong rows:", cnt[offrows][:24].tolist(), flush=True)
                for r in (Gd.max(1).values > 1e-4).nonzero().flatten()[:4].tolist():
                    print("     row", r, "now", [round(x, 3) for x in G_now[r].tolist()])
                    print("     row", r, "ref", [round(x, 3) for x in sc["G_ref"][r].tolist()], flush=True)
                print("   sample diff / ref:", d[wrong][:6].tolist(), sc["gw_ref"].reshape(9, v, f)[wrong][:6].tolist(), flush=True)
                dump_check()
dump_check()
print(f"n={n} K={K}: {bad} bad of {rounds * K} replays")
