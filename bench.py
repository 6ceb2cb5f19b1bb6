#!/usr/bin/env python3
"""Headline benchmark: Mpoints/s for one pass of the permutohedral-lattice hot path
{hash build + splat -> neighbour list + one lattice convolution -> slice}, forward + backward, on a
120k-point SemanticKITTI-like scan (BASELINE.json config C3: d=3, sigma 0.9, capacity 100000,
V=F=32, fp32).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; independent clouds are sharded across ranks (weak scaling, no data-path
collective).  RCCL is used only to broadcast the filter bank once and to reduce timings /
checksums.  Rank 0 prints ONE JSON line.  Inputs are resident in HBM before the timed region.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32-input MFMA peak (MI355X_MICROARCH.md)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 MFMA peak (no sparsity)

WORKLOADS = {
    # name: (n_points, val_dim, nr_filters, sigma, capacity, generator)
    "C3": dict(n=120000, v=32, f=32, sigma=0.9, capacity=100000, gen="lidar",
               desc="C3 SemanticKITTI-like scan: 120k pts, d=3, sigma 0.9, capacity 100k, V=F=32, splat->conv->slice fwd+bwd"),
    "C1": dict(n=1000, v=4, f=4, sigma=0.2, capacity=60000, gen="cube", desc="C1 1k-pt cube (parity-size case)"),
    "C2": dict(n=2500, v=32, f=32, sigma=0.05, capacity=60000, gen="box", desc="C2 ShapeNet-like surface cloud: 2.5k pts, sigma 0.05, capacity 60k, V=F=32"),
    "C4": dict(n=200000, v=32, f=32, sigma=0.08, capacity=5000000, gen="planes",
               desc="C4 ScanNet-like scene: 200k pts on planes, sigma 0.08, capacity 5M, V=F=32"),
    "C5": dict(n=480000, v=64, f=64, sigma=0.9, capacity=400000, gen="lidar4", half=True,
               desc="C5 4 aggregated scans: 480k pts, capacity 400k, V=F=64, fp16 features / fp32 accumulate in the convolution"),
}


def make_cloud(kind: str, n: int, seed: int) -> np.ndarray:
    from lattice_net_amd import synthetic
    if kind == "lidar":
        return synthetic.lidar_cloud(n, seed)
    if kind == "box":
        return synthetic.box_surface_cloud(n, seed)
    if kind == "planes":
        return synthetic.planes_cloud(n, seed)
    if kind == "lidar4":  # four scans taken 6 m apart along x, aggregated
        parts = []
        for k in range(4):
            c = synthetic.lidar_cloud(n // 4, seed * 4 + k)
            c[:, 0] += 6.0 * k
            parts.append(c)
        return np.ascontiguousarray(np.concatenate(parts, 0))
    return synthetic.cube_cloud(n, seed)


def algorithmic_work(kernel: str, n: int, m: int, d: int, v: int, f: int, e: int, cap: int = 0):
    """(bound, unit, amount per launch) — SURVEY.md §8d per-unit figures x units per launch."""
    if kernel == "k_conv_mfma":
        return "mfma", "TFLOP/s", 2.0 * m * e * v * f
    if kernel == "k_conv_mfma_f16":
        return "mfma_f16", "TFLOP/s", 2.0 * m * e * v * f
    if kernel == "k_grad_filter_mfma":
        return "mfma", "TFLOP/s", 2.0 * m * e * v * f
    if kernel == "k_conv_backward_fused":  # value gradient + filter gradient of the convolution in one launch (fp32 matrix cores)
        return "mfma", "TFLOP/s", 4.0 * m * e * v * f
    if kernel in ("k_scatter_point_rows", "k_csr_reduce_segments"):  # splat accumulate / slice backward: read rows+idx+w, write vertex rows
        return "hbm", "GB/s", n * (4.0 * v + 8.0 * (d + 1)) + m * 4.0 * v
    if kernel == "k_reduce_and_neighbours":  # splat accumulate + same-level neighbour list in one launch
        return "hbm", "GB/s", n * (4.0 * v + 8.0 * (d + 1)) + m * 4.0 * v + m * (4.0 * d + 4.0 * e)
    if kernel == "k_slice_forward":
        return "hbm", "GB/s", n * (8.0 * (d + 1) + 4.0 * v) + m * 4.0 * v
    if kernel == "k_insert_points":  # read positions, write idx + w, write keys once
        return "hbm", "GB/s", n * (4.0 * d + 8.0 * (d + 1)) + m * 4.0 * d
    if kernel == "k_point_keys":  # read positions; write w and one (token, packed key) entry per simplex vertex
        return "hbm", "GB/s", n * (4.0 * d + (4.0 + 12.0) * (d + 1))
    if kernel == "k_bucket_build":  # read the entries; write token->slot, the slot CSR and the slot range (keys, first token, start)
        return "hbm", "GB/s", n * (12.0 + 4.0 + 4.0) * (d + 1) + cap * 16.0
    if kernel == "k_neighbours":
        return "hbm", "GB/s", m * (4.0 * d + 4.0 * e)
    raise ValueError(f"no algorithmic model for kernel {kernel}")


def pmc_traffic(kernel: str):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r<round>_pmc_traffic.json, newest round), or None."""
    for rnd in (2, 1):
        path = os.path.join(ROOT, "profiles", f"r{rnd}_pmc_traffic.json")
        try:
            with open(path) as f:
                table = json.load(f)
                # (the launch name of the ABI's timers covers both forms of the fused backward; the profile lists the kernel symbol)
                alias = {"k_conv_mfma": "k_conv_forward_b3"}.get(kernel, kernel + "_b3")  # launch name -> kernel symbol of the C3 step
                hit = (table.get(kernel) or table.get(alias) or {}).get("traffic_bytes")
            if hit is not None:
                return hit
        except OSError:
            continue
    return None


def cpu_baseline(cfg, seconds: float):
    """Pure-PyTorch CPU fallback of the same op chain (oracle/torch_fallback.py), all host cores."""
    from oracle import torch_fallback as TF
    n, v, f = cfg["n"], cfg["v"], cfg["f"]
    pos = torch.from_numpy(make_cloud(cfg["gen"], n, 0))
    rng = np.random.default_rng(0)
    vals = torch.from_numpy(rng.standard_normal((n, v)).astype(np.float32))
    W = torch.from_numpy((rng.standard_normal((9 * v, f)) / np.sqrt(9 * v)).astype(np.float32))
    G = torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32))
    threads = torch.get_num_threads()
    TF.hot_path_step(pos, vals, W, G, cfg["sigma"])  # warm-up
    times = []
    t_end = time.perf_counter() + seconds
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 200):
        t0 = time.perf_counter()
        TF.hot_path_step(pos, vals, W, G, cfg["sigma"])
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": n / med / 1e6, "unit": "Mpoints/s", "cores": threads, "kind": "port",
            "sample": f"{len(times)} full-size steps of the same workload (median {med * 1e3:.1f} ms/step), "
                      f"pure-PyTorch CPU fallback oracle/torch_fallback.py, torch threads={threads}, os.cpu_count()={os.cpu_count()}"}


UNET_CFG = """
model: {
    positions_mode: "xyz"
    values_mode: "none"
    pointnet_layers: [16,32]
    pointnet_start_nr_channels: 32
    nr_downsamples: 2
    nr_blocks_down_stage: [1,1,1]
    nr_blocks_bottleneck: 1
    nr_blocks_up_stage: [1,1,1]
    nr_levels_down_with_normal_resnet: 3
    nr_levels_up_with_normal_resnet: 3
    compression_factor: 1.0
    dropout_last_layer: 0.0
}
lattice_gpu: {
    hash_table_capacity: 100000
    nr_sigmas: 1
    sigma_0: "0.9 3"
}
"""


def full_unet_step(dev, n: int, steps: int = 10, warmup: int = 3, classes: int = 20):
    """Secondary number (BASELINE.json configs[2]: "SemanticKITTI single scan, full U-net with coarsen/finefy"): one training step
    (forward + NLL + backward + AdamW) of the LNN assembled on this backend with the model shape of the reference's
    lnn_train_semantic_kitti.cfg, on the same synthetic scan.  Not the headline metric."""
    import gc
    import tempfile
    from lattice_net_amd import Lattice, ModelParams, synthetic
    from lattice_net_amd.losses import nll_loss_gather
    from lattice_net_amd.models import LNN
    with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as fcfg:
        fcfg.write(UNET_CFG)
        path = fcfg.name
    torch.manual_seed(0)
    mp = ModelParams.create(path)
    lattice = Lattice.create(path, "lattice")
    os.unlink(path)  # both readers are done with the temporary cfg
    net = LNN(classes, mp, device=dev)
    pos = torch.from_numpy(synthetic.lidar_cloud(n, 0)).to(dev)
    vals = torch.zeros((n, 1), device=dev)
    target = torch.from_numpy(np.random.default_rng(0).integers(0, classes, n)).to(dev)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-4, amsgrad=True, fused=True)  # ln_train.py:165, single-launch update

    def step():
        logsoftmax, _ = net(lattice, pos, vals)
        loss = nll_loss_gather(logsoftmax, target)
        opt.zero_grad()
        loss.backward()
        opt.step()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    gc_was_on = gc.isenabled()
    gc.collect()
    gc.disable()  # generation-2 passes over the live module / autograd objects cost milliseconds per step otherwise
    try:
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    finally:
        if gc_was_on:
            gc.enable()
    out = {"what": "LNN training step (forward + NLL + backward + AdamW), reference SemanticKITTI model shape, same 120k-point scan",
           "ms_per_step": round(dt * 1e3, 3), "mpoints_per_s": round(n / dt / 1e6, 2), "parameters": sum(p.numel() for p in net.parameters()),
           "steps": steps}
    # The same step with forward + loss + backward captured as ONE hipGraph (lattice_net_amd.CapturedNetworkStep: static row bounds
    # on every lattice level, GroupNorm over the device-side vertex count), timed by tools/bench_lnn.py in a CHILD
    # process: a secondary number must not be able to take the headline down with it.
    try:
        import subprocess
        import sys
        tool = os.path.join(ROOT, "tools", "bench_lnn.py")
        r = subprocess.run([sys.executable, tool, "--config", "kitti", "--n", str(n), "--graph", "--steps", str(2 * steps), "--warmup", str(warmup)],
                           capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("LNN[")]
        if r.returncode != 0 or not line:
            raise RuntimeError((r.stderr or r.stdout)[-300:])
        ms = float(line[-1].split(":")[1].split("ms")[0])
        out["graph"] = {"what": "forward + NLL + backward as one hipGraph replay, AdamW outside (tools/bench_lnn.py --graph, child process)",
                        "ms_per_step": round(ms, 3), "mpoints_per_s": round(n / ms / 1e3, 2)}
    except Exception as e:
        out["graph"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000,
                    help="timed steps (scans).  The timed region has a fixed cost of ~0.3 ms (first graph launch, drain of the last scans in "
                         "flight, the final synchronize): 0.0905 ms per scan at 2000+ steps, 0.097 at 50, 0.107 at 20")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--roofline-kernel", default="k_conv_backward_fused",
                    help="dominant kernel (largest share of GPU time in profiles/r2_kernel_stats.csv): its launches are timed live "
                         "with HIP events during the timed region")
    ap.add_argument("--extra-kernels", default="k_reduce_and_neighbours,k_csr_reduce_segments,k_conv_mfma,k_bucket_build,k_point_keys,k_slice_forward",
                    help="kernels timed the same way in extra untimed steps AFTER the timed region (reported under roofline_others)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the CPU baseline leg (0 disables)")
    ap.add_argument("--full-unet", type=int, default=1, help="0 = skip the secondary whole-network timing (rank 0, one GPU, workload C3)")
    ap.add_argument("--autograd-threads", type=int, default=0, help="1 = leave torch's per-device autograd worker thread on")
    ap.add_argument("--mode", default="graph", choices=["graph", "eager"],
                    help="graph (default): the whole step (forward + backward, ~10 launches) is captured ONCE into a hipGraph with the "
                         "lattice in static-rows mode and every timed step is one graph replay; eager: one Python autograd pass per step")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="graph mode: independent scans in flight per GPU (own cloud, lattice, hipGraph, stream each); the kernels of one "
                         "scan are latency-bound chains at ~1 workgroup per CU, a second scan fills the idle slots.  1 = strictly one "
                         "scan after the other")
    ap.add_argument("--regions", type=int, default=int(os.environ.get("LN_BENCH_REGIONS", "1")),
                    help="1 (default, graph mode): calibrate kd region planes on the first eager step of every scan (equal token load per "
                         "region) so that the scatter kernels walk one compact region of the lattice per XCD: -40 %% L2-miss traffic "
                         "on the two segment reduces, +5 %% throughput with two scans in flight")
    ap.add_argument("--row-slack", type=float, default=0.06,
                    help="graph mode: static row bound = vertex count of the calibration step x (1 + slack), rounded up to 256")
    args = ap.parse_args()

    from lattice_net_amd import sharding
    world, rank, local_rank = sharding.env_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the lattice backend has no CPU path)")
    # LATTICE_BENCH_SHARE_GPU=1 (testing aid for a one-GPU box): every rank runs on GPU 0 and the ranks talk over gloo,
    # so the multi-rank control flow can be exercised with real kernels; never a measurement
    share_gpu = bool(os.environ.get("LATTICE_BENCH_SHARE_GPU"))
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = sharding.init("gloo" if share_gpu else "nccl", dev)  # RCCL over xGMI; None for a single rank
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world} rank(s)", file=sys.stderr)

    import lattice_net_amd as L
    lib = L.load_library()
    if not args.autograd_threads:
        # run backward on the calling thread: the hand-off to torch's per-device autograd worker costs tens of
        # microseconds per step, which is comparable to the whole GPU time of this path
        torch.autograd.set_multithreading_enabled(False)

    cfg = WORKLOADS[args.workload]
    n, v, f, sigma, cap = cfg["n"], cfg["v"], cfg["f"], cfg["sigma"], cfg["capacity"]
    d, e = 3, 9
    half = bool(cfg.get("half"))
    # independent clouds per rank (weak scaling); parameters broadcast from rank 0 over RCCL
    bound_w = float(np.sqrt(3.0) * np.sqrt(2.0) / np.sqrt(f))  # kaiming-uniform fan_out (lattice_modules.py:202-207)
    W = ((torch.rand((e * v, f), device=dev) * 2 - 1) * bound_w)
    sharding.broadcast_parameters(dist, [W], src=0)
    W.requires_grad_(True)
    in_flight = max(1, args.in_flight) if args.mode == "graph" else 1

    class CloudSet:
        """One scan in flight: its own cloud, lattice, captured step and stream."""

        def __init__(self, k):
            rng = np.random.default_rng(rank + 1000 * k)
            self.pos = torch.from_numpy(make_cloud(cfg["gen"], n, sharding.cloud_seed(rank, k))).to(dev)
            self.vals = torch.from_numpy(rng.standard_normal((n, v)).astype(np.float32)).to(dev)
            self.G = torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).to(dev)
            if half:
                self.vals, self.G = self.vals.half(), self.G.half()
            self.lat = L.Lattice(sigmas=[sigma] * d, capacity=cap, device=dev)
            self.state = {}
            self.graph = None
            self.stream = torch.cuda.Stream() if k > 0 else None  # set 0 stays on the current stream

        def step(self):
            W.grad = None
            lv, wrap, idx, w = L.SplatLattice.apply(self.lat, self.pos, self.vals)  # clear + hash build + accumulate
            m = self.lat.nr_lattice_vertices()                                  # eager: the path's one host readback
            lv = lv[:m].requires_grad_(True)
            if half:  # fp16 feature path: fp16 point features, lattice values and filter bank; fp32 accumulation everywhere
                cv, cwrap = L.ConvIm2RowLattice.apply(lv.half(), self.lat, W.half(), 1)
            else:
                cv, cwrap = L.ConvIm2RowLattice.apply(lv, self.lat, W, 1)      # neighbour list + gather-GEMM
            out = L.SliceLattice.apply(cv, cwrap.lattice, self.pos, idx, w)    # slice
            out.backward(self.G)                                                # slice bwd, conv bwd (values + filter)
            self.state.update(m=m, out=out, gv=lv.grad, gw=W.grad, idx=idx)

        def capture(self):
            """Eager reference result, then lattice_net_amd.capture.CapturedStep: static row bound + kd region planes calibrated
            on this scan, warm-up on a side stream, the whole step captured into one hipGraph."""
            from lattice_net_amd.capture import CapturedStep
            self.step()
            torch.cuda.synchronize()
            self.m_real = self.state["m"]
            self.eager_out = self.state["out"].detach().clone()
            self.eager_gw = self.state["gw"].detach().clone()
            self.cap = CapturedStep(self.step, [self.lat], row_slack=args.row_slack, regions=bool(args.regions),
                                    region_indices=lambda: self.state["idx"], stream=self.stream, before_capture=self.state.clear)
            assert self.cap.vertices[0] == self.m_real
            self.graph = self.cap.graph

        def launch(self):
            if self.graph is None:
                self.step()
            else:
                self.cap.launch()

        def check(self):
            """Replayed build within its bounds, replayed results equal to the eager step's (1e-5 relative)."""
            nr = self.cap.check()[0]  # raises if the replayed build overflowed its row bound or a bucket
            assert nr == self.m_real, (nr, self.m_real)
            scale = float(self.eager_out.abs().max())
            err = {"out_max_rel": float((self.state["out"].detach() - self.eager_out).abs().max()) / max(scale, 1e-30),
                   "grad_filter_max_rel": float((self.state["gw"] - self.eager_gw).abs().max()) / max(float(self.eager_gw.abs().max()), 1e-30)}
            # fp32 path: 1e-5 (BASELINE.json); fp16 feature path (C5): outputs are rounded to fp16 and the order of the
            # hot-vertex atomics differs from run to run, 2e-3 as in tests/test_gpu_surface.py
            if max(err.values()) > (2e-3 if half else 1e-5):
                raise SystemExit(f"[bench] graph replay differs from the eager step: {err}")
            return err

    def barrier():
        sharding.barrier(dist)
        torch.cuda.synchronize()

    sets = [CloudSet(k) for k in range(in_flight)]
    graph_err = None
    if args.mode == "graph":
        for cs in sets:
            cs.capture()
    else:
        sets[0].m_real = None
    if args.mode == "graph":  # validation replays (untimed): every captured scan several times, then compared with its eager result
        for i in range(8 * in_flight):
            sets[i % in_flight].launch()
    for i in range(max(args.warmup, in_flight)):  # the W warm-up steps
        sets[i % in_flight].launch()
    barrier()
    if args.mode == "graph":
        errs = [cs.check() for cs in sets]
        graph_err = {k: max(e[k] for e in errs) for k in errs[0]}
    else:
        sets[0].m_real = sets[0].state["m"]
    prof_name = args.roofline_kernel.encode()
    armed = args.mode == "eager" and lib.ln_profile_begin(prof_name, 8 * args.steps + 8) == 0
    t0 = time.perf_counter()
    for i in range(args.steps):  # K steps = K scans, issued round-robin over the scans in flight
        sets[i % in_flight].launch()
    barrier()
    elapsed = time.perf_counter() - t0
    total_ms, launches = C.c_double(0.0), C.c_int(0)
    if armed:
        lib.ln_profile_end(C.byref(total_ms), C.byref(launches))
    single = None
    if args.mode == "graph":
        for cs in sets:
            cs.check()
        if in_flight > 1:  # the same captured step, one scan at a time (latency of a scan = what a batch-1 training loop sees)
            side_steps = min(args.steps, 200)
            t1 = time.perf_counter()
            for _ in range(side_steps):
                sets[0].launch()
            torch.cuda.synchronize()
            dt1 = (time.perf_counter() - t1) / side_steps
            single = {"what": "one scan in flight (same hipGraph, one stream)", "us_per_step": round(dt1 * 1e6, 1),
                      "mpoints_per_s": round(n / dt1 / 1e6, 1)}
    checksum_local = sum(float(cs.state["out"].double().abs().sum().item()) for cs in sets)
    m_all = [cs.m_real for cs in sets]

    # everything below (per-kernel and per-stage timings) runs eager steps of scan 0
    cs0 = sets[0]
    lat, pos, vals, G, state, step = cs0.lat, cs0.pos, cs0.vals, cs0.G, cs0.state, cs0.step
    m_real = cs0.m_real
    if args.mode == "graph":
        # Per-kernel HIP-event timing needs host-side event records between the launches, which a graph replay has no
        # room for: the roofline kernel is timed in eager steps of the same workload right after the timed region.
        lat.set_static_rows(None)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        roofline_steps = min(args.steps, 200)
        if lib.ln_profile_begin(prof_name, 8 * roofline_steps + 8) == 0:
            for _ in range(roofline_steps):
                step()
            torch.cuda.synchronize()
            lib.ln_profile_end(C.byref(total_ms), C.byref(launches))
            armed = True

    m = m_real
    checksum = checksum_local
    max_elapsed = sharding.max_over_ranks(dist, elapsed, dev)
    checksum = sharding.gather_sum(dist, checksum, dev)

    def roofline_entry(kernel, total_ms_v, launches_v):
        bound_kind, unit, amount = algorithmic_work(kernel, n, m, d, v, f, e, cap)
        avg_s = total_ms_v / launches_v / 1e3
        if bound_kind == "hbm":
            achieved, peak = amount / avg_s / 1e9, HBM_PEAK_GBS
        elif bound_kind == "mfma_f16":
            achieved, peak, bound_kind = amount / avg_s / 1e12, MFMA_F16_PEAK_TFLOPS, "mfma"
        else:
            achieved, peak = amount / avg_s / 1e12, MFMA_F32_PEAK_TFLOPS
        return {"bound": bound_kind, "achieved": round(achieved, 3), "peak": peak, "unit": unit, "frac": round(achieved / peak, 4),
                "traffic": pmc_traffic(kernel) if args.workload == "C3" else None, "kernel": kernel,
                "avg_us": round(avg_s * 1e6, 2), "launches_timed": launches_v}

    # other kernels of the path, timed the same way in a few extra steps outside the timed region
    others = []
    if rank == 0:
        for name in [k for k in args.extra_kernels.split(",") if k and k != args.roofline_kernel]:
            if lib.ln_profile_begin(name.encode(), 64) != 0:
                continue
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            tms, cnt = C.c_double(0.0), C.c_int(0)
            lib.ln_profile_end(C.byref(tms), C.byref(cnt))
            if cnt.value > 0:
                try:
                    others.append(roofline_entry(name, tms.value, cnt.value))
                except ValueError:
                    pass
    barrier()

    # Per-stage GPU time ("report each stage separately and splat+slice alone"), measured with events on
    # the launch stream in extra, untimed steps; the HBM fraction of splat+slice uses the algorithmic bytes of 8d.
    stages = None
    if rank == 0:
        names = ["splat", "conv", "slice", "backward"]
        acc_ms = dict.fromkeys(names, 0.0)
        reps = 10
        for _ in range(reps):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            W.grad = None
            ev[0].record()
            lv, wrap, idx, w = L.SplatLattice.apply(lat, pos, vals)
            ev[1].record()
            mm = lat.nr_lattice_vertices()
            lv = lv[:mm].requires_grad_(True)
            if half:
                cv, cwrap = L.ConvIm2RowLattice.apply(lv.half(), lat, W.half(), 1)
            else:
                cv, cwrap = L.ConvIm2RowLattice.apply(lv, lat, W, 1)
            ev[2].record()
            out = L.SliceLattice.apply(cv, cwrap.lattice, pos, idx, w)
            ev[3].record()
            out.backward(G)
            ev[4].record()
            torch.cuda.synchronize()
            for k, nm in enumerate(names):
                acc_ms[nm] += ev[k].elapsed_time(ev[k + 1])
        us = {nm: acc_ms[nm] / reps * 1e3 for nm in names}
        splat_bytes = n * (4.0 * d + 4.0 * v + 8.0 * (d + 1)) + m * (4.0 * d + 4.0 * v)
        slice_bytes = n * (8.0 * (d + 1) + 4.0 * v) + m * 4.0 * v
        ss_us = us["splat"] + us["slice"]
        stages = {"us": {k: round(x, 1) for k, x in us.items()},
                  "note": "event-to-event on the launch stream; `splat` = clear + hash build + accumulate (+ the neighbour prefetch "
                          "issued behind it), `conv` includes the wait for the vertex-count readback",
                  "splat_plus_slice": {"us": round(ss_us, 1), "algorithmic_bytes": int(splat_bytes + slice_bytes),
                                       "achieved_GBs": round((splat_bytes + slice_bytes) / ss_us / 1e3, 1),
                                       "frac_of_hbm_peak": round((splat_bytes + slice_bytes) / ss_us / 1e3 / HBM_PEAK_GBS, 4)}}
        if args.mode == "graph" and not half:
            # the same two stages in the benchmark's execution mode: splat -> slice (forward) captured per scan, scans in flight
            try:
                from lattice_net_amd.capture import CapturedStep
                chains = []
                for k in range(in_flight):
                    c = {"lat": L.Lattice(sigmas=[sigma] * d, capacity=cap, device=dev), "st": {}, "pos": sets[k].pos, "vals": sets[k].vals}

                    def chain(c=c):
                        with torch.no_grad():
                            lv2, _, idx2, w2 = L.SplatLattice.apply(c["lat"], c["pos"], c["vals"])
                            m2 = c["lat"].nr_lattice_vertices()
                            c["st"].update(idx=idx2, out=L.SliceLattice.apply(lv2[:m2], c["lat"], c["pos"], idx2, w2))

                    chain()
                    torch.cuda.synchronize()
                    ref2 = c["st"]["out"].detach().clone()
                    c["cap"] = CapturedStep(chain, [c["lat"]], row_slack=args.row_slack, regions=bool(args.regions),
                                            region_indices=lambda c=c: c["st"]["idx"], stream=torch.cuda.Stream(), before_capture=c["st"].clear)
                    c["cap"].launch()
                    torch.cuda.synchronize()
                    err2 = float((c["st"]["out"] - ref2).abs().max()) / max(float(ref2.abs().max()), 1e-30)
                    if err2 > 1e-5:
                        raise RuntimeError(f"replayed splat -> slice differs from the eager one: {err2}")
                    chains.append(c)
                reps2 = 600
                for i in range(30):
                    chains[i % in_flight]["cap"].launch()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for i in range(reps2):
                    chains[i % in_flight]["cap"].launch()
                torch.cuda.synchronize()
                us2 = (time.perf_counter() - t2) / reps2 * 1e6
                for c in chains:
                    c["cap"].check()
                stages["splat_plus_slice_in_flight"] = {
                    "what": f"splat -> slice (forward) as one hipGraph per scan, {in_flight} scans in flight, {reps2} scans timed",
                    "us_per_scan": round(us2, 1), "algorithmic_bytes": int(splat_bytes + slice_bytes),
                    "achieved_GBs": round((splat_bytes + slice_bytes) / us2 / 1e3, 1),
                    "frac_of_hbm_peak": round((splat_bytes + slice_bytes) / us2 / 1e3 / HBM_PEAK_GBS, 4)}
            except Exception as ex:  # secondary figure: never endangers the bench line
                stages["splat_plus_slice_in_flight"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    line = None
    if rank == 0:
        value = n * world * args.steps / max_elapsed / 1e6
        roofline = None
        if armed and launches.value > 0:
            roofline = roofline_entry(args.roofline_kernel, total_ms.value, launches.value)
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            cpu = cpu_baseline(cfg, args.cpu_seconds)
        unet = None
        if world == 1 and args.full_unet and args.workload == "C3":
            try:
                unet = full_unet_step(dev, n)
            except Exception as exc:  # secondary number: never take the headline line down with it
                unet = {"error": f"{type(exc).__name__}: {exc}"}
        line = {
            "metric": "Mpoints/sec splat+conv+slice fwd+bwd on 120k-pt SemanticKITTI scan",
            "value": round(value, 3), "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(max_elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 features / f32 accumulate" if half else "f32", "data": "synthetic",
            "config": {"workload": cfg["desc"] + (f"; {in_flight} independent scans in flight per GPU, K steps = K scans" if in_flight > 1 else ""),
                       "points_per_gpu": n, "vertices": m, "val_dim": v, "nr_filters": f,
                       "sharding": f"{world} independent cloud(s), one per GPU", "checksum": round(checksum, 3),
                       "execution": (f"one hipGraph replay per scan (whole forward + backward captured once, static row bound); "
                                     f"{in_flight} independent scan(s) in flight per GPU, each with its own lattice, graph and stream; "
                                     f"K steps = K scans" if args.mode == "graph" else "eager: Python autograd pass per step"),
                       "scans_in_flight": in_flight, "kd_regions": bool(args.regions and args.mode == "graph"),
                       "vertices_per_scan": m_all, "graph_vs_eager": graph_err,
                       "one_scan_in_flight": single},
            "roofline": roofline, "roofline_others": others, "stages": stages, "full_unet": unet, "cpu_baseline": cpu,
        }
    try:  # RCCL prints a version banner through C stdio, which a pipe buffers until exit: every rank flushes it now
        C.CDLL(None).fflush(None)
    except OSError:
        pass
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)  # the last line on stdout


if __name__ == "__main__":
    main()
