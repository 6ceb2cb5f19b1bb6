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
    "C2": dict(n=2500, v=32, f=32, sigma=0.05, capacity=60000, gen="box", batch=16,
               desc="C2 ShapeNet-like surface cloud: 2.5k pts, sigma 0.05, capacity 60k, V=F=32"),
    "C4": dict(n=200000, v=32, f=32, sigma=0.08, capacity=5000000, gen="planes",
               desc="C4 ScanNet-like scene: 200k pts on planes, sigma 0.08, capacity 5M, V=F=32"),
    "C4probe": dict(n=200000, v=32, f=32, sigma=0.08, capacity=1600000, gen="planes",
                    desc="probe: C4 hashed into 2 x tokens = 1.6M slots instead of the cfg's 5M"),
    "C3x4": dict(n=480000, v=32, f=32, sigma=0.9, capacity=400000, gen="lidar4far",
                 desc="probe: 4 C3 scans 200 m apart processed as ONE cloud (what a batched launch over 4 independent scans would cost)"),
    "C5": dict(n=480000, v=64, f=64, sigma=0.9, capacity=400000, gen="lidar4", half=True,
               desc="C5 4 aggregated scans: 480k pts, capacity 400k, V=F=64, fp16 features / fp32 accumulate in the convolution"),
}

FINAL_LINE_LIMIT = 4096  # bytes: the driver keeps only the tail of stdout, so the LAST line must stay far below that


def _pick(src, keys):
    return {k: src[k] for k in keys if isinstance(src, dict) and k in src and src[k] is not None}


def compact_line(full: dict, details_file: str = "bench_details.json") -> dict:
    """The ONE line the driver parses: headline, config, roofline of the dominant kernel group, the splat+slice fractions and
    the CPU baseline.  Everything else of `full` (per-operator table, other kernels, stages, latency, whole-network step,
    prose) lives in `details_file` and on an earlier stdout line prefixed `DETAILS `.  Optional keys are dropped, last first,
    until the line fits FINAL_LINE_LIMIT; the contract keys never are."""
    cfg = full.get("config") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = _pick(cfg, ("workload", "points_per_gpu", "vertices", "val_dim", "nr_filters", "scans_in_flight",
                                 "clouds_per_scan_pool", "clouds_per_step", "sharding", "checksum", "prewarm", "slot_order"))
    if isinstance(line["config"].get("workload"), str):
        line["config"]["workload"] = line["config"]["workload"][:240]
    line["roofline"] = (_pick(full.get("roofline"), ("bound", "kernel", "avg_us", "achieved", "peak", "unit", "frac", "traffic"))
                        if full.get("roofline") else None)
    if line["roofline"] is not None:
        line["roofline"].setdefault("traffic", None)
    cpu = full.get("cpu_baseline")
    line["cpu_baseline"] = (_pick(cpu, ("value", "unit", "cores", "kind", "sample")) if cpu else None)
    if line["cpu_baseline"] and isinstance(line["cpu_baseline"].get("sample"), str):
        line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:160]
    optional = []  # (key, value), most important first
    optional.append(("value_8d_one_pass_mpoints_per_s", full.get("value_8d_one_pass_mpoints_per_s")))
    optional.append(("value_no_prewarm", full.get("value_no_prewarm")))
    rc = full.get("roofline_chain")
    optional.append(("roofline_chain", _pick(rc, ("algorithmic_bytes", "traffic_bytes", "frac_one_pass", "frac_in_flight")) if rc else None))
    st = full.get("stages") or {}
    ss = {}
    for key, short in (("splat_plus_slice", "one_scan"), ("splat_plus_slice_in_flight", "in_flight")):
        if isinstance(st.get(key), dict) and "frac_of_hbm_peak" in st[key]:
            ss[short] = st[key]["frac_of_hbm_peak"]
    optional.append(("splat_plus_slice_frac_of_hbm_peak", ss or None))
    c1 = full.get("cpu_baseline_1thread")
    optional.append(("cpu_baseline_1thread", _pick(c1, ("value", "unit", "cores")) if c1 else None))
    lat = full.get("latency") or {}
    optional.append(("latency_us", _pick(lat, ("us_per_scan_median", "eager_us_per_scan")) or None))
    un = full.get("full_unet") or {}
    unet = _pick(un, ("ms_per_step", "algorithmic_floor_ms"))
    if isinstance(un.get("graph"), dict) and "ms_per_step" in un["graph"]:
        unet["graph_ms_per_step"] = un["graph"]["ms_per_step"]
    optional.append(("full_unet_ms", unet or None))
    for k in ("ms_per_step_min_over_ranks", "ms_per_step_max_over_ranks"):
        optional.append((k, full.get(k)))
    cc = full.get("hbm_copy_ceiling") or {}
    optional.append(("hbm_copy_ceiling_GBs", cc.get("GBs")))
    optional.append(("details_file", details_file))
    for k, val in optional:
        if val is not None:
            line[k] = val
    dropped = []
    for k, _ in reversed(optional[:-1]):
        if len(json.dumps(line)) < FINAL_LINE_LIMIT - 256:
            break
        if k in line:
            dropped.append(k)
            del line[k]
    if len(json.dumps(line)) >= FINAL_LINE_LIMIT - 256:  # still too long: only the two free-text fields can be the cause
        line["config"]["workload"] = str(line["config"].get("workload", ""))[:80]
        if line.get("cpu_baseline"):
            line["cpu_baseline"].pop("sample", None)
    if dropped:
        line["dropped_for_length"] = dropped
    return line


def emit(full: dict, details_path: str | None = None) -> dict:
    """Write `full` to bench_details.json, print it on a `DETAILS ` line, then print the compact line LAST."""
    details_path = details_path or os.path.join(ROOT, "bench_details.json")
    try:
        with open(details_path, "w") as fh:
            json.dump(full, fh, indent=1)
    except OSError as ex:  # a read-only tree must not cost the headline
        print(f"[bench] could not write {details_path}: {ex}", file=sys.stderr)
    line = compact_line(full, os.path.basename(details_path))
    text = json.dumps(line)
    assert len(text) < FINAL_LINE_LIMIT, len(text)
    print("DETAILS " + json.dumps(full), flush=True)
    print(text, flush=True)  # the last line on stdout
    return line


def make_cloud(kind: str, n: int, seed: int) -> np.ndarray:
    from lattice_net_amd import synthetic
    if kind == "lidar":
        return synthetic.lidar_cloud(n, seed)
    if kind == "box":
        return synthetic.box_surface_cloud(n, seed)
    if kind == "planes":
        return synthetic.planes_cloud(n, seed)
    if kind == "lidar4far":  # four scans so far apart that they share no lattice vertex
        parts = []
        for k in range(4):
            c = synthetic.lidar_cloud(n // 4, seed * 4 + k)
            c[:, 0] += 200.0 * k
            parts.append(c)
        return np.ascontiguousarray(np.concatenate(parts, 0))
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
    if kernel == "hash_build":  # k_point_keys + k_bucket_rows: read positions, write idx + w, write keys once (8d "splat fwd" minus the values)
        return "hbm", "GB/s", n * (4.0 * d + 8.0 * (d + 1)) + m * 4.0 * d
    if kernel == "k_conv_mfma":  # HBM-bound below 64 channels, MFMA-bound from 64 on (MFMA figures ride along: mfma_work)
        if v >= 64:
            return "mfma", "TFLOP/s", 2.0 * m * e * v * f
        return "hbm", "GB/s", m * 4.0 * v + m * 4.0 * e + 4.0 * e * v * f + m * 4.0 * f
    if kernel == "k_conv_mfma_f16":
        return "mfma_f16", "TFLOP/s", 2.0 * m * e * v * f
    if kernel == "k_grad_filter_mfma":
        return "mfma", "TFLOP/s", 2.0 * m * e * v * f
    if kernel == "k_conv_backward_fused":  # value gradient + filter gradient of the convolution in one launch
        if v >= 64:
            return "mfma", "TFLOP/s", 4.0 * m * e * v * f
        return "hbm", "GB/s", (m * 4.0 * f + m * 4.0 * e + 4.0 * e * v * f + m * 4.0 * v) + (m * 4.0 * v + m * 4.0 * f + m * 4.0 * e + 4.0 * e * v * f)
    if kernel in ("k_scatter_point_rows", "k_csr_reduce_segments"):  # splat accumulate / slice backward: read rows+idx+w, write vertex rows
        return "hbm", "GB/s", n * (4.0 * v + 8.0 * (d + 1)) + m * 4.0 * v
    if kernel == "k_reduce_and_neighbours":  # splat accumulate + same-level neighbour list in one launch
        return "hbm", "GB/s", n * (4.0 * v + 8.0 * (d + 1)) + m * 4.0 * v + m * (4.0 * d + 4.0 * e)
    if kernel == "k_slice_forward":
        return "hbm", "GB/s", n * (8.0 * (d + 1) + 4.0 * v) + m * 4.0 * v
    if kernel == "k_insert_points":  # read positions, write idx + w, write keys once
        return "hbm", "GB/s", n * (4.0 * d + 8.0 * (d + 1)) + m * 4.0 * d
    if kernel == "k_neighbours":
        return "hbm", "GB/s", m * (4.0 * d + 4.0 * e)
    raise ValueError(f"no algorithmic model for kernel {kernel}")


def mfma_work(kernel: str, m: int, v: int, f: int, e: int, half: bool):
    """(fp32-equivalent flop per launch, bf16 products executed per fp32 product) of the dense launches, or None."""
    b3 = os.environ.get("LN_CONV_EXACT_F32", "0") != "1" and v % 32 == 0 and f % 16 == 0 and not half
    if kernel in ("k_conv_mfma", "k_conv_mfma_f16"):
        return 2.0 * m * e * v * f, (6.0 if b3 else 1.0)
    if kernel == "k_conv_backward_fused":
        return 4.0 * m * e * v * f, (6.0 if b3 else 1.0)
    if kernel == "k_grad_filter_mfma":
        return 2.0 * m * e * v * f, (6.0 if (b3 and f % 32 == 0 and m >= 4096) else 1.0)
    return None


KERNEL_GROUPS = {"hash_build": ["k_point_keys", "k_bucket_rows"]}  # launches that only make sense together


def pmc_traffic(kernel: str):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r<round>_pmc_traffic.json, newest round), or None."""
    for rnd in (6, 5, 4, 3, 2, 1):
        path = os.path.join(ROOT, "profiles", f"r{rnd}_pmc_traffic.json")
        try:
            with open(path) as f:
                table = json.load(f)
        except OSError:
            continue
        total = 0
        for k in KERNEL_GROUPS.get(kernel, [kernel]):
            # (the launch name of the ABI's timers covers both forms of the fused backward; the profile lists the kernel symbol)
            alias = {"k_conv_mfma": "k_conv_forward_b3"}.get(k, k + "_b3")
            hit = (table.get(k) or table.get(alias) or {}).get("traffic_bytes")
            if hit is None:
                total = None
                break
            total += hit
        if total is not None:
            return total
    return None


CHAIN_KERNELS = ("k_point_keys", "k_bucket_rows", "k_reduce_and_neighbours", "k_conv_forward_b3", "k_slice_forward", "k_csr_reduce_segments",
                 "k_conv_backward_fused_b3", "k_reduce_slabs4")


def chain_traffic():
    """(HBM bytes per C3 step summed over the 8 launches of the chain, file) from the newest committed PMC table, or (None, None)."""
    for rnd in (6, 5, 4, 3):
        path = os.path.join(ROOT, "profiles", f"r{rnd}_pmc_traffic.json")
        try:
            with open(path) as f:
                table = json.load(f)
        except OSError:
            continue
        vals = [(table.get(k) or {}).get("traffic_bytes") for k in CHAIN_KERNELS]
        if all(x is not None for x in vals):
            return int(sum(vals)), os.path.relpath(path, ROOT)
    return None, None


def chain_algorithmic_bytes(n, m, d, v, f, e):
    """SURVEY 8(d) / BASELINE.md: splat fwd + neighbour list + conv fwd + slice fwd + slice bwd + conv bwd (2 x conv fwd): 118 MB at C3."""
    splat = n * (4.0 * d + 4.0 * v + 8.0 * (d + 1)) + m * (4.0 * d + 4.0 * v)
    slc = n * (8.0 * (d + 1) + 4.0 * v) + m * 4.0 * v
    nbr = m * (4.0 * d + 4.0 * e)
    conv = m * (4.0 * v + 4.0 * e + 4.0 * f) + 4.0 * e * v * f
    return splat + nbr + 3.0 * conv + 2.0 * slc


def cpu_baseline(cfg, seconds: float, threads: int = 0, min_steps: int = 3):
    """Pure-PyTorch CPU fallback of the same op chain (oracle/torch_fallback.py) on `threads` torch threads (0 = all)."""
    from oracle import torch_fallback as TF
    n, v, f = cfg["n"], cfg["v"], cfg["f"]
    pos = torch.from_numpy(make_cloud(cfg["gen"], n, 0))
    rng = np.random.default_rng(0)
    vals = torch.from_numpy(rng.standard_normal((n, v)).astype(np.float32))
    W = torch.from_numpy((rng.standard_normal((9 * v, f)) / np.sqrt(9 * v)).astype(np.float32))
    G = torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32))
    before = torch.get_num_threads()
    if threads > 0:
        torch.set_num_threads(threads)
    try:
        used = torch.get_num_threads()
        if threads == 0:
            TF.hot_path_step(pos, vals, W, G, cfg["sigma"])  # warm-up
        times = []
        t_end = time.perf_counter() + seconds
        while len(times) < min_steps or (time.perf_counter() < t_end and len(times) < 200):
            t0 = time.perf_counter()
            TF.hot_path_step(pos, vals, W, G, cfg["sigma"])
            times.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(before)
    med = float(np.median(times))
    return {"value": round(n / med / 1e6, 4), "unit": "Mpoints/s", "cores": used, "kind": "port",
            "sample": f"{len(times)} full-size steps of the same workload (median {med * 1e3:.1f} ms/step), "
                      f"pure-PyTorch CPU fallback oracle/torch_fallback.py, torch threads={used}, os.cpu_count()={os.cpu_count()}"}


def stream_copy_ceiling(dev, mib: int = 1024, reps: int = 10):
    """Practical HBM ceiling: device-to-device copy of `mib` MiB, bytes read + written per second."""
    a = torch.empty((mib * 1024 * 1024 // 4,), dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        b.copy_(a)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    return {"GBs": round(2.0 * a.numel() * 4 / ms / 1e6, 1), "what": f"torch device-to-device copy of {mib} MiB, read + written bytes per second, {reps} repetitions"}


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
    floor = None
    try:  # what the lattice operators of this step would cost at their SURVEY 8(d) rooflines (tools/lattice_op_floor.py)
        from tools import lattice_op_floor
        traced = lattice_op_floor.trace(step, Lattice)
        torch.cuda.synchronize()
        floor = lattice_op_floor.price(traced, vertices_per_point_level=lattice.nr_lattice_vertices() if lattice.m_hash_table.is_initialized() else None)
    except Exception as exc:  # a secondary figure
        floor = {"error": f"{type(exc).__name__}: {exc}"[:200]}
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
           "steps": steps, "algorithmic_floor_ms": floor.get("floor_ms") if floor else None,
           "algorithmic_floor": floor, "algorithmic_floor_what": "sum over the lattice operators of the step (SURVEY 8(a) rows: builds, distribute, convolutions "
           "forward + backward, slice / gather / slice_classify) of max(8(d) bytes / 8 TB/s, flop / (2.5 PFLOP/s bf16 / 6 products per fp32 product)); GroupNorm, MLP, loss and optimizer are not in it"}
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
                         "flight, the final synchronize): it is 0.3 % of 2000 steps and 14 % of 20")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed replays for this long BEFORE the W warm-up steps, so that W + K short steps run at the clocks of a busy GPU "
                         "(measured: after set-up and validation the chip needs tens of ms of load to reach them: K = 20 steps ran at "
                         "94.9 us per scan behind W = 5, 88.6 behind W = 2000, 85.3 in a 2000-step run).  0 = off")
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS) + ["ops"],
                    help="ops: only the per-operator roofline table of tools/ops_roofline.py (every SURVEY 8(a) row outside the headline chain)")
    ap.add_argument("--ops-table", type=int, default=1, help="0 = leave the per-operator table out of the default line (rank 0, one GPU, workload C3)")
    ap.add_argument("--roofline-kernel", default="hash_build",
                    help="what the `roofline` block reports: the kernel (group) of the path that sits furthest below its bound — the hash "
                         "build (k_point_keys + k_bucket_rows, profiles/r3_kernel_stats.csv); its dispatches are timed live")
    ap.add_argument("--extra-kernels", default="k_reduce_and_neighbours,k_csr_reduce_segments,k_conv_backward_fused,k_conv_mfma,k_slice_forward",
                    help="kernels timed the same way (reported under roofline_others)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="time budget of the all-cores CPU baseline leg (0 disables both CPU legs)")
    ap.add_argument("--full-unet", type=int, default=1, help="0 = skip the secondary whole-network timing (rank 0, one GPU, workload C3)")
    ap.add_argument("--extras", type=int, default=1,
                    help="0 = only the timed region (no roofline_others / stages / latency legs): profiling runs whose call counts are per step")
    ap.add_argument("--autograd-threads", type=int, default=0, help="1 = leave torch's per-device autograd worker thread on")
    ap.add_argument("--mode", default="graph", choices=["graph", "eager"],
                    help="graph (default): the whole step (forward + backward, ~9 launches) is captured into hipGraphs with the lattice in "
                         "static-rows mode and every timed step is one graph replay; eager: one Python autograd pass per step")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="graph mode: independent scans in flight per GPU (own clouds, lattice, hipGraphs, stream each); the kernels of one "
                         "scan are latency-bound chains at ~1 workgroup per CU, further scans fill the idle slots.  4 = one per hardware "
                         "queue of the default HIP configuration (streams picked by capture.concurrent_streams); 1 = strictly one scan "
                         "after the other")
    ap.add_argument("--pool", type=int, default=8,
                    help="graph mode: distinct clouds per scan in flight; every timed step takes the next one (one captured graph per cloud, "
                         "all sharing the scan's lattice, bounds and workspaces — no input copies)")
    ap.add_argument("--regions", type=int, default=int(os.environ.get("LN_BENCH_REGIONS", "1")),
                    help="1 (default, graph mode): kd region planes (calibrated on a cloud OUTSIDE the pool) so that the scatter kernels walk "
                         "one compact region of the lattice per XCD")
    ap.add_argument("--slot-order", default=os.environ.get("LATTICE_SLOT_ORDER", "hash"), choices=["space", "hash"],
                    help="hash (default): the kd planes steer the segment walks only; space: they also order the SLOTS of the table, so that rows "
                         "follow space (LnTable.slot_map: XCD-local gathers in the convolutions — 48 MB less HBM traffic per step and no gain in "
                         "time, DESIGN.md 8; A/B)")
    ap.add_argument("--batch", type=int, default=0,
                    help="clouds per step (multi-cloud launch form, Lattice.set_cloud_batch): the step's positions are B independent clouds of the "
                         "workload's size in ONE table (per-cloud lattices translated apart in key space), every kernel of the chain launched "
                         "once for the batch; value counts all points.  0 = the workload's default (C2: 8, others: 1)")
    ap.add_argument("--row-slack", type=float, default=0.06,
                    help="graph mode: static row bound = largest vertex count of the calibration clouds x (1 + slack), rounded up to 256")
    args = ap.parse_args()

    from lattice_net_amd import sharding
    world, rank, local_rank = sharding.env_world()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    cores = sharding.pin_launch_thread(local_rank, local_world)  # before anything touches the GPU
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
    from lattice_net_amd.capture import CapturedStep, concurrent_streams
    from lattice_net_amd import lattice as _lattice_mod
    _lattice_mod.set_slot_order(args.slot_order)
    lib = L.load_library()
    if not args.autograd_threads:
        # run backward on the calling thread: the hand-off to torch's per-device autograd worker costs tens of
        # microseconds per step, which is comparable to the whole GPU time of this path
        torch.autograd.set_multithreading_enabled(False)

    if args.workload == "ops":
        from tools import ops_roofline
        table = ops_roofline.run(dev, reps=24) if rank == 0 else None
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            emit({"metric": "per-operator roofline table, SURVEY 8(a) rows outside the headline chain (tools/ops_roofline.py)",
                  "value": None, "unit": "us per call, GB/s, TFLOP/s (see entries)", "n_gpus": 1, "data": "synthetic",
                  "config": {"workload": "C3 scan (120k LiDAR-like points, sigma 0.9, capacity 100k), SemanticKITTI network widths"},
                  **table}, os.path.join(ROOT, "bench_ops_details.json"))
        return
    cfg = dict(WORKLOADS[args.workload])
    batch = args.batch if args.batch > 0 else int(cfg.get("batch", 1))
    n0 = cfg["n"]  # points per cloud
    if batch > 1:  # B clouds per step: one table sized for all of them, the CPU baseline and the algorithmic byte counts per step likewise
        cfg["n"], cfg["capacity"] = n0 * batch, cfg["capacity"] * batch
        cfg["desc"] += f"; {batch} clouds per step (set_cloud_batch: one launch chain for the batch)"
    n, v, f, sigma, cap = cfg["n"], cfg["v"], cfg["f"], cfg["sigma"], cfg["capacity"]
    d, e = 3, 9
    half = bool(cfg.get("half"))
    graph_mode = args.mode == "graph"
    in_flight = max(1, args.in_flight) if graph_mode else 1
    pool = max(1, args.pool) if graph_mode else 1
    _lattice_mod.set_scans_in_flight(in_flight)  # builds that overlap with other scans' kernels take narrower bucket-pass workgroups (ln_build_concurrency)
    # independent clouds per rank (weak scaling); parameters broadcast from rank 0 over RCCL
    # LATTICE_BENCH_RANK_OFFSET=r (testing aid): a one-rank run works on the clouds rank r of a larger job would own
    rank_offset = int(os.environ.get("LATTICE_BENCH_RANK_OFFSET", "0"))
    torch.manual_seed(1234)  # the filter bank is the same in every run (checksums of different runs are comparable)
    bound_w = float(np.sqrt(3.0) * np.sqrt(2.0) / np.sqrt(f))  # kaiming-uniform fan_out (lattice_modules.py:202-207)
    W = ((torch.rand((e * v, f), device=dev) * 2 - 1) * bound_w)
    sharding.broadcast_parameters(dist, [W], src=0)
    W.requires_grad_(True)
    tol = 2e-3 if half else 1e-5  # fp32 path: 1e-5 (BASELINE.json); fp16 feature path (C5): as tests/test_gpu_surface.py

    def new_cloud(seed):
        rng = np.random.default_rng(10_000 + seed)
        pos_np = make_cloud(cfg["gen"], n0, seed) if batch <= 1 else np.concatenate([make_cloud(cfg["gen"], n0, seed * batch + b) for b in range(batch)], 0)
        c = {"pos": torch.from_numpy(np.ascontiguousarray(pos_np)).to(dev),
             "vals": torch.from_numpy(rng.standard_normal((n, v)).astype(np.float32)).to(dev),
             "G": torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).to(dev), "state": {}}
        if half:
            c["vals"], c["G"] = c["vals"].half(), c["G"].half()
        return c

    class ScanSet:
        """One scan in flight: a pool of distinct clouds, ONE lattice (table, bounds, workspaces), one captured graph per cloud,
        one stream."""

        def __init__(self, k):
            base = sharding.cloud_seed(rank + rank_offset, k) * 64
            self.clouds = [new_cloud(base + p) for p in range(pool)]
            self.calibration = [new_cloud(base + 32 + p) for p in range(2)] if graph_mode else []  # never replayed
            self.lat = L.Lattice(sigmas=[sigma] * d, capacity=cap, device=dev)
            if batch > 1:
                self.lat.set_cloud_batch(n0)
            self.stream = scan_streams[k] if k > 0 else None  # set 0 stays on the current stream
            self.cap = None

        def step_on(self, c):
            def step():
                W.grad = None
                lv, wrap, idx, w = L.SplatLattice.apply(self.lat, c["pos"], c["vals"])  # clear + hash build + accumulate
                m = self.lat.nr_lattice_vertices()                                      # eager: the path's one host readback
                if half:  # fp16 feature path: fp16 point features, lattice values and filter bank (rounded per step from the fp32
                    # master copy inside the operator, its gradient returned in fp32); fp32 accumulation everywhere
                    lv = lv[:m].half().requires_grad_(True)
                else:
                    lv = lv[:m].requires_grad_(True)
                cv, cwrap = L.ConvIm2RowLattice.apply(lv, self.lat, W, 1)              # neighbour list + gather-GEMM
                out = L.SliceLattice.apply(cv, cwrap.lattice, c["pos"], idx, w)        # slice
                out.backward(c["G"])                                                    # slice bwd, conv bwd (values + filter)
                c["state"].update(m=m, out=out, gv=lv.grad, gw=W.grad, idx=idx)
            return step

        def clear_states(self):
            for c in self.clouds + self.calibration:
                c["state"].clear()

        def capture(self):
            """Eager reference result of every pool cloud, then CapturedStep: row bound + kd planes from the calibration clouds,
            warm-up, one hipGraph per pool cloud."""
            for c in self.clouds:
                self.step_on(c)()
                torch.cuda.synchronize()
                c["m_real"] = c["state"]["m"]
                c["eager_out"] = c["state"]["out"].detach().clone()
                c["eager_gw"] = c["state"]["gw"].detach().clone()
            steps = [self.step_on(c) for c in self.clouds]
            self.cap = CapturedStep(steps[0], [self.lat], row_slack=args.row_slack, regions=bool(args.regions),
                                    region_indices=lambda: self.calibration[0]["state"]["idx"], stream=self.stream,
                                    before_capture=self.clear_states, calibration_steps=[self.step_on(c) for c in self.calibration],
                                    more_steps=steps[1:])

        def launch(self, i=0):
            if self.cap is None:
                self.step_on(self.clouds[i % pool])()
            else:
                self.cap.launch(i % pool)

        def check(self):
            """Last replayed build inside its bounds; the replayed result of EVERY pool cloud equal to its eager step's."""
            self.cap.check()  # raises if the last replayed build overflowed its row bound or a bucket
            worst = {"out_max_rel": 0.0, "grad_filter_max_rel": 0.0}
            for c in self.clouds:
                st = c["state"]
                if "out" not in st:
                    continue
                worst["out_max_rel"] = max(worst["out_max_rel"], float((st["out"].detach() - c["eager_out"]).abs().max()) /
                                           max(float(c["eager_out"].abs().max()), 1e-30))
                worst["grad_filter_max_rel"] = max(worst["grad_filter_max_rel"], float((st["gw"] - c["eager_gw"]).abs().max()) /
                                                   max(float(c["eager_gw"].abs().max()), 1e-30))
            if max(worst.values()) > tol:
                raise SystemExit(f"[bench] graph replay differs from the eager step: {worst}")
            return worst

    def barrier():
        sharding.barrier(dist)
        torch.cuda.synchronize()

    # streams that land on different hardware queues (an RCCL communicator shifts the mapping: capture.concurrent_streams)
    scan_streams = concurrent_streams(in_flight) if graph_mode else [None] * in_flight
    sets = [ScanSet(k) for k in range(in_flight)]
    graph_err = None
    if graph_mode:
        for cs in sets:
            cs.capture()
        for i in range(pool * in_flight):  # validation replays (untimed): every captured graph once ...
            sets[i % in_flight].launch(i // in_flight)
        barrier()
        errs = [cs.check() for cs in sets]  # ... compared with the eager results before anything is timed
        graph_err = {k: max(er[k] for er in errs) for k in errs[0]}
    value_no_prewarm = None
    if args.prewarm_ms > 0 and graph_mode:  # the same W + K steps ONCE before the pre-warm (the driver's command on a chip that idled through set-up)
        for i in range(max(args.warmup, in_flight)):
            sets[i % in_flight].launch(i // in_flight)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            sets[i % in_flight].launch(i // in_flight)
        barrier()
        el0 = sharding.min_max_over_ranks(dist, time.perf_counter() - t0, dev)[1]
        value_no_prewarm = round(n * world * args.steps / el0 / 1e6, 3)
    prewarm_steps = 0
    if args.prewarm_ms > 0:  # part of the set-up, like capture and validation: brings the GPU to the clocks of sustained load
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
            for i in range(8 * in_flight):
                sets[i % in_flight].launch(i // in_flight)
            prewarm_steps += 8 * in_flight
            torch.cuda.synchronize()
    for i in range(max(args.warmup, in_flight)):  # the W warm-up steps, directly in front of the timed region
        sets[i % in_flight].launch(i // in_flight)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):  # K steps = K scans, issued round-robin over the scans in flight, each taking the next cloud of its pool
        sets[i % in_flight].launch(i // in_flight)
    barrier()
    elapsed = time.perf_counter() - t0
    if graph_mode:
        for cs in sets:
            cs.check()
    checksum_local = sum(float(c["state"]["out"].double().abs().sum().item()) for cs in sets for c in cs.clouds if "out" in c["state"])
    m_all = [[c.get("m_real", c["state"].get("m")) for c in cs.clouds] for cs in sets]
    min_elapsed, max_elapsed = sharding.min_max_over_ranks(dist, elapsed, dev)
    checksum = sharding.gather_sum(dist, checksum_local, dev)
    cs0 = sets[0]
    m = int(np.mean([x for x in m_all[0] if x]))  # vertices of a typical scan (the algorithmic byte counts below use it)
    extras = bool(args.extras) and rank == 0

    # ---- latency of ONE scan (wall time of one pass, hipEvent pairs, median): the same graphs, one scan at a time
    latency = None
    if extras and graph_mode:
        reps = min(max(args.steps, 20), 200)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for i in range(8):
            cs0.launch(i)
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(evs):
            a.record()
            cs0.launch(i)
            b.record()
        torch.cuda.synchronize()
        us = np.array([a.elapsed_time(b) * 1e3 for a, b in evs])
        t1 = time.perf_counter()
        for i in range(reps):
            cs0.launch(i)
        torch.cuda.synchronize()
        wall_us = (time.perf_counter() - t1) / reps * 1e6
        latency = {"what": "one scan in flight: the same hipGraphs replayed one at a time on one stream, clouds rotating through the pool",
                   "us_per_scan_median": round(float(np.median(us)), 1), "us_per_scan_p10_p90": [round(float(np.percentile(us, 10)), 1),
                                                                                               round(float(np.percentile(us, 90)), 1)],
                   "mpoints_per_s": round(n / float(np.median(us)), 1), "timing": f"hipEvent pair around each of {reps} replays, median",
                   "back_to_back_us_per_scan": round(wall_us, 1)}
        cs0.check()

    # ---- per-kernel timing.  Dispatch-bound event pairs (ln_profile_*: the kernel's own duration, as rocprofv3 reports it) on eager
    # steps of scan 0's lattice; with `load` the other scans in flight keep replaying their graphs meanwhile, i.e. the kernels are
    # timed under the contention of the timed region (a graph replay itself has no room for event pairs).
    if graph_mode:
        cs0.lat.set_static_rows(None)
    eager_step = cs0.step_on(cs0.clouds[0])

    def time_kernels(names, reps, load):
        if args.prewarm_ms > 0:  # untimed steps first: the chip has idled through the host-side bookkeeping before this (see --prewarm-ms)
            t_pre = time.perf_counter()
            while (time.perf_counter() - t_pre) * 1e3 < min(args.prewarm_ms, 25.0):
                eager_step()
            torch.cuda.synchronize()
        if lib.ln_profile_begin(",".join(names).encode(), 16 * reps * len(names) + 16) != 0:
            return None
        for r in range(reps):
            if load:
                for other in sets[1:]:
                    other.launch(r)
            eager_step()
        torch.cuda.synchronize()
        tms, cnt = C.c_double(0.0), C.c_int(0)
        lib.ln_profile_end(C.byref(tms), C.byref(cnt))
        return (tms.value, cnt.value) if cnt.value > 0 else None

    def roofline_entry(kernel, reps=20):
        """`avg_us` = average duration of one execution with NOTHING else on the GPU (the kernel's own efficiency; the mode of
        profiles/r3_kernel_stats_one_in_flight.csv); `avg_us_in_flight` = the same dispatches timed while the other scans in flight
        replay their graphs back to back (an upper bound on the stretch a kernel sees in the timed region)."""
        names = KERNEL_GROUPS.get(kernel, [kernel])
        bound_kind, unit, amount = algorithmic_work(kernel, n, m, d, v, f, e, cap)
        alone = time_kernels(names, reps, load=False)
        if alone is None:
            return None
        loaded = time_kernels(names, max(reps // 2, 5), load=True) if (graph_mode and in_flight > 1) else None
        per_group = lambda t: t[0] / (t[1] / len(names)) / 1e3  # seconds per execution of the whole group
        avg_s = per_group(alone)
        if bound_kind == "hbm":
            achieved, peak = amount / avg_s / 1e9, HBM_PEAK_GBS
        elif bound_kind == "mfma_f16":
            achieved, peak, bound_kind = amount / avg_s / 1e12, MFMA_F16_PEAK_TFLOPS, "mfma"
        else:
            achieved, peak = amount / avg_s / 1e12, MFMA_F32_PEAK_TFLOPS
        mw = mfma_work(kernel, m, v, f, e, half)
        mfma = None
        if mw is not None:
            # what the matrix pipe did: the bf16x3 kernels execute six bf16 products per fp32 product (v_mfma_f32_16x16x32_bf16,
            # 2.5 PFLOP/s dense); the fp32-input kernels one v_mfma_f32_16x16x4_f32 product (157.3 TFLOP/s)
            tf = mw[0] / avg_s / 1e12
            ipeak = MFMA_F16_PEAK_TFLOPS if (mw[1] > 1.0 or half) else MFMA_F32_PEAK_TFLOPS
            mfma = {"fp32_equivalent_tflops": round(tf, 2), "executed_tflops": round(tf * mw[1], 2), "instruction_peak_tflops": ipeak,
                    "frac_of_instruction_peak": round(tf * mw[1] / ipeak, 4), "frac_fp32_equivalent_of_f32_peak": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                    "instruction": "bf16 (fp32 operands split three ways, 6 products per fp32 product)" if mw[1] > 1.0 else
                                   ("f16" if half else "v_mfma_f32_16x16x4_f32")}
        rule = ("convolutions: HBM-bound below 64 channels, matrix-bound from 64 on (SURVEY 8(d); rule of rounds 5+, rounds 1-4 switched at 128 — "
                "fractions of 64..127-channel workloads are not comparable across that change)") if "conv" in kernel else None
        return {"bound": bound_kind, "bound_rule": rule, "achieved": round(achieved, 3), "peak": peak, "unit": unit, "frac": round(achieved / peak, 4), "mfma": mfma,
                "traffic": pmc_traffic(kernel) if args.workload == "C3" else None, "kernel": "+".join(names),
                "avg_us": round(avg_s * 1e6, 2), "avg_us_in_flight": round(per_group(loaded) * 1e6, 2) if loaded else None,
                "launches_timed": alone[1],
                "timing": "event pair bound to each dispatch (hipExtLaunchKernelGGL: the kernel's own begin-to-end time) in eager steps of "
                          "scan 0, nothing else on the GPU" +
                          (f"; avg_us_in_flight: while the other {in_flight - 1} scans replay their graphs back to back" if loaded else "")}

    roofline, others = None, []
    if rank == 0:
        for _ in range(3):
            eager_step()
        torch.cuda.synchronize()
        if extras and latency is not None:  # the same step without graphs: one Python autograd pass per scan (round 1's execution)
            t1 = time.perf_counter()
            for _ in range(50):
                eager_step()
            torch.cuda.synchronize()
            latency["eager_us_per_scan"] = round((time.perf_counter() - t1) / 50 * 1e6, 1)
        roofline = roofline_entry(args.roofline_kernel)
    if extras:
        for name in [k for k in args.extra_kernels.split(",") if k and k != args.roofline_kernel]:
            try:
                ent = roofline_entry(name, reps=24)
            except ValueError:
                ent = None
            if ent:
                others.append(ent)
    barrier()

    # ---- per-stage GPU time ("report each stage separately and splat+slice alone"), event-to-event on the launch
    # stream in eager steps; the HBM fraction of splat+slice uses the algorithmic bytes of 8d.
    stages = None
    if extras:
        c0 = cs0.clouds[0]
        names = ["splat", "conv", "slice", "backward"]
        acc_ms = {nm: [] for nm in names}
        reps = 15
        for _ in range(reps):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            W.grad = None
            ev[0].record()
            lv, wrap, idx, w = L.SplatLattice.apply(cs0.lat, c0["pos"], c0["vals"])
            ev[1].record()
            mm = cs0.lat.nr_lattice_vertices()
            lv = (lv[:mm].half() if half else lv[:mm]).requires_grad_(True)
            cv, cwrap = L.ConvIm2RowLattice.apply(lv, cs0.lat, W, 1)
            ev[2].record()
            out = L.SliceLattice.apply(cv, cwrap.lattice, c0["pos"], idx, w)
            ev[3].record()
            out.backward(c0["G"])
            ev[4].record()
            torch.cuda.synchronize()
            for k, nm in enumerate(names):
                acc_ms[nm].append(ev[k].elapsed_time(ev[k + 1]))
        us = {nm: float(np.median(acc_ms[nm])) * 1e3 for nm in names}  # (median: one host hiccup between two launches would own a mean)
        splat_bytes = n * (4.0 * d + 4.0 * v + 8.0 * (d + 1)) + m * (4.0 * d + 4.0 * v)
        slice_bytes = n * (8.0 * (d + 1) + 4.0 * v) + m * 4.0 * v
        ss_bytes = splat_bytes + slice_bytes
        stages = {"us": {k: round(x, 1) for k, x in us.items()},
                  "note": "eager steps, event-to-event on the launch stream, median of 15; `splat` = clear + hash build + accumulate (+ the neighbour "
                          "prefetch issued behind it), `conv` includes the wait for the vertex-count readback"}
        if graph_mode and not half:
            # splat -> slice (forward) alone, in the benchmark's execution mode: one hipGraph per cloud, clouds rotating, scans in flight.
            # (No convolution follows on these lattices: Lattice.prefetch_neighbours is off, the neighbour list is not part of the pair.)
            try:
                chains = []
                for k in range(in_flight):
                    ch = {"lat": L.Lattice(sigmas=[sigma] * d, capacity=cap, device=dev), "clouds": sets[k].clouds, "cal": sets[k].calibration}
                    if batch > 1:
                        ch["lat"].set_cloud_batch(n0)
                    ch["lat"].prefetch_neighbours = False

                    def chain_on(c, ch=ch):
                        def chain():
                            with torch.no_grad():
                                lv2, _, idx2, w2 = L.SplatLattice.apply(ch["lat"], c["pos"], c["vals"])
                                m2 = ch["lat"].nr_lattice_vertices()
                                c["state"].update(cidx=idx2, cout=L.SliceLattice.apply(lv2[:m2], ch["lat"], c["pos"], idx2, w2))
                        return chain

                    for c in ch["clouds"]:
                        chain_on(c)()
                        torch.cuda.synchronize()
                        c["chain_ref"] = c["state"]["cout"].detach().clone()
                    fns = [chain_on(c) for c in ch["clouds"]]
                    ch["cap"] = CapturedStep(fns[0], [ch["lat"]], row_slack=args.row_slack, regions=bool(args.regions),
                                             region_indices=lambda ch=ch: ch["cal"][0]["state"]["cidx"], stream=(scan_streams[k] if k > 0 else None),
                                             calibration_steps=[chain_on(c) for c in ch["cal"]], more_steps=fns[1:])
                    chains.append(ch)
                for i in range(pool * in_flight):
                    chains[i % in_flight]["cap"].launch(i // in_flight % pool)
                torch.cuda.synchronize()
                for ch in chains:
                    ch["cap"].check()
                    for c in ch["clouds"]:
                        err2 = float((c["state"]["cout"] - c["chain_ref"]).abs().max()) / max(float(c["chain_ref"].abs().max()), 1e-30)
                        if err2 > 1e-5:
                            raise RuntimeError(f"replayed splat -> slice differs from the eager one: {err2}")

                def chain_rate(k_in_flight, reps2):
                    for i in range(30):
                        chains[i % k_in_flight]["cap"].launch(i // k_in_flight % pool)
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    for i in range(reps2):
                        chains[i % k_in_flight]["cap"].launch(i // k_in_flight % pool)
                    torch.cuda.synchronize()
                    us2 = (time.perf_counter() - t2) / reps2 * 1e6
                    return {"us_per_scan": round(us2, 1), "achieved_GBs": round(ss_bytes / us2 / 1e3, 1),
                            "frac_of_hbm_peak": round(ss_bytes / us2 / 1e3 / HBM_PEAK_GBS, 4)}

                one = chain_rate(1, 300)
                stages["splat_plus_slice"] = dict(what="splat -> slice (forward) as one hipGraph per cloud, ONE scan at a time (300 scans timed)",
                                                  algorithmic_bytes=int(ss_bytes), **one)
                if in_flight > 1:
                    stages["splat_plus_slice_in_flight"] = dict(
                        what=f"the same, {in_flight} scans in flight (600 scans timed)", algorithmic_bytes=int(ss_bytes), **chain_rate(in_flight, 600))
                for ch in chains:
                    ch["cap"].check()
            except Exception as ex:  # secondary figure: never endangers the bench line
                stages["splat_plus_slice_in_flight"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        else:
            ss_us = us["splat"] + us["slice"]
            stages["splat_plus_slice"] = {"what": "eager stages (includes the neighbour prefetch)", "us_per_scan": round(ss_us, 1),
                                          "algorithmic_bytes": int(ss_bytes), "achieved_GBs": round(ss_bytes / ss_us / 1e3, 1),
                                          "frac_of_hbm_peak": round(ss_bytes / ss_us / 1e3 / HBM_PEAK_GBS, 4)}

    line = None
    if rank == 0:
        value = n * world * args.steps / max_elapsed / 1e6
        cpu = cpu1 = None
        if world == 1 and args.cpu_seconds > 0:
            # the CPU fallback with torch.set_num_threads(1) and with all cores.  index_add_ / scatter on many threads
            # contend (128 threads are SLOWER than one on this chain), so a 16-thread leg is timed too and `cpu_baseline` is the
            # fastest of the multi-threaded legs; `cpu_baseline_1thread` is always reported beside it.
            cfg_cpu = WORKLOADS[args.workload]  # (one cloud of the workload's own size: a CPU has nothing to gain from a batch)
            legs = [cpu_baseline(cfg_cpu, args.cpu_seconds / 2, threads=0)]
            if (os.cpu_count() or 1) > 16:
                legs.append(cpu_baseline(cfg_cpu, args.cpu_seconds / 2, threads=16))
            cpu = max(legs, key=lambda leg: leg["value"])
            cpu["other_thread_counts"] = [{"cores": leg["cores"], "value": round(leg["value"], 4)} for leg in legs if leg is not cpu]
            cpu1 = cpu_baseline(cfg_cpu, 0.0, threads=1, min_steps=3)
        unet = None
        if world == 1 and args.full_unet and args.workload == "C3" and extras:
            try:
                unet = full_unet_step(dev, n)
            except Exception as exc:  # secondary number: never take the headline line down with it
                unet = {"error": f"{type(exc).__name__}: {exc}"}
        copy_ceiling = stream_copy_ceiling(dev) if extras else None
        ops_table = None
        if world == 1 and args.ops_table and args.workload == "C3" and extras:
            try:  # every SURVEY 8(a) row outside the headline chain: distribute, coarse build, level-crossing convolutions, gather,
                  # slice_classify, the dense contraction at 64 / 128 channels (secondary numbers: never take the headline line down)
                from tools import ops_roofline
                ops_table = ops_roofline.run(dev, reps=24)
            except Exception as exc:
                ops_table = {"error": f"{type(exc).__name__}: {exc}"}
        roofline_chain = None
        if args.workload == "C3":
            cb = chain_algorithmic_bytes(n, m, d, v, f, e)
            tb, tfile = chain_traffic()
            step_s = max_elapsed / args.steps
            roofline_chain = {"algorithmic_bytes": int(cb), "traffic_bytes": tb, "traffic_from": tfile,
                              "frac_in_flight": round(cb / step_s / 1e9 / HBM_PEAK_GBS, 4),
                              "frac_one_pass": round(cb / (latency["us_per_scan_median"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if latency else None,
                              "traffic_over_algorithmic": round(tb / cb, 2) if tb else None}
        exec_desc = (f"{in_flight} independent scan(s) in flight per GPU (own lattice, stream and {pool} clouds each); every step = one hipGraph "
                     f"replay of the whole forward + backward on the next cloud of the scan's pool; value = points of all K steps / wall time"
                     if graph_mode else "eager: one Python autograd pass per step")
        line = {
            "metric": "Mpoints/sec splat+conv+slice fwd+bwd on 120k-pt SemanticKITTI scan",
            "value": round(value, 3), "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(max_elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16 features / f32 accumulate" if half else "f32 (convolution products bf16x3-emulated)",
            "dtype_note": None if half else "fp32 everywhere; the convolutions split both fp32 operands three ways into bf16 and run 6 of the 9 "
                          "partial products on the bf16 matrix cores with fp32 accumulation (1e-5 per element against fp64, "
                          "tests/test_gpu_parity.py; LN_CONV_EXACT_F32=1 selects v_mfma_f32_16x16x4_f32 instead)",
            "data": "synthetic",
            "value_definition": f"throughput: {in_flight} scan(s) in flight per GPU, hipGraph replays (config.workload); the SURVEY 8(d) number — one "
                                "pass at a time, hipEvent median — is value_8d_one_pass_mpoints_per_s = latency.mpoints_per_s",
            "value_8d_one_pass_mpoints_per_s": latency["mpoints_per_s"] if latency else None,
            "value_no_prewarm": value_no_prewarm,
            "roofline_chain": roofline_chain,
            "config": {"workload": cfg["desc"] + f"; THROUGHPUT definition: {exec_desc}.  The latency of one scan alone is `latency`",
                       "points_per_gpu": n, "vertices": m, "val_dim": v, "nr_filters": f,
                       "sharding": f"{world} rank(s), independent clouds per GPU", "checksum": round(checksum, 3),
                       "scans_in_flight": in_flight, "clouds_per_scan_pool": pool, "clouds_per_step": batch, "points_per_cloud": n0,
                       "prewarm": f"{prewarm_steps} untimed replays ({args.prewarm_ms:g} ms) before the W warm-up steps" if prewarm_steps else None, "kd_regions": bool(args.regions and graph_mode),
                       "slot_order": args.slot_order if (args.regions and graph_mode) else "hash",
                       "row_bounds": [cs.cap.bounds[0] for cs in sets] if graph_mode else None,
                       "bounds_and_planes_calibrated_on": "2 clouds per scan that are not in its pool" if graph_mode else None,
                       "vertices_per_scan": m_all, "graph_vs_eager": graph_err},
            "latency": latency, "roofline": roofline, "roofline_others": others, "stages": stages, "hbm_copy_ceiling": copy_ceiling,
            "full_unet": unet, "ops": ops_table, "cpu_baseline": cpu, "cpu_baseline_1thread": cpu1,
            "host_cores_of_rank0": len(cores),
        }
        if world > 1:  # spread of the ranks' own clocks over the same K steps (value uses the max)
            line["ms_per_step_min_over_ranks"] = round(min_elapsed / args.steps * 1e3, 4)
            line["ms_per_step_max_over_ranks"] = round(max_elapsed / args.steps * 1e3, 4)
    try:  # RCCL prints a version banner through C stdio, which a pipe buffers until exit: every rank flushes it now
        C.CDLL(None).fflush(None)
    except OSError:
        pass
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(line)


if __name__ == "__main__":
    main()
