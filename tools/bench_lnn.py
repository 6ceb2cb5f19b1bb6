#!/usr/bin/env python3
"""Whole-network timing: LNN forward + backward + AdamW step on one synthetic LiDAR-like scan with
the reference's SemanticKITTI model shape (config/lnn_train_semantic_kitti.cfg:36-47,62-69).  Secondary number; the
headline metric stays bench.py's op chain."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lattice_net_amd import Lattice, ModelParams, synthetic  # noqa: E402
from lattice_net_amd.losses import nll_loss_gather  # noqa: E402
from lattice_net_amd.models import LNN  # noqa: E402

PRESETS = {
    # the model / lattice blocks of the reference's configs (values only), with the synthetic cloud SURVEY.md 8d pairs with each
    "kitti": dict(n=120000, classes=20, cloud="lidar", values=1, cfg="""
model: { positions_mode: "xyz"  values_mode: "none"  pointnet_layers: [16,32]  pointnet_start_nr_channels: 32  nr_downsamples: 2
    nr_blocks_down_stage: [1,1,1]  nr_blocks_bottleneck: 1  nr_blocks_up_stage: [1,1,1]  nr_levels_down_with_normal_resnet: 3
    nr_levels_up_with_normal_resnet: 3  compression_factor: 1.0  dropout_last_layer: 0.0 }
lattice_gpu: { hash_table_capacity: 100000  nr_sigmas: 1  sigma_0: "0.9 3" }
"""),  # config/lnn_train_semantic_kitti.cfg:36-47,62-69
    "shapenet": dict(n=2500, classes=50, cloud="box", values=1, cfg="""
model: { positions_mode: "xyz"  values_mode: "none"  pointnet_channels_per_layer: [16,32,64]  pointnet_start_nr_channels: 32
    nr_downsamples: 3  nr_blocks_down_stage: [4,4,4]  nr_blocks_bottleneck: 3  nr_blocks_up_stage: [2,2,2]
    nr_levels_down_with_normal_resnet: 3  nr_levels_up_with_normal_resnet: 2  compression_factor: 1.0  dropout_last_layer: 0.0 }
lattice_gpu: { hash_table_capacity: 60000  nr_sigmas: 1  sigma_0: "0.05 3" }
"""),  # config/ln_train_shapenet_example.cfg:19-31,45-50
    "scannet": dict(n=200000, classes=21, cloud="planes", values=4, cfg="""
model: { positions_mode: "xyz"  values_mode: "rgb+height"  pointnet_layers: [16,32,64]  pointnet_start_nr_channels: 32
    nr_downsamples: 3  nr_blocks_down_stage: [6,6,8]  nr_blocks_bottleneck: 8  nr_blocks_up_stage: [2,2,2]
    nr_levels_down_with_normal_resnet: 3  nr_levels_up_with_normal_resnet: 3  compression_factor: 1.0  dropout_last_layer: 0.0 }
lattice_gpu: { hash_table_capacity: 5000000  nr_sigmas: 1  sigma_0: "0.08 3" }
"""),  # config/lnn_train_scannet.cfg:22-48
}


def make_graph_step(args, preset, cfg_text, net, opt, gen, dev, nll_loss_gather, first=None):
    """K scans per optimizer step, each captured once (forward + NLL + backward = one hipGraph) on its own stream."""
    from lattice_net_amd import CapturedNetworkStep
    K = max(1, args.in_flight)
    if K > 1 and args.steps > 100:
        print("[bench_lnn] note: long runs of SEVERAL whole-network graphs replaying concurrently abort the HSA queue on this stack "
              "(ROCm 7.2, a few hundred steps in); one graph at a time (--in-flight 1) runs clean for as long as tried", flush=True)
    params = list(net.parameters())
    from lattice_net_amd.capture import concurrent_streams
    scan_streams = concurrent_streams(K + 1)[1:] if K > 1 else [None]  # streams on different hardware queues
    scans = []
    for k in range(K):
        with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
            f.write(cfg_text)
        lat = Lattice.create(f.name, "lattice")
        os.unlink(f.name)
        pos = torch.from_numpy(gen(args.n, k)).to(dev)
        vals = torch.zeros((args.n, 1), device=dev) if preset["values"] == 1 else torch.rand((args.n, preset["values"]), device=dev)
        target = torch.from_numpy(np.random.default_rng(k).integers(0, args.classes, args.n)).to(dev)

        def one(lat=lat, pos=pos, vals=vals, target=target):
            logsoftmax, _ = net(lat, pos, vals)
            loss = nll_loss_gather(logsoftmax, target)
            loss.backward()
            return loss.detach()

        for p in params:
            p.grad = None
        graph_opt = None
        if args.graph_optimizer:  # optimizer step inside the graph: the training loop is nothing but replays
            if K > 1:
                raise SystemExit("--graph-optimizer captures one scan per optimizer step (--in-flight 1)")
            graph_opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-4, amsgrad=True, fused=True, capturable=True)
        cap = CapturedNetworkStep(one, lat, params, stream=scan_streams[k] if K > 1 else None, optimizer=graph_opt)
        scans.append(cap)
        if os.environ.get("LNN_DEBUG"):
            torch.cuda.synchronize()
            print("captured scan", k, cap.bounds, flush=True)
            for it in range(2):
                cap.launch()
                torch.cuda.synchronize()
                print("  replay", it, float(cap.loss), flush=True)
    main_stream = torch.cuda.current_stream()
    if os.environ.get("LNN_DEBUG"):
        for it in range(2):
            for k, cap in enumerate(scans):
                cap.launch()
                torch.cuda.synchronize()
                print("serial replay", it, k, float(cap.loss), flush=True)
        for it in range(2):
            for cap in scans:
                cap.launch()
            torch.cuda.synchronize()
            print("concurrent replay", it, [float(c.loss) for c in scans], flush=True)

    state = {}
    pending = []  # the host stays at most two optimizer steps ahead of the GPU (fewer aborts than without: DESIGN.md 4.7)

    make_graph_step.captures = scans  # (main() checks the row bounds of every replayed build after the timed loop)
    if args.graph_optimizer:
        return lambda: scans[0].launch()

    def step():
        if len(pending) >= 2:
            pending.pop(0).synchronize()
        state["n"] = state.get("n", 0) + 1
        if K > 1 and state["n"] % int(os.environ.get("LNN_SYNC_EVERY", "8")) == 0:
            torch.cuda.synchronize()  # several streams of replays: a device-level wait every few steps (CapturedNetworkStep.launch)
        if K == 1 and not os.environ.get("LNN_CROSS_STREAM"):  # the eager optimizer kernels on the capture stream itself: no event joins
            with torch.cuda.stream(scans[0].stream):
                loss = scans[0].launch()
                scans[0].bind_gradients()
                opt.step()
                ev = torch.cuda.Event()
                ev.record()
            pending.append(ev)
            return loss
        if K > 1 and os.environ.get("LNN_HOST_JOIN"):  # experiment: host-side joins only (no events between the streams)
            torch.cuda.synchronize()
            for cap in scans:
                loss = cap.launch()
            torch.cuda.synchronize()
            CapturedNetworkStep.sum_gradients(scans)
            opt.step()
            return loss
        if K == 1:
            loss = scans[0].launch()
        else:
            for cap in scans:
                cap.stream.wait_stream(main_stream)  # the parameters of the previous optimizer step
                loss = cap.launch()
            for cap in scans:
                main_stream.wait_stream(cap.stream)
        CapturedNetworkStep.sum_gradients(scans)
        opt.step()
        ev = torch.cuda.Event()
        ev.record(main_stream)
        pending.append(ev)
        return loss

    return step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="kitti", choices=sorted(PRESETS))
    ap.add_argument("--n", type=int, default=0, help="points per cloud (0 = the preset's)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--infer", action="store_true", help="forward only, under torch.no_grad()")
    ap.add_argument("--graph", action="store_true", help="forward + loss + backward of a scan as ONE hipGraph replay (CapturedNetworkStep); "
                                                         "the optimizer step stays outside the graph")
    ap.add_argument("--graph-optimizer", action="store_true", help="with --graph: capture the AdamW step (capturable=True) behind the backward pass")
    ap.add_argument("--in-flight", type=int, default=1, help="with --graph: scans per optimizer step, each with its own lattice, graph and "
                                                             "stream, replayed concurrently; their gradients are summed (batch of K scans)")
    ap.add_argument("--host-profile", action="store_true", help="cProfile of the host side of the timed steps")
    ap.add_argument("--gc", type=int, default=0, help="1 = leave Python's cyclic garbage collector on during the timed steps; 2 = on, after gc.freeze()")
    args = ap.parse_args()
    preset = PRESETS[args.config]
    args.n = args.n or preset["n"]
    args.classes = preset["classes"]
    dev = torch.device("cuda", 0)
    with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
        f.write(preset["cfg"])
        path = f.name
    torch.manual_seed(0)
    torch.autograd.set_multithreading_enabled(False)
    mp = ModelParams.create(path)
    lattice = Lattice.create(path, "lattice")
    os.unlink(path)  # both readers are done with the temporary cfg
    path_cfg = preset["cfg"]
    net = LNN(args.classes, mp)
    gen = {"lidar": synthetic.lidar_cloud, "box": synthetic.box_surface_cloud, "planes": synthetic.planes_cloud}[preset["cloud"]]
    pos = torch.from_numpy(gen(args.n, 0)).to(dev)
    vals = torch.zeros((args.n, 1), device=dev) if preset["values"] == 1 else torch.rand((args.n, preset["values"]), device=dev)
    target = torch.from_numpy(np.random.default_rng(0).integers(0, args.classes, args.n)).to(dev)
    opt = None

    def step():
        nonlocal opt
        if args.infer:
            with torch.no_grad():
                logsoftmax, _ = net(lattice, pos, vals)
            return logsoftmax[0, 0]
        logsoftmax, _ = net(lattice, pos, vals)
        loss = nll_loss_gather(logsoftmax, target)
        if opt is None:  # parameters of the PointNet MLP exist only after the first forward (ln_train.py:162-165)
            opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-4, amsgrad=True, fused=True)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if args.graph:
        step = make_graph_step(args, preset, path_cfg, net, opt, gen, dev, nll_loss_gather, first=(lattice, pos, vals, target))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    if args.gc == 2:
        # collector left on, but everything alive after the warm-up (modules, torch internals) is moved out of its reach:
        # full collections then only walk what the steps themselves allocate
        import gc
        gc.collect()
        gc.freeze()
    if not args.gc:
        # the step allocates ~10k short-lived Python objects; generation-2 collections walking the live module / autograd
        # objects cost ~3 ms per step.  Nothing in the step relies on the cycle collector (reference counts free the graph).
        import gc
        gc.collect()
        gc.disable()
    if args.host_profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
    t0 = time.perf_counter()
    for it in range(args.steps):
        loss = step()
        if os.environ.get("LNN_TRACE") and it % 20 == 0:
            print("step", it, float(loss), flush=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    for cap in getattr(make_graph_step, "captures", []) if args.graph else []:
        cap.check()  # raises if a replayed build left its static row bounds (its step would have trained on a truncated lattice)
    if args.host_profile:
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(35)
    nparams = sum(p.numel() for p in net.parameters())
    scans = max(1, args.in_flight) if args.graph else 1
    mode = (f"train step, graph x{scans} in flight" + (" incl. AdamW" if args.graph_optimizer else "") if args.graph else "train step") if not args.infer else "forward"
    print(f"LNN[{args.config}] {mode}: {dt * 1e3 / scans:.2f} ms per scan  ({args.n * scans / dt / 1e6:.2f} Mpoints/s), {nparams} parameters, loss {loss.item():.4f}")


if __name__ == "__main__":
    main()
