#!/usr/bin/env python3
"""Data-parallel LNN training on synthetic scans (reference latticenet_py/ln_train.py:120-190):
every rank builds its own lattices from its own cloud, forward / backward run locally, gradients are averaged with one
bucketed RCCL all-reduce, AdamW steps identically everywhere.  Loss = 0.5 Lovasz-Softmax + 0.5 NLL; IoU summed over
ranks at the end.

  python tools/train_lnn.py --steps 30                                   # one GPU
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_lnn.py --steps 30
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lattice_net_amd import Lattice, ModelParams, sharding, synthetic  # noqa: E402
from lattice_net_amd.losses import LovaszSoftmax, Scores, nll_loss_gather  # noqa: E402
from lattice_net_amd.models import LNN  # noqa: E402

CFG = """
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


def labels_for(pos: np.ndarray, nr_classes: int) -> np.ndarray:
    """A position-dependent labelling (range rings x height bands) so that there is structure to learn."""
    r = np.sqrt(pos[:, 0] ** 2 + pos[:, 1] ** 2)
    ring = np.minimum((r / 10.0).astype(np.int64), 4)
    band = (pos[:, 2] > -1.2).astype(np.int64)
    return (1 + ring * 2 + band) % nr_classes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=120000)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--classes", type=int, default=20)
    ap.add_argument("--clouds", type=int, default=4, help="distinct clouds per rank, cycled")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--graph", action="store_true", help="forward + loss + backward captured once as a hipGraph (lattice_net_amd."
                                                         "CapturedNetworkStep) and fed in place; all-reduce and AdamW stay outside")
    args = ap.parse_args()
    world, rank, local_rank = sharding.env_world()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = sharding.init("nccl", dev)
    with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
        f.write(CFG)
        path = f.name
    torch.manual_seed(0)
    torch.autograd.set_multithreading_enabled(False)
    mp = ModelParams.create(path)
    lattice = Lattice.create(path, "lattice")
    os.unlink(path)  # both readers are done with the temporary cfg
    net = LNN(args.classes, mp)
    sharding.broadcast_parameters(dist, list(net.parameters()) + list(net.buffers()))
    opt = torch.optim.AdamW(net.parameters(), lr=args.lr, weight_decay=1e-4, amsgrad=True, fused=True)  # ln_train.py:165 (+ fused update)
    lovasz = LovaszSoftmax(ignore_index=0)

    def nll(logp, tgt):
        return nll_loss_gather(logp, tgt, ignore_index=0)
    clouds = []
    for k in range(args.clouds):
        pos_np = synthetic.lidar_cloud(args.n, sharding.cloud_seed(rank, k))
        clouds.append((torch.from_numpy(pos_np).to(dev), torch.zeros((args.n, 1), device=dev),
                       torch.from_numpy(labels_for(pos_np, args.classes)).to(dev)))
    scores = Scores()
    losses, t_start = [], None
    import gc
    cap = None
    if args.graph:
        # ONE captured step; every cloud of the rotation is copied into the tensors the graph reads
        from lattice_net_amd import CapturedNetworkStep
        pos_s, vals_s, target_s = (t.clone() for t in clouds[0])
        held = {}

        def one():
            logsoftmax, _ = net(lattice, pos_s, vals_s)
            loss = 0.5 * lovasz(logsoftmax, target_s) + 0.5 * nll(logsoftmax, target_s)
            loss.backward()
            held["logsoftmax"] = logsoftmax.detach()
            return loss.detach()

        def loader(c):
            def load():
                pos_s.copy_(c[0])
                vals_s.copy_(c[1])
                target_s.copy_(c[2])
            return load

        # row bounds over every cloud of the rotation (+10 %); replays are checked against them below
        cap = CapturedNetworkStep(one, lattice, list(net.parameters()), row_slack=0.10, calibration_loaders=[loader(c) for c in clouds])
    import contextlib
    loop_stream = contextlib.ExitStack()
    if cap is not None:
        # the whole loop on the capture stream: eager kernels (input copies, all-reduce hand-off, AdamW) queue behind the replay
        # without cross-stream event joins (CapturedNetworkStep.launch)
        torch.cuda.synchronize()
        loop_stream.enter_context(torch.cuda.stream(cap.stream))
    for step in range(args.steps):
        if step == 2:
            # Python's cycle collector costs ~4 ms per step here: its full passes walk every live module / torch object.  The
            # step itself frees everything by reference counting; gc.freeze() moves what is alive now out of the collector's
            # reach, so it stays on (for whatever user code creates cycles) and only walks what later steps allocate.
            gc.collect()
            gc.freeze()
        if step == min(3, args.steps - 1):
            torch.cuda.synchronize()
            sharding.barrier(dist)
            t_start, timed_from = time.perf_counter(), step
        pos, vals, target = clouds[step % len(clouds)]
        if cap is not None and step % 32 == 31:
            # every replayed build must have stayed inside its static row bounds (vertices beyond a bound are dropped silently on the
            # device): one wait per 32 steps; on a violation the rest of the run is eager
            torch.cuda.current_stream().synchronize()
            try:
                cap.check()
            except Exception as exc:  # noqa: BLE001
                print(f"[train_lnn] rank {rank}: {exc}; continuing in eager mode", flush=True)
                loop_stream.close()
                lattice.set_static_rows(None)
                cap = None
        if cap is not None:
            pos_s.copy_(pos)
            vals_s.copy_(vals)
            target_s.copy_(target)
            loss = cap.launch().clone()
            logsoftmax = held["logsoftmax"]
            cap.bind_gradients()
        else:
            logsoftmax, _ = net(lattice, pos, vals)
            loss = 0.5 * lovasz(logsoftmax, target) + 0.5 * nll(logsoftmax, target)
            opt.zero_grad()
            loss.backward()
        sharding.allreduce_gradients(dist, net.parameters())
        opt.step()
        losses.append(loss.detach())
        if step >= args.steps - len(clouds):
            scores.accumulate_scores(logsoftmax, target, 0)
    torch.cuda.synchronize()
    sharding.barrier(dist)
    dt = (time.perf_counter() - t_start) / max(args.steps - timed_from, 1)
    scores.all_reduce(dist)
    # every rank must hold the same parameters after identical optimizer steps on averaged gradients
    check = torch.stack([p.detach().double().sum() for p in net.parameters()]).sum().reshape(1)
    lo, hi = check.clone(), check.clone()
    if dist is not None:
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if rank == 0:
        ls = torch.stack(losses).tolist()
        print(f"ranks {world}: {dt * 1e3:.2f} ms/step, {world * args.n / dt / 1e6:.2f} Mpoints/s aggregate; loss {ls[0]:.4f} -> {ls[-1]:.4f}; "
              f"mean IoU {scores.avg_class_iou():.3f}; parameter checksum spread over ranks {float(hi - lo):.3e}", flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
