"""world_size-2 gloo test of the N>1 control flow used by bench.py: independent clouds per rank,
parameters broadcast from rank 0, MAX-reduced timing, gathered checksums; no data-path collective."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from lattice_net_amd import sharding, synthetic
    from oracle import torch_fallback as TF

    torch.set_num_threads(1)
    dist = sharding.init("gloo")
    assert dist is not None and dist.get_world_size() == world
    dev = torch.device("cpu")
    n, v, f = 400, 4, 8
    pos = torch.from_numpy(synthetic.lidar_cloud(n, sharding.cloud_seed(rank)))
    g = torch.Generator().manual_seed(100 + rank)
    W = torch.rand((9 * v, f), generator=g)          # different on every rank before the broadcast
    sharding.broadcast_parameters(dist, [W], src=0)
    vals = torch.ones((n, v))
    out, gf, _, lat = TF.hot_path_step(pos, vals, W, torch.ones((n, f)), 0.9)   # each rank: its own cloud, its own table
    checksum = float(out.double().abs().sum())
    total = sharding.gather_sum(dist, checksum, dev)
    tmax = sharding.max_over_ranks(dist, 1.0 + rank, dev)
    tlo, thi = sharding.min_max_over_ranks(dist, 1.0 + rank, dev)     # the spread bench.py puts on the line at N > 1
    cores = sharding.pin_launch_thread(rank, world)                    # each rank on its own slice of the host's cores
    sharding.barrier(dist)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), W=W.numpy(), m=lat.m, checksum=checksum, total=total, tmax=tmax,
             pos0=pos[0].numpy(), tlo=tlo, thi=thi, cores=np.array(cores))
    dist.destroy_process_group()


def test_two_rank_sharding(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"r{i}.npz") for i in range(world)]
    np.testing.assert_array_equal(r[0]["W"], r[1]["W"])           # parameters replicated from rank 0
    assert not np.array_equal(r[0]["pos0"], r[1]["pos0"])          # independent clouds
    assert r[0]["total"] == r[1]["total"]
    assert abs(float(r[0]["total"]) - (float(r[0]["checksum"]) + float(r[1]["checksum"]))) < 1e-6 * float(r[0]["total"])
    assert float(r[0]["tmax"]) == float(r[1]["tmax"]) == 2.0       # whole-job time = slowest rank
    assert [float(r[i]["tlo"]) for i in range(2)] == [1.0, 1.0] and [float(r[i]["thi"]) for i in range(2)] == [2.0, 2.0]
    if len(os.sched_getaffinity(0)) >= 4:                          # disjoint core slices
        assert not set(r[0]["cores"].tolist()) & set(r[1]["cores"].tolist())


def test_cloud_assignment_round_robin():
    from lattice_net_amd import sharding
    assert sharding.clouds_of_rank(8, 8, 3) == [3]
    assert sharding.clouds_of_rank(10, 4, 1) == [1, 5, 9]
    got = sorted(sum((sharding.clouds_of_rank(13, 4, r) for r in range(4)), []))
    assert got == list(range(13))
