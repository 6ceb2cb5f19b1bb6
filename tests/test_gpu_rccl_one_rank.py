"""RCCL on the GPU box: the pool's boxes have ONE GPU (two ranks on one device are refused by RCCL), so the collectives of
lattice_net_amd.sharding and of bench.py / tools/train_lnn.py are exercised on a one-rank "nccl" process group
(LATTICE_FORCE_DIST=1): communicator set-up, broadcast, bucketed all-reduce, all-gather and barrier run as RCCL kernels
on the device the lattice kernels use.  World sizes > 1 are covered on CPU with gloo (tests/test_distributed_cpu.py,
tests/test_training_pieces.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, os.environ["LN_ROOT"])
from lattice_net_amd import sharding
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist = sharding.init("nccl", dev)
assert dist is not None and dist.get_backend() == "nccl" and dist.get_world_size() == 1
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 20)).to(dev)
before = [p.detach().clone() for p in net.parameters()]
sharding.broadcast_parameters(dist, [p.data for p in net.parameters()])
for a, p in zip(before, net.parameters()):
    assert torch.equal(a, p)
net(torch.randn(128, 32, device=dev)).square().mean().backward()
grads = [p.grad.detach().clone() for p in net.parameters()]
sharding.allreduce_gradients(dist, net.parameters(), bucket_bytes=4096)  # several buckets
for g, p in zip(grads, net.parameters()):
    assert torch.equal(g, p.grad)  # the mean over one rank
assert sharding.max_over_ranks(dist, 3.25, dev) == 3.25 and sharding.gather_sum(dist, 2.5, dev) == 2.5
t = sharding.allreduce_sum_(dist, torch.arange(8, dtype=torch.float32, device=dev))
assert t.tolist() == list(range(8))
sharding.barrier(dist)
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
"""


def _env():
    env = dict(os.environ)
    env.update(LATTICE_FORCE_DIST="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29700 + os.getpid() % 200), HSA_ENABLE_IPC_MODE_LEGACY="0", LN_ROOT=ROOT)
    return env


def test_sharding_collectives_run_over_rccl_on_one_rank():
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_bench_brackets_its_timed_region_with_rccl_collectives():
    """bench.py under a one-rank nccl group: the barrier / max-over-ranks of the contract are RCCL calls on the benchmark's device."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "3", "--extras", "0", "--cpu-seconds", "0",
                        "--full-unet", "0", "--pool", "2"], env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 30 and line["value"] > 100.0


@pytest.mark.parametrize("graph", [False, True])
def test_data_parallel_training_step_over_rccl_on_one_rank(graph):
    """tools/train_lnn.py (forward + loss + backward, bucketed gradient all-reduce, AdamW) with its collectives on
    RCCL: parameters broadcast, gradients all-reduced every step (eager and with the step captured into a hipGraph), the
    parameter checksum gathered at the end."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "train_lnn.py"), "--n", "20000", "--steps", "12", "--clouds", "2"] + (["--graph"] if graph else [])
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("ranks 1:")][-1]
    a, b = last.split("loss ")[1].split(";")[0].split(" -> ")
    assert float(b) < float(a), last
    assert "spread over ranks 0.000e+00" in last


def _bench_line(cmd, env):
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])


def test_two_rank_bench_control_flow_with_real_kernels():
    """The N = 2 path of bench.py with the HIP kernels (LATTICE_BENCH_SHARE_GPU=1: both ranks on GPU 0, collectives over gloo — the
    pool's boxes have one GPU; never a measurement): launched the way the driver launches it, n_gpus == 2, and the job's checksum is
    the sum of what rank 0 and rank 1 compute alone on their own clouds (independent clouds per rank, filter bank broadcast from rank 0)."""
    args = ["--steps", "16", "--warmup", "2", "--extras", "0", "--cpu-seconds", "0", "--full-unet", "0", "--pool", "2", "--in-flight", "2"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LATTICE_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LATTICE_FORCE_DIST"):
        env.pop(k, None)
    port = 29900 + os.getpid() % 90
    two = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                       "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, env)
    assert two["n_gpus"] == 2 and two["steps"] == 16 and two["scaling"] == "weak" and two["value"] > 0
    env1 = dict(env)
    env1.pop("LATTICE_BENCH_SHARE_GPU")
    alone = []
    for r in (0, 1):
        line = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, dict(env1, LATTICE_BENCH_RANK_OFFSET=str(r)))
        assert line["n_gpus"] == 1
        alone.append(line)
    c0, c1 = alone[0]["config"]["checksum"], alone[1]["config"]["checksum"]
    assert alone[0]["config"]["vertices_per_scan"] != alone[1]["config"]["vertices_per_scan"], "the two ranks must work on different clouds"
    assert abs(c0 - c1) > 1e-6 * abs(c0)
    assert abs(two["config"]["checksum"] - (c0 + c1)) <= 2e-6 * abs(c0 + c1), (two["config"]["checksum"], c0, c1)
    assert two["config"]["vertices_per_scan"] == alone[0]["config"]["vertices_per_scan"]  # the line's scan table is rank 0's
