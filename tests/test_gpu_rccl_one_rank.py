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
