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


def _bench_line(cmd, env, details=False):
    """The compact line (the LAST stdout line, what the driver parses) and, on request, the `DETAILS ` line in front of it."""
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert lines[-1].startswith('{"metric"') and len(lines[-1]) < 4096, (len(lines[-1]), lines[-1][:200])
    line = json.loads(lines[-1])
    if not details:
        return line
    full = json.loads([ln for ln in lines if ln.startswith("DETAILS ")][-1][len("DETAILS "):])
    for k in ("value", "n_gpus", "steps", "ms_per_step"):
        assert full[k] == line[k]
    return line, full


def test_two_rank_bench_control_flow_with_real_kernels():
    """The N = 2 path of bench.py with the HIP kernels (LATTICE_BENCH_SHARE_GPU=1: both ranks on GPU 0, collectives over gloo — the
    pool's boxes have one GPU; never a measurement): launched the way the driver launches it, n_gpus == 2, and the job's checksum is
    the sum of what rank 0 and rank 1 compute alone on their own clouds (independent clouds per rank, filter bank broadcast from rank 0)."""
    args = ["--steps", "16", "--warmup", "2", "--extras", "0", "--cpu-seconds", "0", "--full-unet", "0", "--pool", "2", "--in-flight", "2"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LATTICE_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LATTICE_FORCE_DIST"):
        env.pop(k, None)
    port = 29900 + os.getpid() % 90
    two, two_full = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                 "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, env, details=True)
    assert two["n_gpus"] == 2 and two["steps"] == 16 and two["scaling"] == "weak" and two["value"] > 0
    assert 0 < two["ms_per_step_min_over_ranks"] <= two["ms_per_step_max_over_ranks"] == two["ms_per_step"]
    env1 = dict(env)
    env1.pop("LATTICE_BENCH_SHARE_GPU")
    alone = []
    for r in (0, 1):
        line, full = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args,
                                 dict(env1, LATTICE_BENCH_RANK_OFFSET=str(r)), details=True)
        assert line["n_gpus"] == 1
        alone.append(full)
    c0, c1 = alone[0]["config"]["checksum"], alone[1]["config"]["checksum"]
    assert alone[0]["config"]["vertices_per_scan"] != alone[1]["config"]["vertices_per_scan"], "the two ranks must work on different clouds"
    assert abs(c0 - c1) > 1e-6 * abs(c0)
    assert abs(two["config"]["checksum"] - (c0 + c1)) <= 2e-6 * abs(c0 + c1), (two["config"]["checksum"], c0, c1)
    assert two_full["config"]["vertices_per_scan"] == alone[0]["config"]["vertices_per_scan"]  # the line's scan table is rank 0's


_ALONE = {}


def _alone_checksum(offset, args, env1):
    """Checksum of what rank `offset` of a larger job computes, from a one-rank run on that rank's clouds (cached per module)."""
    if offset not in _ALONE:
        _, full = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args,
                              dict(env1, LATTICE_BENCH_RANK_OFFSET=str(offset)), details=True)
        _ALONE[offset] = full["config"]["checksum"]
    return _ALONE[offset]


@pytest.mark.parametrize("ranks", [4, 8])
def test_many_rank_bench_control_flow_on_one_host(ranks):
    """bench.py as the driver launches it at N = 4 and N = 8 (all ranks on GPU 0 over gloo, LATTICE_BENCH_SHARE_GPU=1: the pool has
    one-GPU boxes; never a measurement): rendezvous, per-rank core pinning, the stream probe, the barrier bracket and the final
    reductions have seen N processes on one host, and the job's checksum is the sum of the N single-rank checksums."""
    args = ["--steps", "8", "--warmup", "2", "--extras", "0", "--cpu-seconds", "0", "--full-unet", "0", "--pool", "1", "--in-flight", "1"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LATTICE_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LATTICE_FORCE_DIST", "MASTER_PORT"):
        env.pop(k, None)
    port = 29600 + (os.getpid() + ranks) % 90
    line, full = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
                              "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks)] + args, env,
                             details=True)
    assert line["n_gpus"] == ranks and line["steps"] == 8 and line["value"] > 0
    assert 0 < line["ms_per_step_min_over_ranks"] <= line["ms_per_step_max_over_ranks"] == line["ms_per_step"]
    assert full["host_cores_of_rank0"] <= max(1, (os.cpu_count() or 1) // ranks) or os.environ.get("LATTICE_NO_AFFINITY")
    env1 = dict(env)
    env1.pop("LATTICE_BENCH_SHARE_GPU")
    total = sum(_alone_checksum(r, args, env1) for r in range(ranks))
    assert abs(line["config"]["checksum"] - total) <= 2e-6 * abs(total), (line["config"]["checksum"], total)


def test_driver_command_prints_a_short_parseable_last_line():
    """The driver's exact command (`python3 bench.py --gpus 1 --steps 20 --warmup 5`, every secondary leg on): the LAST stdout line
    is under 4 KB, parses, and carries the headline, `roofline`, `cpu_baseline` and `config.workload`; the per-operator table and
    the other secondary figures are on the `DETAILS ` line in front of it and in bench_details.json."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LATTICE_FORCE_DIST"):
        env.pop(k, None)
    line, full = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"], env, details=True)
    assert line["metric"].startswith("Mpoints/sec") and line["unit"] == "Mpoints/s" and line["n_gpus"] == 1
    assert line["steps"] == 20 and line["warmup"] == 5 and line["value"] > 100.0 and line["vs_baseline"] is None
    assert abs(line["value"] - 120000 / (line["ms_per_step"] * 1e3)) <= 0.01 * line["value"]
    rf = line["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port"
    assert isinstance(line["config"]["workload"], str) and line["config"]["points_per_gpu"] == 120000
    assert "dropped_for_length" not in line
    assert "ops" in full and "stages" in full and "roofline_others" in full
    with open(os.path.join(ROOT, line["details_file"])) as fh:
        assert json.load(fh)["value"] == line["value"]
