"""Data-parallel sharding of independent point clouds: one process per GPU, cloud i -> rank i mod G,
no data-path collective (each cloud owns its hash table).  torch.distributed ("nccl" = RCCL over xGMI
on ROCm, "gloo" on CPU for tests) is used only to broadcast parameters once and to reduce timings /
checksums."""
from __future__ import annotations

import os
from typing import Iterable, List

import torch


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: str, device=None):
    """Initialises torch.distributed from the launcher's environment; returns the module or None for 1 rank."""
    world, rank, _ = env_world()
    if world <= 1 and not os.environ.get("LATTICE_FORCE_DIST"):  # the override exercises the RCCL calls on one rank
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kwargs = {}
    if "MASTER_PORT" not in os.environ:
        # Launchers (torch.distributed.run, the tests) set the port.  Without one every rank must still arrive at the SAME port
        # whatever started it (one shell or ssh session per rank, container entrypoints): a fixed default, outside torchrun's own
        # 29500.  Two jobs side by side on one host have to be given different MASTER_PORTs by whoever starts them.
        # LATTICE_JOB_ID (any integer) moves the default so that two launcher-less jobs on one host do not meet in one store.
        job = os.environ.get("LATTICE_JOB_ID", "0")
        try:
            offset = int(job) % 2000
        except ValueError:  # (any string works: a stable hash of it)
            import zlib
            offset = zlib.crc32(job.encode()) % 2000
        os.environ["MASTER_PORT"] = str(29511 + offset)
    import datetime
    kwargs["timeout"] = datetime.timedelta(seconds=int(os.environ.get("LATTICE_RENDEZVOUS_TIMEOUT_S", "600")))
    if backend == "nccl" and device is not None:
        kwargs["device_id"] = device
    try:
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    except Exception as exc:
        raise RuntimeError(f"rendezvous of rank {rank}/{world} at {os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']} failed "
                           f"({type(exc).__name__}: {exc}); every rank needs the same MASTER_ADDR / MASTER_PORT") from exc
    return dist


def _parse_cpulist(text: str) -> List[int]:
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes(sysfs_root: str = "/sys") -> List[int]:
    """NUMA node of every AMD GPU of the host in PCI-address order (the order HIP enumerates devices in by default), read from
    sysfs — no GPU call.  Display controllers (class 0x03..) and processing accelerators (class 0x12..) of vendor 0x1002; -1 where
    the kernel does not know the node."""
    base = os.path.join(sysfs_root, "bus", "pci", "devices")
    found = []
    try:
        names = sorted(os.listdir(base))
    except OSError:
        return []
    for name in names:
        try:
            with open(os.path.join(base, name, "vendor")) as f:
                if f.read().strip().lower() != "0x1002":
                    continue
            with open(os.path.join(base, name, "class")) as f:
                cls = f.read().strip().lower()
            if not (cls.startswith("0x03") or cls.startswith("0x12")):
                continue
            with open(os.path.join(base, name, "numa_node")) as f:
                found.append(int(f.read().strip()))
        except (OSError, ValueError):
            continue
    return found


def pin_launch_thread(local_rank: int, local_world: int, sysfs_root: str = "/sys") -> List[int]:
    """Pins the calling (kernel-launching) process to cores of its own BEFORE the first GPU call: a step of the hot path is ~100 us
    of GPU time, so the launch thread is on the critical path, and eight unpinned ranks migrate across sockets.
    Where sysfs names the NUMA node of every GPU (and there are at least `local_world` GPUs), rank r takes the cores of GPU r's node
    that this process may use, divided among the ranks whose GPUs share that node; otherwise the allowed cores are cut into
    `local_world` contiguous slices.  LATTICE_NO_AFFINITY=1 opts out; a no-op for one rank or without sched_setaffinity.
    Returns the cores in use afterwards."""
    have = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    if local_world <= 1 or os.environ.get("LATTICE_NO_AFFINITY") or len(have) < 2 * local_world:
        return have
    mine = None
    nodes = gpu_numa_nodes(sysfs_root)
    # sysfs lists ALL AMD GPUs in PCI order; a job that was handed a subset (HIP_/ROCR_/CUDA_VISIBLE_DEVICES) sees them renumbered.
    # Integer lists are mapped through (rank r uses physical GPU list[r]); anything else (UUIDs, several variables at once) is not
    # trusted: contiguous slices of the allowed cores instead of the NUMA node of the WRONG GPUs.
    visible = [os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if os.environ.get(k)]
    if visible:
        try:
            ids = [int(x) for x in visible[0].split(",")] if len(set(visible)) == 1 else None
        except ValueError:
            ids = None
        nodes = [nodes[i] for i in ids] if (ids and all(0 <= i < len(nodes) for i in ids)) else []
    if len(nodes) >= local_world and all(n >= 0 for n in nodes[:local_world]):
        node = nodes[local_rank]
        try:
            with open(os.path.join(sysfs_root, "devices", "system", "node", f"node{node}", "cpulist")) as f:
                node_cpus = [c for c in _parse_cpulist(f.read()) if c in set(have)]
        except (OSError, ValueError):
            node_cpus = []
        sharers = [r for r in range(local_world) if nodes[r] == node]
        per = len(node_cpus) // max(len(sharers), 1)
        if per >= 1:
            k = sharers.index(local_rank)
            mine = node_cpus[k * per:(k + 1) * per]
    if not mine:
        per = len(have) // local_world
        mine = have[local_rank * per:(local_rank + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return have
    return mine


def clouds_of_rank(num_clouds: int, world: int, rank: int) -> List[int]:
    """Cloud i is processed by rank i mod world."""
    return [i for i in range(num_clouds) if i % world == rank]


def cloud_seed(rank: int, step_cloud: int = 0) -> int:
    """Seed of the synthetic cloud a rank works on (distinct clouds per rank: weak scaling)."""
    return rank + 1000 * step_cloud


def broadcast_parameters(dist, tensors: Iterable[torch.Tensor], src: int = 0):
    if dist is None:
        return
    for t in tensors:
        dist.broadcast(t, src=src)


def max_over_ranks(dist, value: float, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_max_over_ranks(dist, value: float, device):
    """(min, max) of `value` over the ranks (one all-reduce of the pair [-v, v] with MAX)."""
    t = torch.tensor([-value, value], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(-t[0].item()), float(t[1].item())


def gather_sum(dist, value: float, device) -> float:
    if dist is None:
        return float(value)
    world = dist.get_world_size()
    outs = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
    dist.all_gather(outs, torch.tensor([value], dtype=torch.float64, device=device))
    return float(sum(o.item() for o in outs))


def barrier(dist):
    if dist is not None:
        dist.barrier()


def allreduce_gradients(dist, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20):
    """Data-parallel gradient exchange of the training step: every rank has run forward/backward
    on its own cloud; gradients are averaged with bucketed all-reduces (RCCL rings over xGMI are per-link bound, so few
    large messages: the whole LNN is < 4 MB, i.e. one bucket).  Parameters that received no gradient on this rank
    (none in LNN) contribute zeros so that all ranks issue the same collectives."""
    if dist is None:
        return
    world = dist.get_world_size()
    params = [p for p in params if p.requires_grad]
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
        bucket, size = [], 0

    for p in params:
        nbytes = p.numel() * p.element_size()
        if bucket and (size + nbytes > bucket_bytes or p.dtype != bucket[0].dtype):
            flush()
        bucket.append(p)
        size += nbytes
    flush()


def allreduce_sum_(dist, t: torch.Tensor) -> torch.Tensor:
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
