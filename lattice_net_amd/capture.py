"""Whole-step capture: run a splat -> conv -> slice style step (forward + backward) as ONE hipGraph replay.

The reference reads the vertex count back to the host after every lattice build (src/Lattice.cu:1320-1352), which no stream
capture can contain.  `CapturedStep` packages the recipe that removes that wait:

  1. calibrate: run the step eagerly once; for every lattice it builds, read the vertex count, set a static row bound
     (count x (1 + row_slack), rounded up to 256, snapped down to a multiple of 16384 when half the slack survives) and, optionally, kd region planes balanced on this cloud;
  2. warm up the static-rows step on a side stream (allocates the table buffers, pinned counters and build workspaces
     outside the capture);
  3. capture the step with torch.cuda.graph;
  4. `launch()` replays it (on `stream`, so that several captured steps — independent scans — can be in flight at once),
     `check()` verifies after a synchronise that every replayed build stayed inside its bounds.

The step function must be self-contained: it reads its inputs from tensors that stay alive (overwrite them IN PLACE to feed
new data of the same shape), uses `lattice.nr_lattice_vertices()` wherever it needs a row count, and must not synchronise.
`CapturedStep` covers the splat -> conv -> slice family on one lattice; `CapturedNetworkStep` (below) covers a whole LNN-style
network: one row bound per lattice level and GroupNorm statistics over the device-side vertex count.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Sequence

import torch

from .lattice import Lattice

__all__ = ["CapturedStep", "CapturedNetworkStep", "concurrent_streams"]

_ROUND_ROWS = 256 * 64  # one 64-vertex tile on every CU of an MI355X


def concurrent_streams(k: int, candidates: int = 16, spin_us: float = 250.0):
    """k streams (the current stream first) whose kernels the GPU really runs side by side.

    HIP multiplexes streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and two streams that share a
    queue run one after the other.  Which streams collide depends on what else created streams in the process — an RCCL
    communicator is enough: three captured scans "in flight" measured 1131 instead of 1330 Mpoints/s with a process group
    initialised, because two of the three streams had landed on one queue.  This helper takes streams from torch's pool and
    keeps those that overlap with every stream picked so far: a spin kernel on both, wall time of the pair against the time of one.
    Falls back to plain pool streams when it cannot find k (the caller still gets k streams)."""
    cur = torch.cuda.current_stream()
    picked = [cur]
    if k <= 1:
        return picked
    cycles = 200_000
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def one(stream, c):
        with torch.cuda.stream(stream):
            ev0.record()
            torch.cuda._sleep(c)
            ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) * 1e3

    import time
    one(cur, 1000)  # (first call: lazy initialisation)
    t = max(one(cur, cycles), 1.0)
    cycles = max(1000, int(cycles * spin_us / t))  # calibrated to ~spin_us

    def wall(streams):
        """Host wall time (best of 3) of one spin kernel on each of `streams`, launched back to back."""
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for st in streams:
                with torch.cuda.stream(st):
                    torch.cuda._sleep(cycles)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e6
            best = dt if best is None else min(best, dt)
        return best

    single = wall([cur])  # the same clock as the pairs below: launch and synchronise overheads (a profiler's, too) cancel

    def overlaps(a, b):
        return wall([a, b]) < 1.5 * single + 20.0  # (two serialised spins take 2 x single)

    spare = []
    for _ in range(candidates):
        if len(picked) >= k:
            break
        s = torch.cuda.Stream()
        if all(overlaps(p, s) for p in picked):
            picked.append(s)
        else:
            spare.append(s)
    while len(picked) < k:  # not enough independent queues: serialised streams are still correct
        picked.append(spare.pop(0) if spare else torch.cuda.Stream())
    return picked


class CapturedStep:
    def __init__(self, step: Callable[[], object], lattices: Sequence[Lattice], *, row_slack: float = 0.06, regions: bool = True,
                 region_indices: Optional[Callable[[], torch.Tensor]] = None, stream: Optional[torch.cuda.Stream] = None,
                 before_capture: Optional[Callable[[], None]] = None, calibration_steps: Optional[Sequence[Callable[[], object]]] = None,
                 more_steps: Optional[Sequence[Callable[[], object]]] = None):
        """`step`: the function to capture.  `lattices`: the Lattice objects it builds (each gets a static row bound).
        `region_indices`: returns the splat-index tensor of the calibration step when kd region planes are wanted (they are
        balanced on the token counts of that build); None: no regions.  `before_capture`: called right before the warm-up and
        the capture to drop references to earlier autograd graphs (a parameter's AccumulateGrad node lives as long as one of
        them does and would run on the stream it was created on, which a capture of another stream cannot include).
        `calibration_steps`: functions like `step` on OTHER clouds; when given, the row bounds (largest vertex count over them
        x (1 + row_slack)) and the region planes (first of them) come from these instead of from `step`'s own cloud.
        `more_steps`: further step functions — the same step reading other input tensors — each captured into a graph of its
        own that shares the lattices, their bounds and workspaces with the first: `launch(i)` replays graph i.  A pool of
        distinct clouds can then rotate through one captured scan without copying inputs."""
        self.step = step
        self.lattices = list(lattices)
        self.stream = stream
        self.vertices = [0] * len(self.lattices)
        for ci, cal in enumerate(list(calibration_steps) if calibration_steps else [step]):
            self.result = cal()  # calibration (eager: reads the vertex counts back)
            for li, lat in enumerate(self.lattices):
                self.vertices[li] = max(self.vertices[li], lat.nr_lattice_vertices())
                if ci == 0 and regions and region_indices is not None:
                    lat.calibrate_regions(region_indices(), vertex_weight=float(os.environ.get("LATTICE_PLANE_VERTEX_WEIGHT", "0")))
        if calibration_steps or (regions and region_indices is not None):
            self.result = step()  # the eager reference result of `step` itself, with the regions in place
        torch.cuda.synchronize()
        for lat, m in zip(self.lattices, self.vertices):
            rows = min(lat.capacity(), ((int(m * (1.0 + row_slack)) + 255) // 256) * 256)
            # The vertex-tiled kernels work in rounds of 256 CUs x 64 rows: a bound a little above a multiple of that costs the
            # fused convolution backward a whole extra sub-tile per CU.  Snap down to the multiple if at least half the slack survives.
            snapped = rows // _ROUND_ROWS * _ROUND_ROWS
            if snapped >= m * (1.0 + 0.5 * row_slack):
                rows = snapped
            lat.set_static_rows(rows)
        self.bounds = [lat.m_hash_table._static_rows for lat in self.lattices]
        steps = [step] + list(more_steps or [])
        if before_capture is not None:
            before_capture()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
            for s in steps[1:]:
                s()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graphs, self.captured_results = [], []
        if before_capture is not None:
            before_capture()  # (once: the captures share torch's capture stream, and their results must stay referenced)
        for s in steps:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.captured_results.append(s())
            self.graphs.append(g)
        self.graph, self.captured = self.graphs[0], self.captured_results[0]

    def launch(self, which: int = 0):
        """One replay of graph `which`, on `stream` if one was given (asynchronous)."""
        if self.stream is None:
            self.graphs[which].replay()
        else:
            with torch.cuda.stream(self.stream):
                self.graphs[which].replay()

    def check(self):
        """After a synchronise: the LAST replayed build of every lattice inside its row bound, no bucket overflow.  Returns the
        vertex counts."""
        return [lat.static_build_report()[0] for lat in self.lattices]

    def release(self):
        """Back to eager mode (the lattices read their vertex counts again)."""
        for lat in self.lattices:
            lat.set_static_rows(None)


class CapturedNetworkStep:
    """A whole-network training step (forward + loss + backward of an LNN-style model: several lattice levels, GroupNorm between the
    lattice operators) as ONE hipGraph replay.

      1. calibrate: one eager `step()` under Lattice.start_level_trace() -> vertex count of every lattice level the network builds;
      2. `lattice.set_static_rows(bound(level 1), coarse_bounds=[bound(level 2), ...])`: every `[M, C]` tensor of the network gets
         its level's bound as height, the GroupNorm kernels read the real count from the lattice's device counter
         (Lattice.rows_device()), rows beyond it stay isolated (no neighbour, no splat index, zero gradient);
      3. warm-up on a side stream, capture, `launch()` = one replay.  Parameter gradients live in the graph's memory pool:
         after a replay `p.grad` holds that step's gradients; the optimizer runs outside the graph, or inside with `optimizer=`.
         Training loop: `with torch.cuda.stream(cap.stream): for ...: cap.launch(); optimizer.step()` (see launch()).

    `step` must read its inputs (positions, values, targets) from tensors that stay alive — overwrite them in place to feed another
    cloud of the same size — set `p.grad = None` itself is NOT needed (done here), and must not synchronise."""

    def __init__(self, step: Callable[[], torch.Tensor], lattice: Lattice, parameters, *, row_slack: float = 0.07,
                 stream: Optional[torch.cuda.Stream] = None, optimizer: Optional[torch.optim.Optimizer] = None,
                 calibration_loaders: Optional[Sequence[Callable[[], None]]] = None):
        """`optimizer` (optional): its step() is captured behind the backward pass, so that a training loop is nothing but replays —
        it must have been built with `capturable=True` (torch.optim.Adam / AdamW / SGD ...: the step count lives on the device).
        `calibration_loaders` (optional): functions that each load one cloud into the tensors `step` reads; the row bounds then
        cover the largest lattice of all of them (a bound calibrated on one cloud drops vertices of a larger one: check())."""
        from .lattice_blocks import new_gn_workspace, reset_gn_workspaces, use_gn_workspace
        self.step, self.lattice, self.stream = step, lattice, stream
        self.optimizer = optimizer
        if optimizer is not None and not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise ValueError("an optimizer captured into the graph must be created with capturable=True")
        self.parameters = list(parameters)
        Lattice.start_level_trace()
        try:
            for load in (calibration_loaders or [None]):
                if load is not None:
                    load()
                self.eager_loss = step()
            torch.cuda.synchronize()
        finally:
            self.levels = Lattice.stop_level_trace()
        lv = sorted(self.levels)
        # bounds are handed to the levels by position (finest first): the traced levels must be exactly lattice.m_lvl, +1, +2, ...
        if not lv or lv != list(range(lattice.m_lvl, lattice.m_lvl + len(lv))):
            raise ValueError(f"the calibration step built lattice levels {lv}, expected consecutive levels starting at {lattice.m_lvl} "
                             "(every builder — splat, distribute, create_verts, coarsen — records its level: did `step` use `lattice`?)")
        bounds = [min(lattice.capacity(), ((int(self.levels[k] * (1.0 + row_slack)) + 255) // 256) * 256) for k in lv]
        self.bounds = dict(zip(lv, bounds))
        for k in lv:
            if self.levels[k] > self.bounds[k]:
                raise ValueError(f"lattice level {k} has {self.levels[k]} vertices, more than its capacity allows as a row bound ({self.bounds[k]})")
        lattice.set_static_rows(bounds[0], coarse_bounds=bounds[1:])
        # this step's own GroupNorm accumulators (never shared with eager launches, never freed while the graph lives)
        self._gn_entry = new_gn_workspace(lattice._dev())

        def guarded():
            reset_gn_workspaces()
            loss = step()
            if optimizer is not None:
                optimizer.step()
            return loss

        # The graph is ALWAYS replayed on the stream it was captured on (launch() joins it with the caller's stream): replaying on
        # another stream — legal, and what CapturedStep does for its ten-node graphs — aborted 4 of 10 forty-step training runs of
        # tools/probes/graph_training_flake.py when eager optimizer steps ran between the replays; on the capture stream 0 of 30.
        # What makes training loops solid is to run the WHOLE loop on that stream (see launch()): DESIGN.md 4.7, "State of the mode".
        side = stream if stream is not None else torch.cuda.Stream()
        self.own_stream = stream is None
        self.stream = side
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), use_gn_workspace(self._gn_entry):
            for _ in range(3 if optimizer is not None else 2):  # torch: three warm-up iterations before capturing an optimizer
                for p in self.parameters:
                    p.grad = None
                guarded()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for p in self.parameters:
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        Lattice.start_static_build_log()
        try:
            with torch.cuda.stream(side), use_gn_workspace(self._gn_entry):
                with torch.cuda.graph(self.graph, stream=side):
                    self.loss = guarded()
        finally:
            # (level, row bound, pinned report word) of every table the graph builds: what check() reads after a replay
            self.builds = Lattice.stop_static_build_log()
        torch.cuda.synchronize()
        self.grads = [p.grad for p in self.parameters]  # this graph's gradient tensors (another capture rebinds p.grad)

    def launch(self):
        """One replay (asynchronous), on the capture stream.  Without a `stream` argument at construction the object owns that
        stream and joins it with the caller's current stream before and after the replay (drop-in for an eager step); with one,
        the caller orders the streams itself (several scans in flight).  RUN TRAINING LOOPS ON THE CAPTURE STREAM
        (`with torch.cuda.stream(cap.stream): ...`): then the eager kernels between two replays (optimizer, input copies) are
        simply queued behind the graph and no cross-stream event joins exist — loops with such joins aborted 25-75 % of 65-step
        runs on this stack (HSA queue exception), loops on one stream 0 of 18 runs of 100-300 steps, and a step is 0.4 ms shorter.  Two bounds on a host that runs ahead: at most two replays of this graph are queued
        behind the running one (event wait), and every 8th launch waits for the launch stream (`LN_GRAPH_SYNC_EVERY`).  (The second
        one dates from the cross-stream loops, whose aborts after ~10^5 graph nodes looked like launch resources that are only
        recycled by a stream-level wait; on the capture stream 4000 SemanticKITTI-shaped and 600 ScanNet-shaped steps — 2 M and 1.5 M
        nodes — run without any such wait.  It costs nothing measurable and stays.)"""
        pending = self.__dict__.setdefault("_pending", [])
        if len(pending) >= 2:
            pending.pop(0).synchronize()
        self._launches = getattr(self, "_launches", 0) + 1
        stream = self.stream
        if self._launches % max(1, int(os.environ.get("LN_GRAPH_SYNC_EVERY", "8"))) == 0:
            stream.synchronize()
        caller = torch.cuda.current_stream()
        join = self.own_stream and caller != stream  # a caller that works on the capture stream itself needs no joins
        if join:
            stream.wait_stream(caller)  # inputs written / parameters updated on the caller's stream
        with torch.cuda.stream(stream):
            self.graph.replay()
            ev = torch.cuda.Event()
            ev.record()
        if join:
            caller.wait_stream(stream)  # the caller's next operations see the loss and the gradients
        pending.append(ev)
        return self.loss

    def check(self):
        """After a synchronise: every lattice level the last replay built stayed inside its static row bound and no bucket
        overflowed.  Returns {level: vertex count}; raises LatticeNetHipError otherwise — the cloud must then be redone eagerly
        (`lattice.set_static_rows(None)`), its replayed result is not usable (vertices beyond a bound were left un-inserted)."""
        from . import _lib
        out = {}
        for level, bound, word in self.builds:
            nr, status = Lattice.decode_report(word[0])
            if status & (_lib.LN_STATUS_BUCKET_OVERFLOW | _lib.LN_STATUS_TABLE_FULL | _lib.LN_STATUS_KEY_RANGE):
                raise _lib.LatticeNetHipError(f"replayed build of lattice level {level} failed (status bits {status}): redo this cloud eagerly")
            if nr > bound:
                raise _lib.LatticeNetHipError(f"lattice level {level} of this cloud has {nr} vertices, its static row bound is {bound}: "
                                              "rows beyond the bound were dropped; redo this cloud eagerly or re-capture with larger bounds")
            out[level] = max(out.get(level, 0), nr)
        return out

    def bind_gradients(self):
        """Binds this graph's gradient tensors as `p.grad` (they hold the last replay's gradients; another capture over the same
        parameters rebinds `p.grad` to its own).  (Copying them into persistent buffers with torch._foreach_copy_ first faulted
        on the 9 M-parameter ScanNet-shaped model on this stack; the optimizer reads the graph's tensors directly.)"""
        for p, g in zip(self.parameters, self.grads):
            p.grad = g

    @staticmethod
    def sum_gradients(captures):
        """Adds the gradients of captures[1:] into those of captures[0] (in place: the next replay overwrites them) and binds the
        result — K scans per optimizer step, all replays joined on the current stream."""
        first = captures[0]
        for cap in captures[1:]:
            pairs = [(a, b) for a, b in zip(first.grads, cap.grads) if a is not None and b is not None]
            torch._foreach_add_([a for a, _ in pairs], [b for _, b in pairs])
        first.bind_gradients()
