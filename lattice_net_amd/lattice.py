"""`Lattice` / `HashTable`: Python host side of the MI355X permutohedral-lattice backend.

Mirrors the pybind11 classes of the reference (src/PyBridge.cxx:33-113) method for method —
same names, argument meaning and return shapes — on top of the C ABI in
include/latticenet_hip.h.  Host orchestration follows src/Lattice.cu (cited per method); tensors
are plain torch tensors on a ROCm device, kernels run on torch's current HIP stream.

Differences that are deliberate:
  * errors raise Python exceptions instead of glog CHECK aborts;
  * vertex numbering is deterministic: hash-slot order by default, first occurrence in (point, remainder) order under
    set_row_order("canonical") (the reference's is thread-arrival order);
  * the 9-probe neighbour traversal is done ONCE per (query, neighbour, dilation, flip) and the
    [M, E] neighbour list is cached with the table structure, then shared by im2row,
    im2rowindices, row2im, the fused convolution and its backward;
  * `convolve_im2row_standalone` never materialises the [M, E*V] rowified tensor.
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
import re
import weakref
from typing import Optional, Sequence

import torch
import torch.utils.weak

from . import _lib

__all__ = ["Lattice", "HashTable", "set_row_order", "get_row_order", "set_hash_capacity_policy", "set_slot_order", "set_bank_cache", "set_deterministic"]


_SIZE_CACHE = {}
_FORCE_ATOMIC_BUILD = os.environ.get("LATTICE_BUILD_PATH", "") == "atomic"  # A/B switch: skip the bucketed build

# Vertex numbering of builds that start from a cleared table (begin_splat + splat / distribute / just_create_verts):
#   "slot"      (default) rows bucket by bucket (runs of hash slots; first occurrence inside a bucket) — deterministic, two launches.  The reference numbers its
#               vertices in thread-arrival order (HashTableGPU.cuh:454), so nothing it computes depends on the numbering;
#   "canonical" rows by first occurrence in (point, remainder) order — what a serial run of the reference produces and
#               what the golden vectors hold; costs a relabelling pass behind the build (ln_canonicalize).
_ROW_ORDER = [os.environ.get("LATTICE_ROW_ORDER", "slot")]


# How many slots of the table a build hashes into: "tokens" (default) = min(capacity, max(16384, 2 x tokens of the build)) for
# builds that start from a cleared table, "full" = always the cfg's capacity (what the reference does).
_HASH_POLICY = [os.environ.get("LATTICE_HASH_CAPACITY", "tokens")]
_STATIC_SLOTS_PER_ROW = float(os.environ.get("LATTICE_STATIC_SLOTS_PER_ROW", "2.3"))  # static-rows mode: slots hashed into per bounded row


# What kd region planes (Lattice.set_region_planes) steer:
#   "hash"  (default) which XCD walks which CSR segments; slots stay hashed over the whole table (rounds 2-5);
#   "space" the SLOT function as well (LnTable.slot_map, csrc/ln_common.h): a key starts probing in the slot run of its kd region, rows
#           are numbered bucket by bucket, so the rows of the table follow space, and the convolution kernels let XCD x work on the
#           rows of region x.  Measured on the C3 chain (round 6, profiles/r6_pmc_traffic*.json): HBM traffic of the two convolutions
#           37.9 -> 20.6 MB and 53.5 -> 36.2 MB, of the whole chain 310 -> 262 MB — and NO gain in time: the kernels are latency-bound
#           chains, alone and with four scans in flight (1345 against 1391 Mpoints/s).  What it costs: the in-region hash has to be a
#           stirred one (the reference's raw hash is a low-discrepancy sequence over lattice keys for SOME moduli — 1.89 probes per
#           unsuccessful retrieval at load 0.47 where a random hash gives 2.26, p99 6 against 11 — and a poor one for others), and the
#           neighbour traversal, half of whose lookups are for absent vertices, waits for the slowest lane of every wave: 10.3 -> 15.6 us.
#           Kept as an option for workloads that ARE bound by HBM traffic (wide value rows, many scans in flight).
_SLOT_ORDER = [os.environ.get("LATTICE_SLOT_ORDER", "hash")]
_PLANES_KEEPALIVE = []
_SLOT_MAPS = {}
# share of the slots handed out equally to the 8 leaves whatever their calibrated vertex share (the rest follows the shares): slack for
# clouds that differ from the calibration cloud.  0.3 left the two hot leaves of a LiDAR scan (1 % of the vertices each) at load 0.1 and
# pushed the others from 0.47 to 0.51 — every unsuccessful retrieval pays for that (round 6: tools/probes/r6_probe_lengths.py).
_SLOT_RUN_FLOOR = float(os.environ.get("LATTICE_SLOT_RUN_FLOOR", "0.08"))


def _make_slot_map(planes, shares, capacity: int, device):
    """(int32[LN_SLOT_MAP_INTS] device tensor, largest bucket) for LnTable.slot_map, or (None, 0) when the table has too few buckets: every
    leaf gets the same number of buckets (the planes balance tokens, so the bucket workgroups of a build see equal loads) and a
    bucket SIZE in proportion to the share of the vertices it is expected to hold, blended with an equal split so that a cloud that
    differs from the calibration cloud still fits (equal load factor everywhere = equal probe lengths)."""
    import numpy as np
    nbk = int(_lib.load().ln_table_bucket_count(int(capacity)))
    G = _lib.LN_XCD_GROUPS
    if nbk < 2 * G:
        return None, 0
    bpl = nbk // G  # (the C side launches G * bpl bucket workgroups over a mapped table)
    units = int(capacity) // bpl  # the bucket sizes of the 8 leaves may add up to this
    sh = np.full((G,), 1.0 / G) if shares is None else np.asarray(shares, np.float64)
    sh = (1.0 - _SLOT_RUN_FLOOR) * sh / max(sh.sum(), 1e-30) + _SLOT_RUN_FLOOR / G
    sb = np.maximum(np.floor(sh * units).astype(np.int64), min(16, units // G))
    while sb.sum() > units:
        sb[int(np.argmax(sb))] -= 1
    left = units - int(sb.sum())
    sb[np.argsort(-sh)[:left]] += 1  # (left < 8)
    if int(sb.min()) < 1:
        return None, 0
    m = np.zeros((_lib.LN_SLOT_MAP_INTS,), np.int32)
    m[0:7] = np.asarray(planes, np.int64).astype(np.int32)
    m[7] = bpl
    m[8:8 + G + 1] = np.concatenate([[0], np.cumsum(sb * bpl)])
    m[17:17 + G] = sb
    t = torch.from_numpy(m).to(device)
    _PLANES_KEEPALIVE.append(t)  # captured graphs hold the raw pointer (128 bytes per map)
    return t, int(sb.max())
# Slice in the order of the build's CSR over space-ordered tables (ln_slice_forward_ordered).  OFF by default: measured on the shuffled
# C3 scan the value rows are then fetched 1.9x instead of 3.2x, but the (index, weight) pairs of a point — 16 + 16 bytes at a random
# place of two arrays — cost a 64-byte sector each where the input-order walk reads them as streams: 28.5 MB fetched against 22.8 MB,
# 15.0 us against 8.3 us (profiles/r6_pmc_traffic_slice_orders.json).  Worth switching on only for clouds whose input order is random AND
# whose rows are wide (the value rows then dominate).
_ORDERED_SLICE = [os.environ.get("LATTICE_ORDERED_SLICE", "0") == "1"]


def set_slot_order(order: str) -> str:
    """Selects what region planes steer in subsequent builds ("space" or "hash"); returns the previous setting."""
    if order not in ("space", "hash"):
        raise ValueError(f"slot order must be 'space' or 'hash', got {order!r}")
    prev, _SLOT_ORDER[0] = _SLOT_ORDER[0], order
    return prev


# How many scans the caller keeps in flight on this GPU (own lattice and stream each: bench.py's throughput mode, a data loader that
# builds ahead).  Passed to the library with every build (ln_build_concurrency): overlapping builds over small buckets take narrower
# bucket-pass workgroups.  A speed hint only; a captured graph keeps what was set when it was captured.
_SCANS_IN_FLIGHT = [max(1, int(os.environ.get("LATTICE_SCANS_IN_FLIGHT", "1")))]


def set_scans_in_flight(n: int) -> int:
    prev, _SCANS_IN_FLIGHT[0] = _SCANS_IN_FLIGHT[0], max(1, int(n))
    return prev


# Run-to-run identical floating-point results of everything that sums over the tokens of a vertex (splat values, position means,
# slice / gather / slice_classify gradients).  The default kernels are deterministic in WHAT they sum and free in the ORDER: a
# token's place in its vertex's list comes from an atomic counter of the build, and rows with several segments combine through float
# atomics.  With this switch every build sorts the token lists (LN_BUILD_SORTED_CSR) and every reduce walks a row with one lane
# group in list order (LnCsr.dense & 2) — slower on hot vertices; meant for tests and for debugging a training run.
_DETERMINISTIC = [os.environ.get("LATTICE_DETERMINISTIC", "0") == "1"]


def set_deterministic(on: bool) -> bool:
    prev, _DETERMINISTIC[0] = _DETERMINISTIC[0], bool(on)
    return prev


def set_hash_capacity_policy(policy: str) -> str:
    if policy not in ("tokens", "full"):
        raise ValueError(f"hash capacity policy must be 'tokens' or 'full', got {policy!r}")
    prev, _HASH_POLICY[0] = _HASH_POLICY[0], policy
    return prev


def set_row_order(order: str) -> str:
    """Selects the vertex numbering of subsequent builds ("slot" or "canonical"); returns the previous setting."""
    if order not in ("slot", "canonical"):
        raise ValueError(f"row order must be 'slot' or 'canonical', got {order!r}")
    prev, _ROW_ORDER[0] = _ROW_ORDER[0], order
    return prev


def get_row_order() -> str:
    return _ROW_ORDER[0]


def _build_sizes(tokens: int, capacity: int):
    """(build workspace bytes, csr workspace bytes, max segments) — pure functions of the sizes, memoised."""
    key = (tokens, capacity)
    hit = _SIZE_CACHE.get(key)
    if hit is None:
        lib = _lib.load()
        hit = (int(lib.ln_build_workspace_bytes(tokens, capacity)), int(lib.ln_csr_workspace_bytes(tokens, capacity)),
               int(lib.ln_csr_max_segments(tokens, capacity)))
        _SIZE_CACHE[key] = hit
    return hit


class _PinnedReportPool:
    """8-byte aligned report slots (16 bytes each) in pinned host memory for the tables of the static-rows mode.  A captured
    graph keeps writing to the slot of every table it built for as long as it is replayed, long after the per-forward table
    object is gone: slots handed out during a stream capture are never reused; slots handed out eagerly go back to the free
    list when their table dies.  Blocks of 256 slots are allocated on demand, outside captures only, and never freed."""

    def __init__(self):
        self.blocks = []
        self.free = []

    def take(self, owner) -> torch.Tensor:
        capturing = torch.cuda.is_current_stream_capturing()
        if not self.free:
            if capturing:
                raise _lib.LatticeNetHipError("static-rows mode: the pinned report pool is exhausted inside a stream capture; run the "
                                              "step eagerly once before capturing it (the pool grows outside captures only)")
            block = torch.zeros((256, 4), dtype=torch.int32, pin_memory=True)
            self.blocks.append(block)
            self.free.extend((len(self.blocks) - 1, r) for r in reversed(range(256)))
        b, r = self.free.pop()
        if not capturing:
            weakref.finalize(owner, self.free.append, (b, r))
        return self.blocks[b][r]


_STATIC_PINNED = _PinnedReportPool()


def _static_pinned(owner) -> torch.Tensor:
    return _STATIC_PINNED.take(owner)


def _require_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise ValueError(f"{name} must live on a ROCm/HIP device (got {t.device}); the lattice backend has no CPU path")


_DENSE_TOKENS_PER_VERTEX = float(os.environ.get("LATTICE_DENSE_TOKENS_PER_VERTEX", "16"))  # LnCsr.dense from here on (0: always, 1e9: never)
# Keep the split bank of an unchanged FROZEN filter between convolve_im2row_standalone calls.  Opt-in (LATTICE_BANK_CACHE=1 or
# set_bank_cache(True)): the only change detector is (data_ptr, _version), which writes through `.data`, checkpoint patching and fused
# optimizers do not move — a caller that enables it promises not to modify frozen filters behind torch's back.  Entries live in a
# weak-keyed dictionary (nothing rides in Parameter.__dict__ into pickles) and remember the stream that produced them: a hit from
# another stream is ignored (the bank may still be being written there).
_BANK_CACHE = [os.environ.get("LATTICE_BANK_CACHE", "0") == "1"]
_BANK_ENTRIES = torch.utils.weak.WeakTensorKeyDictionary()  # (keys compared by identity: `==` on tensors is elementwise)


def set_bank_cache(on: bool) -> bool:
    prev, _BANK_CACHE[0] = _BANK_CACHE[0], bool(on)
    if not on:
        _BANK_ENTRIES.clear()
    return prev


def _pad4(words: int) -> int:
    return (int(words) + 3) & ~3


class _TableStorage:
    """Structure buffers shared shallowly between lattice clones (Lattice.cu:88-92)."""

    _uids = itertools.count(1)

    def __init__(self, capacity: int, pos_dim: int, device, spare_row_width: int = 0):
        self.uid = next(_TableStorage._uids)  # never reused (unlike id()): cache keys may outlive the storage they name
        self.capacity = int(capacity)
        self.pos_dim = int(pos_dim)
        self.device = device
        self.fresh_counters = None  # int32[2] zeros / float32 [1, spare_row_width] zeros carved from the same allocation, for the
        self.fresh_row = None       # table that is created together with this storage (one launch instead of five fills)
        # Slots actually hashed into (0 = not chosen yet: all of them).  The cfg's capacity is an upper bound chosen for the largest
        # cloud (5 M slots for ScanNet scenes that fill 2.5 % of them); a build that starts from a cleared table hashes into
        # min(capacity, 2 x its tokens) slots instead, so that clearing / emitting the slot range costs what the cloud needs
        # (Lattice._build).  Slot positions are internal — only row ids are reference-visible.  Never shrinks: the rows every
        # earlier build wrote stay inside the range a later clear covers.
        self.hash_capacity = 0
        # keys = 0 (rows beyond a build's reach are never touched) | entries = -1 (empty: slots beyond the hashed range stay so) |
        # slot_cnt = 0 (scratch that is all-zero between builds) | counters | placeholder row: ONE allocation, ONE launch
        capacity, pos_dim = self.capacity, self.pos_dim
        kw, ew, cw, sw = _pad4(capacity * pos_dim), _pad4(capacity), _pad4(capacity), _pad4(max(int(spare_row_width), 0))
        arena = torch.empty((kw + ew + cw + 4 + sw,), dtype=torch.int32, device=device)
        _lib.check(_lib.load().ln_arena_init(_lib.ptr(arena), arena.numel(), kw, kw + capacity, _lib.stream_ptr(arena.device)), "ln_arena_init")
        self.keys = arena[: capacity * pos_dim].view(capacity, pos_dim)
        self.entries = arena[kw: kw + capacity]
        self.slot_cnt = arena[kw + ew: kw + ew + capacity]
        self.fresh_counters = arena[kw + ew + cw: kw + ew + cw + 2]
        if spare_row_width > 0:
            self.fresh_row = arena[kw + ew + cw + 4: kw + ew + cw + 4 + spare_row_width].view(torch.float32).view(1, spare_row_width)
        self.slot_keys = torch.empty((capacity,), dtype=torch.int64, device=device)
        self.slot_tok = torch.empty((capacity,), dtype=torch.int32, device=device)
        # every table built from positions holds lattice points only: the wide packed-key format (LnTable.key_format); the target
        # of the key-based coarsening (create_coarse_verts) receives halved fine keys and switches to the raw format
        self.key_format = _lib.LN_KEYS_LATTICE
        self.version = 0
        self.nbr_cache = {}
        self.csr_cache = {}
        self.replay = None  # [rebuild on the atomic path, then the work queued behind the build], see Lattice._build
        self.planes = None  # int32[8] device tensor (7 planes + padding): kd split of key space for the NEXT build that clears, None = no regions
        self.plane_values = None  # the same 7 ints on the host, and the calibrated share of the vertices each of the 8 leaves holds
        self.leaf_shares = None
        self.slot_map = None      # int32[LN_SLOT_MAP_INTS] device tensor the CONTENTS of the table were inserted under (LnTable.slot_map):
        self.slot_map_sb_max = 0  # adopted by every build that starts from a cleared table, kept by incremental builds and retrievals
        self.row_regions = None  # int32[16] device tensor: first row of each kd region, written by bucketed builds over space-ordered slots
        self.rows_follow_space = False  # the last build numbered the rows region by region (row_regions is current)

    def adopt_planes(self):
        """Called where the table is known to be empty (a clear, a build that clears first) and its hashed capacity is settled: the
        slot map of the contents to come = the planes set last + a slot run per leaf in proportion to the leaf's calibrated vertex share."""
        self.rows_follow_space = False
        if self.planes is None or _SLOT_ORDER[0] != "space":
            self.slot_map, self.slot_map_sb_max = None, 0
            return
        key = (self.plane_values, self.leaf_shares, self.hashed())
        hit = _SLOT_MAPS.get(key)
        if hit is None:
            hit = _make_slot_map(self.plane_values, self.leaf_shares, self.hashed(), self.device)
            if len(_SLOT_MAPS) > 256:
                _SLOT_MAPS.clear()  # (the tensors stay alive in _PLANES_KEEPALIVE: captured graphs hold raw pointers)
            _SLOT_MAPS[key] = hit
        self.slot_map, self.slot_map_sb_max = hit
        if self.slot_map is not None and self.row_regions is None:
            self.row_regions = torch.zeros((16,), dtype=torch.int32, device=self.device)

    def hashed(self) -> int:
        return self.hash_capacity or self.capacity

    def clone(self) -> "_TableStorage":
        s = _TableStorage.__new__(_TableStorage)
        s.uid = next(_TableStorage._uids)
        s.capacity, s.pos_dim, s.device = self.capacity, self.pos_dim, self.device
        s.fresh_counters = s.fresh_row = None
        s.hash_capacity = self.hash_capacity
        s.key_format = self.key_format
        s.keys = self.keys.clone()
        s.entries = self.entries.clone()
        s.slot_keys = self.slot_keys.clone()
        s.slot_tok = self.slot_tok.clone()
        s.slot_cnt = self.slot_cnt.clone()
        s.version = 0
        s.nbr_cache = {}
        s.csr_cache = {}
        s.replay = None
        s.planes = self.planes
        s.plane_values, s.leaf_shares = self.plane_values, self.leaf_shares
        s.slot_map, s.slot_map_sb_max = self.slot_map, self.slot_map_sb_max
        s.row_regions = None if self.row_regions is None else self.row_regions.clone()
        s.rows_follow_space = self.rows_follow_space
        return s

    def touch(self):
        """Structure changed: drop cached neighbour lists."""
        self.version += 1
        self.nbr_cache.clear()
        self.csr_cache.clear()


class HashTable:
    """Host owner of the table tensors (src/HashTable.cu:11-115)."""

    def __init__(self, capacity: int):
        self.m_capacity = int(capacity)
        self._storage: Optional[_TableStorage] = None
        self.m_values_tensor: Optional[torch.Tensor] = None
        self._counters: Optional[torch.Tensor] = None  # int32[2]: [nr_filled, status] on the device (builds mirror them into pinned host memory)
        self.m_nr_filled_is_dirty = True
        self.m_nr_filled = -1
        self._pos_dim_hint = -1  # dimensions known before any CAP-sized buffer exists
        self._val_dim_hint = 0
        self._static_rows = None  # capture-safe mode (Lattice.set_static_rows): fixed row bound instead of the host readback
        self._batch = (0, 0)  # (points per cloud, key step) of a batch of independent clouds in this table (Lattice.set_cloud_batch)

    def flush(self):
        """Issues a deferred begin_splat clear, if any (every reader of table state goes through this)."""
        if getattr(self, "_clear_pending", False):
            self.clear()

    # -- reference-visible tensors (PyBridge.cxx:35-36) --
    @property
    def m_keys_tensor(self):
        self.flush()
        return None if self._storage is None else self._storage.keys

    @property
    def m_nr_filled_tensor(self):  # PyBridge.cxx:36
        self.flush()
        return None if self._counters is None else self._counters[0:1]

    @property
    def m_entries_tensor(self):
        self.flush()
        return None if self._storage is None else self._storage.entries

    def init(self, pos_dim: int, val_dim: int, device):  # HashTable.cu:21-47
        self._storage = _TableStorage(self.m_capacity, pos_dim, device)
        self.m_values_tensor = torch.zeros((self.m_capacity, val_dim), dtype=torch.float32, device=device)
        self._zero_beyond = (weakref.ref(self.m_values_tensor), 0)  # (weak reference to the tensor, row from which it is known to be zero): see Lattice._build
        self._counters = self._storage.fresh_counters
        self.m_nr_filled_is_dirty = True
        self.clear(lazy=True)  # rides in the first build call; every other reader flushes it

    def is_initialized(self) -> bool:
        return self._storage is not None

    def c_table(self) -> _lib.LnTable:
        s = self._storage
        if s is None:
            raise _lib.LatticeNetHipError("hash table is not initialised (no splat / create_verts happened yet)")
        if getattr(self, "_clear_pending", False):
            self.clear()  # a deferred begin_splat must land before anybody looks at the table
        if getattr(self, "_pinned", None) is None:
            # builds write {nr_filled, status} straight into this pinned (device-visible) pair from their scan kernel
            if self._static_rows is not None:
                # static-rows mode: a captured graph keeps writing to this address for as long as it is replayed, long after
                # this (per-forward) table object is gone — the buffer comes from a pool that is never freed
                self._pinned = _static_pinned(self)
            else:
                self._pinned = torch.zeros((4,), dtype=torch.int32, pin_memory=True)
            self._pinned_np = self._pinned.numpy().view("int64")  # same memory: the host polls the build's 64-bit report word
            self._readback_event = torch.cuda.Event()
        key = (s.uid, self._counters.data_ptr(), self._static_rows, s.hashed(), None if s.slot_map is None else s.slot_map.data_ptr(), self._batch)
        if getattr(self, "_c_table_key", None) == key:
            return self._c_table
        self._c_table_key = key
        self._c_table = self._make_c_table(s)
        return self._c_table

    def _make_c_table(self, s) -> _lib.LnTable:
        return _lib.LnTable(s.hashed(), s.pos_dim, s.slot_keys.data_ptr(), s.slot_tok.data_ptr(), s.slot_cnt.data_ptr(), s.entries.data_ptr(),
                            s.keys.data_ptr(), self._counters.data_ptr(), self._counters.data_ptr() + 4, self._pinned.data_ptr(), 0,
                            s.key_format, self._static_rows or 0, _lib.ptr(s.slot_map), s.slot_map_sb_max, int(self._batch[0]), int(self._batch[1]),
                            _lib.ptr(s.row_regions) if s.slot_map is not None else None)

    def clear(self, lazy: bool = False):  # HashTable.cu:49-57, one launch instead of four fill_ kernels
        """`lazy=True` only records that a clear is due: the next build issues it inside its own C call (no host
        round trip between the two launches); anything else that reads the table flushes it first."""
        if not self.is_initialized():
            return
        if lazy:
            self._clear_pending = True
            self._storage.touch()
            self.m_nr_filled_is_dirty = True
            self._readback_pending = False
            return
        self._clear_pending = False
        lib = _lib.load()
        v = self.m_values_tensor
        if v is not None and not (v.is_contiguous() and v.data_ptr() % 16 == 0):
            v.zero_()
            v = None
        self._storage.adopt_planes()  # the table is empty behind this launch: the slot function may change
        t = self.c_table()
        _lib.check(lib.ln_table_clear(C.byref(t), _lib.ptr(v), 0 if v is None else v.numel(),
                                      _lib.stream_ptr(self._storage.device)), "ln_table_clear")
        self._storage.touch()
        self.m_nr_filled_is_dirty = True
        self._readback_pending = False

    def arm_count_readback(self):
        """Call right before a build is issued (after c_table()): gives the build a fresh sequence number, which its scan
        kernel stores behind the counters; read_counters() spins until it sees that number."""
        self._host_seq = (getattr(self, "_host_seq", 0) % 0xFFFFFF) + 1  # (24 bits travel back in the report word)
        self._c_table.host_seq = self._host_seq

    def start_count_readback(self):
        """Marks the end of a build on the stream.  The build's scan kernel writes {nr_filled, status} into the pinned
        triple and then sets its third word; nr_lattice_vertices() spins on that word, so it returns as soon as the scan
        has run — before the rest of the build and anything queued behind it."""
        if self._counters is None or getattr(self, "_pinned", None) is None:
            return
        self._readback_event.record(torch.cuda.current_stream(self._counters.device))
        self._readback_pending = True

    def read_counters(self):
        """[nr_filled, status]; one wait (a spin on the pinned flag if a build armed one, else a device read)."""
        self.flush()
        if getattr(self, "_readback_pending", False):
            self._readback_pending = False
            arr = self._pinned_np  # int64 view: nr_filled | status << 32 | seq << 40, ONE device store per build
            spins = 0
            seq = self._host_seq
            word = int(arr[0])
            while ((word >> 40) & 0xFFFFFF) != seq:
                spins += 1
                if spins > 20000:  # ~ms: fall back to the event (a build that failed to launch, exotic memory settings)
                    self._readback_event.synchronize()
                    word = int(arr[0])
                    break
                word = int(arr[0])
            return [word & 0xFFFFFFFF, (word >> 32) & 0xFF]
        return self._counters.tolist()

    def take_pending_clear(self):
        """(values tensor to zero or None, flag) for a build that will issue the deferred clear itself."""
        if not getattr(self, "_clear_pending", False):
            return None, False
        self._clear_pending = False
        v = self.m_values_tensor
        if v is not None and not (v.is_contiguous() and v.data_ptr() % 16 == 0):
            v.zero_()
            v = None
        return v, True

    def clear_only_values(self):  # HashTable.cu:59-64
        self.flush()
        if self.is_initialized() and self.m_values_tensor is not None:
            self.m_values_tensor.zero_()

    def pos_dim(self) -> int:
        return self._storage.pos_dim if self._storage is not None else self._pos_dim_hint

    def val_dim(self) -> int:  # HashTable.cu:102-104: defined as values.size(1)
        return int(self.m_values_tensor.shape[1]) if self.m_values_tensor is not None else self._val_dim_hint

    def capacity(self) -> int:
        return self._storage.capacity if self._storage is not None else self.m_capacity

    def set_values(self, new_values: torch.Tensor):  # HashTable.cu:112-115
        self.flush()  # a deferred clear must hit the OLD values tensor, not the one being installed
        self.m_values_tensor = new_values.contiguous()
        self._zero_beyond = None  # content unknown: the next clear covers the whole tensor


def _parse_lattice_cfg(path: str) -> dict:
    """Reads the `lattice_gpu: { ... }` block of a configuru CFG file (Lattice.cu:107-132)."""
    with open(path, "r") as f:
        text = f.read()
    text = re.sub(r"//[^\n]*", "", text)
    m = re.search(r"lattice_gpu\s*:\s*\{(.*?)\}", text, flags=re.S)
    if not m:
        raise ValueError(f"{path}: no lattice_gpu block")
    out = {}
    for key, val in re.findall(r"(\w+)\s*:\s*(\"[^\"]*\"|[^\s,]+)", m.group(1)):
        out[key] = val.strip('"')
    return out


class Lattice:
    """Drop-in for `latticenet.Lattice` (src/PyBridge.cxx:41-113)."""

    m_expected_position_dimensions = -1  # static, Lattice.cu:44
    # splat_standalone also launches the same-level neighbour traversal (fused into the accumulate launch): every lattice
    # convolution starts with that list.  A caller that only splats and slices (no convolution on this lattice) sets the
    # attribute to False on its lattice and saves the traversal.
    prefetch_neighbours = True

    # ---------------------------------------------------------------- construction
    def __init__(self, config_file: Optional[str] = None, name: str = "", *, sigmas: Optional[Sequence[float]] = None,
                 capacity: Optional[int] = None, device=None):
        self.m_name = name
        self.m_lvl = 1
        self.m_positions = None
        self.m_sigmas_val_and_extent = []
        self._device = torch.device(device) if device is not None else None
        if config_file is not None:
            self._init_params(config_file)
        else:
            if sigmas is None or capacity is None:
                raise ValueError("Lattice needs either a config file or sigmas= and capacity=")
            self.m_hash_table = HashTable(int(capacity))
            self.m_sigmas = [float(s) for s in sigmas]
            if len(set(self.m_sigmas)) == 1:
                self.m_sigmas_val_and_extent = [(self.m_sigmas[0], len(self.m_sigmas))]
            else:
                self.m_sigmas_val_and_extent = [(s, 1) for s in self.m_sigmas]
            Lattice.m_expected_position_dimensions = len(self.m_sigmas)
        self._sigmas_tensor = None

    @staticmethod
    def create(config_file: str, name: str = "") -> "Lattice":  # PyBridge.cxx:47-48
        return Lattice(config_file, name)

    def _init_params(self, config_file: str):  # Lattice.cu:107-132
        path = config_file
        if not os.path.isabs(path) and not os.path.exists(path):
            root = os.environ.get("LATTICE_NET_CONFIG_ROOT", os.getcwd())
            path = os.path.join(root, config_file)
        cfg = _parse_lattice_cfg(path)
        self.m_hash_table = HashTable(int(cfg["hash_table_capacity"]))
        pairs = []
        for i in range(int(cfg["nr_sigmas"])):
            tok = cfg[f"sigma_{i}"].split()
            if len(tok) != 2:
                raise ValueError(f"sigma_{i} must be '<value> <extent>', got {cfg[f'sigma_{i}']!r}")
            pairs.append((float(tok[0]), int(float(tok[1]))))
        self.m_sigmas_val_and_extent = pairs
        self.set_sigmas(pairs)

    def set_sigmas(self, sigmas_list):  # Lattice.cu:135-161
        self.m_sigmas = []
        for sigma, nr_dim in sigmas_list:
            self.m_sigmas.extend([float(sigma)] * int(nr_dim))
        Lattice.m_expected_position_dimensions = len(self.m_sigmas)
        self._sigmas_tensor = None

    @classmethod
    def _clone_of(cls, other: "Lattice") -> "Lattice":
        """Lattice(Lattice* other), Lattice.cu:73-101: structure shared shallowly, nr_filled deep-copied."""
        new = cls.__new__(cls)
        new.m_name = ""
        new.m_lvl = other.m_lvl
        new.m_sigmas = list(other.m_sigmas)
        new.m_sigmas_val_and_extent = list(other.m_sigmas_val_and_extent)
        new._sigmas_tensor = None
        new._device = other._device
        new.m_positions = other.m_positions
        ht = HashTable(other.m_hash_table.capacity())
        oh = other.m_hash_table
        oh.flush()
        ht._storage = oh._storage
        ht.m_values_tensor = oh.m_values_tensor
        # The reference deep-copies the 1-element counter (Lattice.cu:93) and leaves the clone dirty
        # (HashTable.cu:15), which costs a copy kernel and a blocking readback per convolution.  Nothing
        # ever builds into a clone without first replacing its buffers (distribute / expand / coarse),
        # so the counter is shared and the cached host count is kept.
        ht._counters = oh._counters
        ht._pos_dim_hint, ht._val_dim_hint = oh.pos_dim(), oh.val_dim()
        ht._static_rows = oh._static_rows
        ht._batch = oh._batch
        ht._static_levels = getattr(oh, "_static_levels", None)
        ht.m_nr_filled_is_dirty = oh.m_nr_filled_is_dirty
        ht.m_nr_filled = oh.m_nr_filled
        new.m_hash_table = ht
        return new

    # ---------------------------------------------------------------- helpers
    def _dev(self, like: Optional[torch.Tensor] = None):
        if like is not None and like.is_cuda:
            self._device = like.device
        if self._device is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        return self._device

    def _stream(self):
        return _lib.stream_ptr(self._dev())

    def _sigmas_host(self):
        key = tuple(self.m_sigmas)
        if getattr(self, "_sig_key", None) != key:
            self._sig_key = key
            self._sig_arr = _lib.host_floats(self.m_sigmas)
        return self._sig_arr

    def _check_positions(self, positions_raw: torch.Tensor):  # Lattice.cu:162-170
        if positions_raw.dtype != torch.float32:
            raise ValueError("positions should be of type float")
        if positions_raw.dim() != 2:
            raise ValueError(f"positions should have dim 2 (N x pos_dim), got sizes {tuple(positions_raw.shape)}")
        pos_dim = positions_raw.shape[1]
        if len(self.m_sigmas) != pos_dim:
            raise ValueError(f"one sigma per position dimension is required: {len(self.m_sigmas)} sigmas, pos_dim {pos_dim}")
        if not positions_raw.is_contiguous():
            raise ValueError("positions_raw is not contiguous, call .contiguous() on it")
        if pos_dim > _lib.LN_MAX_POS_DIM:
            raise ValueError(f"pos_dim {pos_dim} unsupported (max {_lib.LN_MAX_POS_DIM})")
        _require_cuda(positions_raw, "positions")

    def _check_values(self, values: torch.Tensor):  # Lattice.cu:171-175
        if values.dtype not in (torch.float32, torch.float16) or values.dim() != 2 or not values.is_contiguous():
            raise ValueError("values should be a contiguous 2-D float tensor (float16 features are accumulated in float32)")
        _require_cuda(values, "values")

    def _check_positions_and_values(self, positions_raw, values):  # Lattice.cu:176-181
        if positions_raw.shape[0] != values.shape[0]:
            raise ValueError(f"positions {tuple(positions_raw.shape)} and values {tuple(values.shape)} must have the same number of rows")
        self._check_positions(positions_raw)
        self._check_values(values)

    def _workspace(self, nbytes: int) -> torch.Tensor:
        """Scratch for one C-ABI call.  Reused across calls of this object (all launches are ordered on
        torch's current stream, so the next call cannot start before the previous one has finished with it)."""
        nbytes = max(int(nbytes), 256)
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() < nbytes or ws.device != self._dev():
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=self._dev())
            self._ws = ws
        return ws

    def _ensure_table(self, pos_dim: int, val_dim: int, like: torch.Tensor):
        if not self.m_hash_table.is_initialized():
            self.m_hash_table.init(pos_dim, val_dim, self._dev(like))

    def _alloc_csr(self, tokens: int, groups_upper: int, planes=None):
        """One int32 allocation: seg_desc[G*S*4] (16-byte aligned) | grp_start[groups+1] | csr_tok[tokens] | seg_count[G+2]
        (G = LN_XCD_GROUPS segment regions of S = max_segments descriptors each)."""
        max_seg = _build_sizes(tokens, groups_upper)[2]
        tk = max(tokens, 1)
        G = _lib.LN_XCD_GROUPS
        buf = torch.empty((4 * G * max_seg + groups_upper + 1 + tk + G + 2,), dtype=torch.int32, device=self._dev())
        base = buf.data_ptr()  # torch allocations are at least 256-byte aligned
        o1 = 4 * G * max_seg
        o2 = o1 + groups_upper + 1
        o3 = o2 + tk
        c = _lib.LnCsr(base + 4 * o1, base + 4 * o2, base, base + 4 * o3, max_seg, _lib.ptr(planes))
        return buf, c, max_seg

    def _build(self, positions_raw, write: bool, vals=None, distributed=None):
        n, d = positions_raw.shape
        dev = self._dev(positions_raw)
        ht = self.m_hash_table
        idx = w = None
        if write:
            idx = torch.empty((n * (d + 1),), dtype=torch.int32, device=dev)
            w = torch.empty((n * (d + 1),), dtype=torch.float32, device=dev)
        tokens = n * (d + 1)
        st = ht._storage
        clear_vals, do_clear = ht.take_pending_clear()  # begin_splat's clear rides in the same C call
        self._choose_hash_capacity(tokens, fresh=do_clear)
        if do_clear:
            st.adopt_planes()  # space-ordered slots: the slot map is bound to the table's contents from here on
        cap = st.hashed()
        csr_buf, csr, max_seg = self._alloc_csr(tokens, cap, st.planes)
        # HashTable::clear fills the whole tensor (HashTable.cu:49-57).  Here only the rows that can be non-zero are cleared: a build
        # writes rows < its hashed range, so when THIS tensor is known to be zero from some row on (`_zero_beyond`: it was allocated
        # zeroed by this class, or an earlier build of this table cleared it) the clear stops at max(hashed range, that row) — which
        # also covers a hashed range that shrank since.  A tensor of unknown content (installed through set_values) is cleared whole.
        clear_elems = 0
        if clear_vals is not None:
            rows = int(clear_vals.shape[0])
            known = getattr(ht, "_zero_beyond", None)  # (a weak reference: a replaced accumulator is not kept alive by this note)
            zb = known[1] if (known is not None and known[0]() is clear_vals) else None
            clear_rows = rows if zb is None else min(rows, max(cap, zb))
            clear_elems = clear_rows * int(clear_vals.shape[1])
            ht._zero_beyond = (weakref.ref(clear_vals), min(rows, cap))  # after this build: every row the build cannot reach is zero

        def issue(force_atomic: bool):
            lib = _lib.load()
            if force_atomic and st.slot_map is not None:
                # replay of a bucketed build that overflowed under a slot map: the map does not fit this cloud (leaves fuller than their
                # slot runs).  Spilling past full buckets would keep the table correct but push keys hundreds of probes away from where
                # retrieval starts (its 300-probe cap, HashTableGPU.cuh:494, would lose them): the contents go back to hashed slots
                st.slot_map, st.slot_map_sb_max = None, 0
            ws = self._workspace(_build_sizes(tokens, cap)[0])
            t = ht.c_table()
            flags = (_lib.LN_BUILD_WRITE_IDX if write else 0) | (_lib.LN_BUILD_CLEAR_FIRST if do_clear else 0)
            if force_atomic or _FORCE_ATOMIC_BUILD:
                flags |= _lib.LN_BUILD_ATOMIC_PATH
            if _ROW_ORDER[0] == "canonical":
                flags |= _lib.LN_BUILD_CANONICAL_ROWS
            if _DETERMINISTIC[0]:
                flags |= _lib.LN_BUILD_SORTED_CSR
            cv, cn = _lib.ptr(clear_vals), clear_elems
            lib.ln_build_concurrency(_SCANS_IN_FLIGHT[0])  # (thread-local in the library: an assignment)
            ht.arm_count_readback()  # t is ht's cached struct: the sequence number travels in it
            if distributed is None:
                rc = lib.ln_build_splat(C.byref(t), _lib.ptr(positions_raw), self._sigmas_host(), n, _lib.ptr(idx), _lib.ptr(w), flags,
                                        C.byref(csr), _lib.ptr(ws), ws.numel(), cv, cn, self._stream())
                _lib.check(rc, "ln_build_splat")
            else:
                rc = lib.ln_distribute(C.byref(t), _lib.ptr(positions_raw), self._sigmas_host(), _lib.ptr(vals), n, vals.shape[1],
                                       _lib.ptr(idx), _lib.ptr(w), _lib.ptr(distributed), flags, C.byref(csr), _lib.ptr(ws), ws.numel(),
                                       cv, cn, self._stream())
                _lib.check(rc, "ln_distribute")
            st.touch()
            st.rows_follow_space = bool(st.slot_map is not None and do_clear and not (flags & (_lib.LN_BUILD_ATOMIC_PATH | _lib.LN_BUILD_CANONICAL_ROWS)))
            ht.m_nr_filled_is_dirty = True
            if n > 0 and ht._static_rows is None:
                ht.start_count_readback()
            else:
                ht._readback_pending = False  # nothing was launched that writes the pinned pair: read the device counters
            if write:
                # the build's slot -> tokens adjacency serves every scatter that uses these indices (groups = slots,
                # row of a group = entries[slot]); `idx` is kept alive by the entry so its address cannot be recycled
                st.csr_cache[(idx.data_ptr(), idx._version, idx.numel())] = (csr_buf, csr, max_seg, st.entries, idx)

        issue(False)
        if Lattice._static_build_log is not None and ht._static_rows is not None:
            Lattice._static_build_log.append((self.m_lvl, int(ht._static_rows), ht._pinned_np))
        # A bucketed build that reports LN_STATUS_BUCKET_OVERFLOW is replayed on the atomic path, together with the
        # work that was queued behind it (nr_lattice_vertices() is where the status is read).
        # (static-rows mode: the launches may be inside a stream capture, where nothing can be replayed; an overflowing
        # build is reported by static_build_report() instead)
        st.replay = [lambda: issue(True)] if (do_clear and ht._static_rows is None) else None
        return idx, w

    def _choose_hash_capacity(self, tokens: int, fresh: bool):
        """Before a build: a build that starts from a cleared table may (re)choose how many slots it hashes into; an incremental
        build into a table that hashes into fewer slots than it owns first re-hashes the existing vertices into all of them
        (ln_rehash), so that the cfg's capacity — not the first cloud's size — bounds what can still be inserted.

        Static-rows mode knows more than the token count: at most `bound` vertices exist, so _STATIC_SLOTS_PER_ROW x bound slots
        keep the load factor of the headline configuration (46.6 k vertices in 100 k slots) whatever the tokens per vertex are
        (a ScanNet scene: 800 k tokens on 129 k vertices -> 314 k slots = 512 bucket workgroups instead of 2048).  That range
        may be SMALLER than what earlier eager builds hashed into: the slots given up are emptied here, once."""
        ht = self.m_hash_table
        st = ht._storage
        if fresh:
            want = st.capacity if _HASH_POLICY[0] == "full" else min(st.capacity, max(16384, 2 * int(tokens)))
            bound = ht._static_rows
            if bound is not None and _HASH_POLICY[0] != "full":
                want = min(want, max(16384, int(_STATIC_SLOTS_PER_ROW * bound) + 1))
                # (a capture cannot re-shape the table: without a warm-up build in static-rows mode the wider range stays)
                if st.hash_capacity and want < st.hash_capacity and not torch.cuda.is_current_stream_capturing():
                    st.entries[want:st.hash_capacity] = -1
                    st.hash_capacity = want
                    st.touch()
            st.hash_capacity = max(st.hash_capacity, want)
        elif st.hash_capacity and st.hash_capacity < st.capacity:
            ht.flush()
            ht._zero_beyond = None  # the hashed range widens under an uncleared table: the next clear covers the whole accumulator
            st.hash_capacity = st.capacity
            t = ht.c_table()
            _lib.check(_lib.load().ln_rehash(C.byref(t), self._stream()), "ln_rehash")
            st.touch()

    def _after_build(self, fn):
        """Runs fn() now and again if the build it depends on has to be replayed."""
        fn()
        st = self.m_hash_table._storage
        if getattr(st, "replay", None) is not None:
            st.replay.append(fn)

    def canonicalize_rows(self, idx: torch.Tensor):
        """Relabels the rows of the table the last (bucketed, slot-order) build produced, `idx` and the build's cached CSR into the
        reference's serial numbering (ln_canonicalize), and drops every cache that holds the old row ids (neighbour lists).
        PRECONDITION: call it straight after the build — value rows already accumulated (splat, set_values) and tensors derived from
        the old numbering (neighbour lists handed out, splat indices other than `idx`) are NOT permuted."""
        ht = self.m_hash_table
        st = ht._storage
        lib = _lib.load()
        tokens = int(idx.numel())
        hit = st.csr_cache.get((idx.data_ptr(), idx._version, idx.numel()))
        ws = self._workspace(_build_sizes(tokens, st.hashed())[0])
        t = ht.c_table()
        _lib.check(lib.ln_canonicalize(C.byref(t), _lib.ptr(idx), tokens, C.byref(hit[1]) if hit is not None else None, _lib.ptr(ws), ws.numel(),
                                       self._stream()), "ln_canonicalize")
        keep = {k: v for k, v in st.csr_cache.items() if hit is not None and v is hit}
        st.touch()  # neighbour lists and any other CSR of this table name the old rows
        st.csr_cache.update(keep)

    # ---------------------------------------------------------------- atomics-free scatter (CSR)
    def _csr(self, idx: torch.Tensor):
        """CSR adjacency (group -> contributing tokens, cut into segments) for a splat-index tensor, cached with
        the table structure: the one emitted by the build that produced `idx`, else built from `idx` itself."""
        st = self.m_hash_table._storage
        key = (idx.data_ptr(), idx._version, idx.numel())
        hit = st.csr_cache.get(key)
        if hit is not None:
            return hit
        lib = _lib.load()
        tokens = idx.numel()
        rows_upper = self.m_hash_table.capacity()
        csr_buf, csr, max_seg = self._alloc_csr(tokens, rows_upper)
        ws = self._workspace(_build_sizes(tokens, rows_upper)[1])
        _lib.check(lib.ln_csr_build(_lib.ptr(idx), tokens, rows_upper, C.byref(csr), _lib.ptr(ws), ws.numel(), self._stream()), "ln_csr_build")
        if _DETERMINISTIC[0]:
            _lib.check(lib.ln_csr_sort(C.byref(csr), rows_upper, _lib.ptr(ws), ws.numel(), tokens, self._stream()), "ln_csr_sort")
        if len(st.csr_cache) >= 4:
            st.csr_cache.pop(next(iter(st.csr_cache)))
        entry = (csr_buf, csr, max_seg, None, idx)  # groups are rows: no indirection
        st.csr_cache[key] = entry
        return entry

    def _dense_hint(self, tokens: int) -> int:
        """LnCsr.dense: 1 when the cloud puts about 16 or more tokens on a vertex (judged by the row bound in static-rows mode, else
        by the vertex count of the last build this table reported — a fresh table answers 0; the hint only selects a kernel variant)."""
        ht = self.m_hash_table
        m = ht._static_rows if ht._static_rows is not None else ht.m_nr_filled
        return (1 if (m is not None and m > 0 and tokens >= _DENSE_TOKENS_PER_VERTEX * m) else 0) | (2 if _DETERMINISTIC[0] else 0)

    def _scatter_rows(self, src: torch.Tensor, idx: torch.Tensor, w: torch.Tensor, dst: torch.Tensor, val_dim: int, src_div: int,
                      src_stride: int):
        """dst[row] += sum of the row's contributions (segment-balanced reduce over the CSR adjacency); dst pre-zeroed."""
        _, csr, max_seg, grp_row, _ = self._csr(idx)
        csr.dense = self._dense_hint(idx.numel())
        lib = _lib.load()
        fn, what = (lib.ln_csr_reduce_rows_f16, "ln_csr_reduce_rows_f16") if src.dtype == torch.float16 else \
            (lib.ln_csr_reduce_rows, "ln_csr_reduce_rows")
        _lib.check(fn(C.byref(csr), _lib.ptr(grp_row), max_seg, _lib.ptr(src), _lib.ptr(w), val_dim, src_div, src_stride, _lib.ptr(dst),
                      self._stream()), what)

    def scatter_max(self, src: torch.Tensor, idx: torch.Tensor):
        """Per-vertex maximum of per-token rows src[T, C] with its argmax token (torch_scatter.scatter_max over the
        splat indices, lattice_modules.py:688).  Vertices without tokens get 0 / -1; ties go to the smallest token."""
        if src.dim() != 2 or src.dtype != torch.float32 or not src.is_contiguous() or src.shape[0] != idx.numel():
            raise ValueError("src must be a contiguous float [nr_tokens, C] tensor matching the splat indices")
        _require_cuda(src, "src")
        m = self.nr_lattice_vertices()
        c = int(src.shape[1])
        _, csr, max_seg, grp_row, _ = self._csr(idx)
        dev = self._dev()
        out_max = torch.empty((m, c), dtype=torch.float32, device=dev)
        out_arg = torch.empty((m, c), dtype=torch.int32, device=dev)
        packed = torch.empty((m * c,), dtype=torch.int64, device=dev)
        lib = _lib.load()
        _lib.check(lib.ln_csr_segment_max(C.byref(csr), _lib.ptr(grp_row), max_seg, _lib.ptr(src), c, m, _lib.ptr(packed), _lib.ptr(out_max),
                                          _lib.ptr(out_arg), self._stream()), "ln_csr_segment_max")
        return out_max, out_arg

    def vertex_point_counts(self, idx: torch.Tensor) -> torch.Tensor:
        """Number of tokens on every vertex (torch_scatter.scatter_add of ones, lattice_modules.py:692): int32 [M]."""
        m = self.nr_lattice_vertices()
        _, csr, _, grp_row, _ = self._csr(idx)
        counts = torch.empty((m,), dtype=torch.int32, device=self._dev())
        lib = _lib.load()
        # groups are hash slots (the CSR a build emitted: as many as that build hashed into) or rows (ln_csr_build)
        groups_upper = self.m_hash_table._storage.hashed() if grp_row is not None else self.m_hash_table.capacity()
        _lib.check(lib.ln_csr_group_sizes(C.byref(csr), _lib.ptr(grp_row), groups_upper, m, _lib.ptr(counts), self._stream()),
                   "ln_csr_group_sizes")
        return counts

    # ---------------------------------------------------------------- splat family
    def begin_splat(self, reset_hashmap: bool = True):  # Lattice.cu:185-193
        if reset_hashmap:
            self.m_hash_table.clear(lazy=True)
        else:
            self.m_hash_table.clear_only_values()
        self.m_hash_table.m_nr_filled_is_dirty = True

    def splat_standalone(self, positions_raw: torch.Tensor, values: torch.Tensor):  # Lattice.cu:196-241
        self._check_positions_and_values(positions_raw, values)
        n, d = positions_raw.shape
        v = values.shape[1]
        self.m_positions = positions_raw
        ht = self.m_hash_table
        if not ht.is_initialized():
            ht.init(d, v, self._dev(positions_raw))
        tv = ht.m_values_tensor
        # the accumulator is capacity rows tall (HashTable.cu:32) — except in static-rows mode, where no row beyond the bound can
        # be a vertex: the bound is enough (a ScanNet-sized table, 5 M slots x 32 channels, is 640 MB to zero per build otherwise)
        cap = ht._static_rows if ht._static_rows is not None else ht.capacity()
        if tv is None or tuple(tv.shape) != (cap, v) or tv.requires_grad or tv._base is not None or not tv.is_contiguous():
            # The table must own a plain [capacity, V] accumulator (HashTable.cu:32).  Python code re-points the
            # values of this object between ops (set_values in every Function); the reference would then splat into
            # whatever tensor was left there (out of bounds if it is shorter, Lattice.cu:230).  Install a fresh one.
            pending = getattr(ht, "_clear_pending", False)
            ht._clear_pending = False
            if pending:
                # a deferred begin_splat clear zeroes the rows a build can reach (the hashed range) inside the build call: a second
                # fill of the same rows here would only cost (6 MB per step at the headline configuration); rows beyond are zeroed once
                self._choose_hash_capacity(n * (d + 1), fresh=True)
                reach = min(cap, ht._storage.hashed())
                ht.m_values_tensor = torch.empty((cap, v), dtype=torch.float32, device=self._dev(positions_raw))
                if reach < cap:
                    ht.m_values_tensor[reach:].zero_()
                ht._zero_beyond = (weakref.ref(ht.m_values_tensor), reach)
            else:
                ht.m_values_tensor = torch.zeros((cap, v), dtype=torch.float32, device=self._dev(positions_raw))
                ht._zero_beyond = (weakref.ref(ht.m_values_tensor), 0)
            ht._clear_pending = pending
        idx, w = self._build(positions_raw, True)
        tv = ht.m_values_tensor
        # splatCacheNaive (LatticeGPU.cuh:926-973) as a token-balanced reduce; begin_splat zeroed the table values
        self._after_build(lambda: self._accumulate_and_prefetch(values, idx, w, tv, v, d + 1, n * (d + 1)))
        self._trace_level()
        return idx, w

    def just_create_verts(self, positions_raw: torch.Tensor, return_indices_and_weights: bool):  # Lattice.cu:244-290
        self._check_positions(positions_raw)
        d = positions_raw.shape[1]
        self._ensure_table(d, 1, positions_raw)
        idx, w = self._build(positions_raw, bool(return_indices_and_weights))
        self._trace_level()
        return idx, w

    def distribute(self, positions_raw: torch.Tensor, values: torch.Tensor, reset_hashmap: bool = True):  # Lattice.cu:351-410
        self._check_positions_and_values(positions_raw, values)
        n, d = positions_raw.shape
        v = values.shape[1]
        dev = self._dev(positions_raw)
        self.m_positions = positions_raw
        oh = self.m_hash_table
        if not oh.is_initialized():
            if reset_hashmap:
                # the reference allocates CAP-sized buffers here only to clone and clear them below
                # (Lattice.cu:362-391); this lattice just needs to know its dimensions
                oh._pos_dim_hint, oh._val_dim_hint = d, v
            else:
                oh.init(d, v, dev)
        distributed = torch.empty((n * (d + 1), d + v + 1), dtype=torch.float32, device=dev)
        new = Lattice._clone_of(self)
        new.m_name = "distributed_lattice"
        nh = new.m_hash_table
        if reset_hashmap:
            # deep copy + clear of three CAP-sized tensors (Lattice.cu:376-391) == fresh cleared buffers
            nh._storage = _TableStorage(oh.capacity(), d, dev)
            nh.m_values_tensor = torch.zeros((oh.capacity(), oh.val_dim() or v), dtype=torch.float32, device=dev)
            nh._zero_beyond = (weakref.ref(nh.m_values_tensor), 0)
            nh._counters = nh._storage.fresh_counters
            nh.clear(lazy=True)  # issued inside the build call
        else:
            nh._storage = oh._storage.clone()
            nh.m_values_tensor = torch.zeros_like(oh.m_values_tensor)
            nh._counters = oh._counters.clone()
        idx, w = new._build(positions_raw, True, vals=values, distributed=distributed)
        new._after_build(lambda: new._prefetch_neighbours(n * (d + 1)))
        new._trace_level()
        return new, distributed, idx, w

    def expand(self, positions_raw: torch.Tensor, point_multiplier: int, noise_stddev: float, expand_values: bool):  # Lattice.cu:292-348
        self._check_positions(positions_raw)
        d = positions_raw.shape[1]
        self._ensure_table(d, 1, positions_raw)
        pos = positions_raw.repeat(point_multiplier, 1)
        pos = pos + torch.randn_like(pos) * noise_stddev
        new = Lattice._clone_of(self)
        new.m_name = "expanded_lattice"
        oh, nh = self.m_hash_table, new.m_hash_table
        nh._storage = oh._storage.clone()
        nh.m_values_tensor = torch.zeros((1, self.val_dim()), dtype=torch.float32, device=self._dev())
        nh._counters = oh._counters.clone()
        new.just_create_verts(pos.contiguous(), False)
        if expand_values:
            diff = new.nr_lattice_vertices() - self.nr_lattice_vertices()
            if diff < 0:
                raise _lib.LatticeNetHipError("expand produced fewer vertices than the source lattice")
            new.set_values(torch.nn.functional.pad(self.values()[: self.nr_lattice_vertices()], (0, 0, 0, diff)))
        return new

    def _accumulate_and_prefetch(self, values, idx, w, dst, val_dim, src_div, tokens):
        """splatCacheNaive as a CSR segment reduce and the same-level neighbour prefetch in ONE launch (they only depend on
        the build; see _prefetch_neighbours for why the list is computed this early)."""
        ht = self.m_hash_table
        st = ht._storage
        rows_upper = min(ht.capacity(), tokens)
        _, csr, max_seg, grp_row, _ = self._csr(idx)
        if rows_upper <= 0 or max_seg <= 0 or not self.prefetch_neighbours:
            return self._scatter_rows(values, idx, w, dst, val_dim, src_div, val_dim)
        lib = _lib.load()
        nbr = torch.empty((rows_upper, self.get_filter_extent(1)), dtype=torch.int32, device=self._dev())
        t = ht.c_table()
        fn, what = (lib.ln_splat_accumulate_and_neighbours_f16, "ln_splat_accumulate_and_neighbours_f16") if values.dtype == torch.float16 \
            else (lib.ln_splat_accumulate_and_neighbours, "ln_splat_accumulate_and_neighbours")
        csr.dense = self._dense_hint(tokens)  # (issued behind the build, before its vertex count is known: the previous build's)
        _lib.check(fn(C.byref(csr), _lib.ptr(grp_row), max_seg, _lib.ptr(values), _lib.ptr(w), val_dim, src_div, val_dim, _lib.ptr(dst),
                      C.byref(t), rows_upper, _lib.ptr(nbr), self._stream()), what)
        st.nbr_cache[("prefetch", st.uid, st.version, self.m_lvl)] = (nbr,)

    # ---------------------------------------------------------------- neighbour list (shared)
    def _prefetch_neighbours(self, tokens: int):
        """Launches the same-level, dilation-1 neighbour traversal right behind a build, sized by an upper bound on
        the vertex count (the kernel stops at the device-side nr_filled).  Every lattice convolution starts with
        this list; issuing it here keeps it off the critical path that follows the host readback of nr_filled."""
        ht = self.m_hash_table
        st = ht._storage
        rows_upper = min(ht.capacity(), tokens)
        if rows_upper <= 0:
            return
        lib = _lib.load()
        E = self.get_filter_extent(1)
        nbr = torch.empty((rows_upper, E), dtype=torch.int32, device=self._dev())
        t = ht.c_table()
        _lib.check(lib.ln_neighbours(C.byref(t), rows_upper, C.byref(t), self.m_lvl, self.m_lvl, 1, 0, _lib.ptr(nbr), self._stream()),
                   "ln_neighbours")
        st.nbr_cache[("prefetch", st.uid, st.version, self.m_lvl)] = (nbr,)


    def neighbours(self, lattice_neighbours: Optional["Lattice"], dilation: int, flip_neighbours: bool) -> torch.Tensor:
        """[M, E] int32 neighbour list of this (query) lattice in `lattice_neighbours`, cached.  The flipped list
        is the un-flipped one with the np/nm slots of every axis swapped (LatticeGPU.cuh:1622-1626,1645-1649)."""
        nb = lattice_neighbours if lattice_neighbours is not None else self
        if abs(self.m_lvl - nb.m_lvl) > 1:  # Lattice.cu:439
            raise ValueError(f"query lvl {self.m_lvl} and neighbours lvl {nb.m_lvl} must differ by at most 1")
        m = self.nr_lattice_vertices()
        if m == 0:
            raise _lib.LatticeNetHipError("this lattice has zero vertices")
        sq, sn = self.m_hash_table._storage, nb.m_hash_table._storage
        key = (sn.uid, sn.version, self.m_lvl, nb.m_lvl, int(dilation), bool(flip_neighbours), m)
        hit = sq.nbr_cache.get(key)
        if hit is not None:
            return hit[0]
        E = self.get_filter_extent(1)
        pre = sq.nbr_cache.get(("prefetch", sn.uid, sn.version, self.m_lvl)) if (sn is sq and not flip_neighbours and dilation == 1 and
                                                                                  nb.m_lvl == self.m_lvl) else None
        if pre is not None and pre[0].shape[0] >= m:
            nbr = pre[0][:m]
            sq.nbr_cache[key] = (nbr,)
            return nbr
        if flip_neighbours:
            base = self.neighbours(nb, dilation, False)
            perm = [e ^ 1 for e in range(E - 1)] + [E - 1]
            nbr = base[:, perm].contiguous()
        else:
            lib = _lib.load()
            nbr = torch.empty((m, E), dtype=torch.int32, device=self._dev())
            tq, tn = self.m_hash_table.c_table(), nb.m_hash_table.c_table()
            _lib.check(lib.ln_neighbours(C.byref(tq), m, C.byref(tn), self.m_lvl, nb.m_lvl, int(dilation), 0, _lib.ptr(nbr),
                                         self._stream()), "ln_neighbours")
        sq.nbr_cache[key] = (nbr,)  # keyed by the neighbour storage's uid: no reference to it (a storage in its own cache is a cycle)
        return nbr

    def _row_partition(self):
        """Device pointer of LnTable.row_regions when the rows of this lattice follow space (bucketed build over space-ordered slots),
        else None: the argument of ln_conv_row_partition for convolutions whose output rows are this lattice's."""
        st = self.m_hash_table._storage
        if st is None or not st.rows_follow_space or st.row_regions is None:
            return None
        return st.row_regions.data_ptr()

    def _check_filter_extent(self, filter_extent: int):
        if filter_extent != self.get_filter_extent(1):
            raise ValueError(f"filter extent should be {self.get_filter_extent(1)} (1-hop neighbourhood + centre), got {filter_extent}")

    # ---------------------------------------------------------------- convolution family
    def convolve_im2row_standalone(self, filter_bank: torch.Tensor, dilation: int, lattice_neighbours: Optional["Lattice"],
                                   flip_neighbours: bool, filter_is_transposed: bool = False) -> "Lattice":  # Lattice.cu:424-474
        """`filter_is_transposed=True` (extension): `filter_bank` is the [E*F, V] bank of the convolution being
        differentiated; its per-slot transpose — the reference's filter_bank_backwards (lattice_funcs.py:307-311) —
        is applied inside the kernel instead of being materialised."""
        nb = lattice_neighbours if lattice_neighbours is not None else self
        if filter_bank is None or filter_bank.dim() != 2:
            raise ValueError("filter bank should be 2-D: (filter_extent * val_dim) x nr_filters")
        filter_bank = filter_bank.contiguous()
        v = nb.val_dim()
        E = self.get_filter_extent(1)
        if filter_is_transposed:
            if filter_bank.shape[1] != v or filter_bank.shape[0] % E != 0:
                raise ValueError(f"transposed filter bank should be (filter_extent * nr_filters) x val_dim={v}, got {tuple(filter_bank.shape)}")
            nr_filters = int(filter_bank.shape[0]) // E
            filter_extent = E
        else:
            nr_filters = int(filter_bank.shape[1])
            filter_extent = filter_bank.shape[0] // v
            self._check_filter_extent(filter_extent)
            if filter_bank.shape[0] != filter_extent * v:
                raise ValueError("filter bank rows must be filter_extent * val_dim")
        nbr = self.neighbours(nb, dilation, False)  # the flipped traversal is applied in-kernel
        m = nbr.shape[0]
        vals = nb.values()
        half = vals.dtype == torch.float16  # fp16 feature path (C5): fp16 operands, fp32 accumulation
        if filter_bank.dtype != vals.dtype:
            raise ValueError(f"filter bank is {filter_bank.dtype}, lattice values are {vals.dtype}")
        out = torch.empty((m, nr_filters), dtype=vals.dtype, device=self._dev())
        flags = (_lib.LN_CONV_FLIP_NEIGHBOURS if flip_neighbours else 0) | (_lib.LN_CONV_TRANSPOSED_FILTER if filter_is_transposed else 0)
        lib = _lib.load()
        if half:
            _lib.check(lib.ln_conv_forward_f16(_lib.ptr(nbr), _lib.ptr(vals), _lib.ptr(filter_bank), m, filter_extent, v, nr_filters, flags,
                                               _lib.ptr(out), self._stream()), "ln_conv_forward_f16")
        else:
            # few vertices (coarse levels): the kernel splits the contraction over the filter slots and needs room for the partials
            wsb = int(lib.ln_conv_forward_workspace_bytes(m, filter_extent, v, nr_filters))
            # The split bank of an unchanged filter is kept with the filter tensor (same storage, same version, same sizes -> same
            # kernel and bank layout): repeated convolutions with FROZEN weights skip the split launch.  Only for filters that do not
            # require a gradient: the version counter is the only change detector there is, and the fused optimizers update
            # parameters without moving it (measured: torch.optim.AdamW(fused=True) leaves p._version untouched,
            # tools/probes/fused_adamw_version_probe.py — a cache keyed on it served stale banks to a training run).  Only where the
            # workspace is the bank alone (no slot-split partials, which concurrent streams would share), never inside a stream capture.
            ws = None
            key = None
            bank_b = int(lib.ln_conv_bank_workspace_bytes(m, filter_extent, v, nr_filters))
            if bank_b > 0 and wsb == bank_b + 256 and _BANK_CACHE[0] and not filter_bank.requires_grad and \
                    not torch.cuda.is_current_stream_capturing():
                key = (filter_bank.data_ptr(), filter_bank._version, m, filter_extent, v, nr_filters, flags, self._stream())
                hit = _BANK_ENTRIES.get(filter_bank)
                if hit is not None and hit[0] == key:
                    ws, flags = hit[1], flags | _lib.LN_CONV_BANK_READY
            if ws is None and wsb > 256:
                ws = torch.empty((wsb,), dtype=torch.uint8, device=self._dev())
            part = self._row_partition()
            if part is not None:
                lib.ln_conv_row_partition(part)
            try:
                _lib.check(lib.ln_conv_forward_ws(_lib.ptr(nbr), _lib.ptr(vals), _lib.ptr(filter_bank), m, filter_extent, v, nr_filters, flags,
                                                  _lib.ptr(out), _lib.ptr(ws), 0 if ws is None else ws.numel(), self._stream()),
                           "ln_conv_forward")
            finally:
                if part is not None:
                    lib.ln_conv_row_partition(None)
            if key is not None and not (flags & _lib.LN_CONV_BANK_READY):
                try:
                    _BANK_ENTRIES[filter_bank] = (key, ws)
                except TypeError:  # (not weak-referenceable)
                    pass
        conv = Lattice._clone_of(self)
        conv.m_name = "convolved_lattice"
        conv.m_hash_table.set_values(out)
        return conv

    def convolve_im2row_backward(self, grad_out: torch.Tensor, filter_bank: torch.Tensor, dilation: int,
                                 query: Optional["Lattice"], neighbours: Optional["Lattice"], filter_grad_fp32: bool = False):
        """Both gradients of `query.convolve_im2row_standalone(filter_bank, dilation, neighbours, False)` in one call:
        (grad wrt the neighbour values, grad wrt the filter bank), both as gather-GEMMs on the current stream.
        `self` is unused beyond device bookkeeping; `query` / `neighbours` default to self."""
        q = query if query is not None else self
        nb = neighbours if neighbours is not None else self
        lib = _lib.load()
        main = self._stream()
        E = q.get_filter_extent(1)
        grad_out = grad_out.contiguous()
        filter_bank = filter_bank.contiguous()
        v = nb.val_dim()               # forward input channels
        f = int(filter_bank.shape[1])  # forward output channels
        if filter_bank.shape[0] != E * v or grad_out.shape[1] != f:
            raise ValueError("filter bank / gradient shapes do not match the forward convolution")
        # ---- filter gradient on the side stream: needs nbr(query -> neighbours), neighbour values, grad_out
        nbr_q = q.neighbours(nb, dilation, False)
        mq = nbr_q.shape[0]
        if grad_out.shape[0] != mq:
            raise ValueError(f"grad_out has {grad_out.shape[0]} rows, the query lattice has {mq} vertices")
        dev = self._dev()
        half = grad_out.dtype == torch.float16
        if filter_bank.dtype != grad_out.dtype or nb.values().dtype != grad_out.dtype:
            raise ValueError("grad_out, filter bank and lattice values must share one dtype (float32 or float16)")
        gf = torch.empty((E * v, f), dtype=torch.float32, device=dev)
        ws_bytes = lib.ln_conv_grad_filter_f16_workspace_bytes(mq, E, v, f) if half else lib.ln_conv_grad_filter_workspace_bytes(mq, E, v, f)
        ws_bytes = int(ws_bytes)
        if not half:  # + the slot-split partials of the value-gradient convolution (behind the filter gradient's slabs)
            conv_ws = int(lib.ln_conv_forward_workspace_bytes(int(nb.nr_lattice_vertices()), E, f, v))
            if conv_ws > 256:
                ws_bytes = ((ws_bytes + 255) // 256) * 256 + conv_ws
        ws = torch.empty((max(ws_bytes, 256),), dtype=torch.uint8, device=dev)
        # ---- value gradient on the main stream: the query and neighbour roles swap (funcs:307-313, 380-387)
        nbr_n = nb.neighbours(q, dilation, False)
        mn = nbr_n.shape[0]
        gvals = torch.empty((mn, v), dtype=grad_out.dtype, device=dev)
        if half:
            _lib.check(lib.ln_conv_grad_filter_f16(_lib.ptr(nbr_q), _lib.ptr(nb.values()), _lib.ptr(grad_out), mq, E, v, f, _lib.ptr(gf),
                                                   _lib.ptr(ws), ws.numel(), main), "ln_conv_grad_filter_f16")
            flags = _lib.LN_CONV_FLIP_NEIGHBOURS | _lib.LN_CONV_TRANSPOSED_FILTER
            _lib.check(lib.ln_conv_forward_f16(_lib.ptr(nbr_n), _lib.ptr(grad_out), _lib.ptr(filter_bank), mn, E, f, v, flags, _lib.ptr(gvals),
                                               main), "ln_conv_forward_f16")
            return gvals, (gf if filter_grad_fp32 else gf.to(torch.float16))  # (fp32 master weights: the kernel's own fp32 sums)
        # (Measured on MI355X: running the filter gradient on a second stream made the step SLOWER, 0.254 -> 0.286 ms:
        # both kernels already fill the chip and the event hand-offs cost more than the overlap.  What does pay is
        # putting the slab sum of the filter gradient into the value-gradient launch: ln_conv_backward.)
        part = nb._row_partition() if q.m_hash_table._storage is nb.m_hash_table._storage else None
        if part is not None:
            lib.ln_conv_row_partition(part)
        try:
            _lib.check(lib.ln_conv_backward(_lib.ptr(nbr_q), _lib.ptr(nbr_n), _lib.ptr(nb.values()), _lib.ptr(grad_out), _lib.ptr(filter_bank), mq,
                                            mn, E, v, f, _lib.ptr(gvals), _lib.ptr(gf), _lib.ptr(ws), ws.numel(), main), "ln_conv_backward")
        finally:
            if part is not None:
                lib.ln_conv_row_partition(None)
        return gvals, gf

    def convolve_im2row_grad_filter(self, grad_out: torch.Tensor, dilation: int, lattice_neighbours: Optional["Lattice"],
                                    filter_extent: int) -> torch.Tensor:
        """grad_filter = im2row(...)^T @ grad_out (lattice_funcs.py:298-302) without the rowified tensor."""
        nb = lattice_neighbours if lattice_neighbours is not None else self
        self._check_filter_extent(filter_extent)
        grad_out = grad_out.contiguous()
        nbr = self.neighbours(nb, dilation, False)
        m = nbr.shape[0]
        v = nb.val_dim()
        f = int(grad_out.shape[1])
        if grad_out.shape[0] != m:
            raise ValueError(f"grad_out has {grad_out.shape[0]} rows, lattice has {m} vertices")
        lib = _lib.load()
        gf = torch.empty((filter_extent * v, f), dtype=torch.float32, device=self._dev())
        ws = self._workspace(lib.ln_conv_grad_filter_workspace_bytes(m, filter_extent, v, f))
        _lib.check(lib.ln_conv_grad_filter(_lib.ptr(nbr), _lib.ptr(nb.values()), _lib.ptr(grad_out), m, filter_extent, v, f,
                                           _lib.ptr(gf), _lib.ptr(ws), ws.numel(), self._stream()), "ln_conv_grad_filter")
        return gf

    def im2row(self, lattice_neighbours: Optional["Lattice"], filter_extent: int, dilation: int, flip_neighbours: bool):  # Lattice.cu:612-644
        nb = lattice_neighbours if lattice_neighbours is not None else self
        self._check_filter_extent(filter_extent)
        nbr = self.neighbours(nb, dilation, flip_neighbours)
        m = nbr.shape[0]
        v = nb.val_dim()
        out = torch.empty((m, filter_extent * v), dtype=torch.float32, device=self._dev())
        lib = _lib.load()
        _lib.check(lib.ln_im2row(_lib.ptr(nbr), _lib.ptr(nb.values()), m, filter_extent, v, _lib.ptr(out), self._stream()), "ln_im2row")
        return out

    def im2rowindices(self, lattice_neighbours: Optional["Lattice"], filter_extent: int, dilation: int, flip_neighbours: bool):  # Lattice.cu:578-610
        nb = lattice_neighbours if lattice_neighbours is not None else self
        self._check_filter_extent(filter_extent)
        nbr = self.neighbours(nb, dilation, flip_neighbours)
        m = nbr.shape[0]
        v = nb.val_dim()
        out = torch.empty((m, filter_extent * v), dtype=torch.int32, device=self._dev())
        lib = _lib.load()
        _lib.check(lib.ln_im2rowindices(_lib.ptr(nbr), m, filter_extent, v, _lib.ptr(out), self._stream()), "ln_im2rowindices")
        return out

    def row2im(self, lattice_rowified: torch.Tensor, dilation: int, filter_extent: int, nr_filters: int,
               lattice_neighbours: Optional["Lattice"]):  # Lattice.cu:646-667
        nb = lattice_neighbours if lattice_neighbours is not None else self
        if not lattice_rowified.is_contiguous():
            raise ValueError("lattice rowified is not contiguous, call .contiguous() on it")
        v = lattice_rowified.shape[1] // filter_extent
        if v != self.val_dim():  # Lattice.cu:648
            raise ValueError(f"each rowified row should be val_dim*filter_extent long: row {lattice_rowified.shape[1]}, val_dim {self.val_dim()}")
        nbr = self.neighbours(nb, dilation, False)
        m = nbr.shape[0]
        out = torch.empty((m, v), dtype=torch.float32, device=self._dev())
        lib = _lib.load()
        _lib.check(lib.ln_row2im(_lib.ptr(nbr), _lib.ptr(lattice_rowified), m, filter_extent, v, _lib.ptr(out), self._stream()), "ln_row2im")
        self.m_hash_table.m_values_tensor = out  # the reference writes the result into this lattice's values
        return out

    # ---------------------------------------------------------------- coarse levels
    def _new_coarse(self) -> "Lattice":
        capacity = self.m_hash_table.capacity()
        d = self.pos_dim()
        dev = self._dev()
        coarse = Lattice._clone_of(self)
        coarse.m_name = "coarse_lattice"
        coarse.m_lvl = self.m_lvl + 1
        coarse.m_sigmas = [s * 2.0 for s in self.m_sigmas]  # Lattice.cu:679-682 / 718-722
        coarse._sigmas_tensor = None
        ht = HashTable(capacity)
        bp, step = self.m_hash_table._batch
        # (a batch of clouds: the key step halves with every coarser level, so that fine key x 2^-1 lands in the same cloud's block)
        ht._batch = (bp, (step // (2 * (d + 1))) * (d + 1)) if bp else (0, 0)
        ht._storage = _TableStorage(capacity, d, dev, spare_row_width=self.val_dim())
        # [1, val_dim] zeros: a placeholder until the coarse values exist
        ht.m_values_tensor = ht._storage.fresh_row if ht._storage.fresh_row is not None else torch.zeros((1, self.val_dim()), dtype=torch.float32, device=dev)
        ht._counters = ht._storage.fresh_counters
        levels = getattr(self.m_hash_table, "_static_levels", None)
        if self.m_hash_table._static_rows is not None:  # static-rows mode: every level needs its own bound
            if not levels or coarse.m_lvl not in levels:
                raise _lib.LatticeNetHipError(f"static-rows mode: no row bound for lattice level {coarse.m_lvl} "
                                              "(set_static_rows(bound, coarse_bounds=[...]))")
            ht._static_rows = min(int(levels[coarse.m_lvl]), capacity)
        ht._static_levels = levels
        coarse.m_hash_table = ht
        ht.clear(lazy=True)  # rides in the build call of create_coarse_verts_naive; c_table() flushes it for ln_coarsen
        return coarse

    def create_coarse_verts(self) -> "Lattice":  # Lattice.cu:670-703
        coarse = self._new_coarse()
        coarse.m_hash_table._storage.key_format = _lib.LN_KEYS_RAW  # halved fine keys are not all lattice points
        lib = _lib.load()
        m = self.nr_lattice_vertices()
        tokens = m * (2 * (self.pos_dim() + 1) + 1)
        coarse._choose_hash_capacity(tokens, fresh=True)  # (the coarse table is new: its deferred clear is flushed by c_table() below)
        cap = coarse.m_hash_table._storage.hashed()
        ws = self._workspace(_build_sizes(tokens, cap)[0])
        csr_buf, csr, _ = self._alloc_csr(tokens, cap)
        tf, tc = self.m_hash_table.c_table(), coarse.m_hash_table.c_table()
        _lib.check(lib.ln_coarsen(C.byref(tf), m, C.byref(tc), C.byref(csr), _lib.ptr(ws), ws.numel(), self._stream()), "ln_coarsen")
        coarse.m_hash_table._storage.touch()
        coarse.m_hash_table.m_nr_filled_is_dirty = True
        nr = coarse.nr_lattice_vertices()
        coarse.m_hash_table.m_values_tensor = torch.zeros((nr, self.val_dim()), dtype=torch.float32, device=self._dev())
        coarse._trace_level()
        return coarse

    def create_coarse_verts_naive(self, positions_raw: torch.Tensor) -> "Lattice":  # Lattice.cu:706-740
        self._check_positions(positions_raw)
        coarse = self._new_coarse()
        coarse.just_create_verts(positions_raw, False)
        coarse._trace_level()
        return coarse

    # ---------------------------------------------------------------- slice family
    def _check_slice_inputs(self, positions_raw, idx, w):
        self._check_positions(positions_raw)
        if self.val_dim() <= 0:
            raise ValueError("val_dim is 0: splat something first")
        if positions_raw.shape[1] != self.pos_dim():
            raise ValueError("position dimension does not match the lattice")
        n = positions_raw.shape[0]
        expect = n * (self.pos_dim() + 1)
        if idx is None or w is None or idx.numel() != expect or w.numel() != expect:  # Lattice.cu:771-773
            raise ValueError(f"indices / weights must have {expect} elements")
        return n

    def slice_standalone_with_precomputation(self, positions_raw, splatting_indices_tensor, splatting_weights_tensor,
                                             grad_accumulator: Optional[torch.Tensor] = None):  # Lattice.cu:744-786
        """`grad_accumulator` (extension): a float buffer that the same launch zero-fills — the tensor the backward pass of
        this slice will scatter into (SliceLattice hands it to slice_backwards_..._no_homogeneous)."""
        idx, w = splatting_indices_tensor.contiguous(), splatting_weights_tensor.contiguous()
        n = self._check_slice_inputs(positions_raw, idx, w)
        vals = self.values()
        lib = _lib.load()
        if vals.dtype == torch.float16:  # fp16 feature path: fp16 rows in, fp16 rows out, fp32 arithmetic
            out = torch.empty((n, self.val_dim()), dtype=torch.float16, device=self._dev())
            if grad_accumulator is None:
                _lib.check(lib.ln_slice_forward_f16(_lib.ptr(vals), _lib.ptr(idx), _lib.ptr(w), n, self.pos_dim(), self.val_dim(), _lib.ptr(out),
                                                    self._stream()), "ln_slice_forward_f16")
            else:  # (an fp32 buffer: the backward scatter accumulates in fp32 whatever the features are)
                _lib.check(lib.ln_slice_forward_f16_prepare_backward(_lib.ptr(vals), _lib.ptr(idx), _lib.ptr(w), n, self.pos_dim(), self.val_dim(),
                                                                     _lib.ptr(out), _lib.ptr(grad_accumulator), grad_accumulator.numel(),
                                                                     self._stream()), "ln_slice_forward_f16_prepare_backward")
            return out
        out = torch.empty((n, self.val_dim()), dtype=torch.float32, device=self._dev())
        # indices written by the last build of this table, whose rows follow space: take the points in the order of that build's CSR,
        # so that the rows a workgroup gathers sit in one kd region / one XCD's L2 (ln_slice_forward_ordered; bit-identical rows)
        st = self.m_hash_table._storage
        hit = st.csr_cache.get((idx.data_ptr(), idx._version, idx.numel())) if (st is not None and st.rows_follow_space and _ORDERED_SLICE[0]) else None
        if hit is not None and hit[3] is not None:
            t = self.m_hash_table.c_table()
            _lib.check(lib.ln_slice_forward_ordered(C.byref(t), C.byref(hit[1]), _lib.ptr(vals), _lib.ptr(idx), _lib.ptr(w), n, self.val_dim(),
                                                    _lib.ptr(out), _lib.ptr(grad_accumulator),
                                                    0 if grad_accumulator is None else grad_accumulator.numel(), self._stream()),
                       "ln_slice_forward_ordered")
            return out
        if grad_accumulator is None:
            _lib.check(lib.ln_slice_forward(_lib.ptr(vals), _lib.ptr(idx), _lib.ptr(w), n, self.pos_dim(), self.val_dim(), _lib.ptr(out),
                                            self._stream()), "ln_slice_forward")
        else:
            _lib.check(lib.ln_slice_forward_prepare_backward(_lib.ptr(vals), _lib.ptr(idx), _lib.ptr(w), n, self.pos_dim(), self.val_dim(),
                                                             _lib.ptr(out), _lib.ptr(grad_accumulator), grad_accumulator.numel(),
                                                             self._stream()), "ln_slice_forward_prepare_backward")
        return out

    def slice_standalone_no_precomputation(self, positions_raw):  # Lattice.cu:789-832
        self._check_positions(positions_raw)
        if positions_raw.shape[1] != self.pos_dim():
            raise ValueError("position dimension does not match the lattice")
        n, d = positions_raw.shape
        dev = self._dev()
        out = torch.empty((n, self.val_dim()), dtype=torch.float32, device=dev)
        idx = torch.empty((n * (d + 1),), dtype=torch.int32, device=dev)
        w = torch.empty((n * (d + 1),), dtype=torch.float32, device=dev)
        lib = _lib.load()
        t = self.m_hash_table.c_table()
        _lib.check(lib.ln_slice_no_precomputation(C.byref(t), _lib.ptr(self.values()), _lib.ptr(positions_raw), self._sigmas_host(), n,
                                                  self.val_dim(), _lib.ptr(out), _lib.ptr(idx), _lib.ptr(w), self._stream()),
                   "ln_slice_no_precomputation")
        return out, idx, w

    def gather_standalone_with_precomputation(self, positions_raw, splatting_indices_tensor, splatting_weights_tensor):  # Lattice.cu:878-917
        idx, w = splatting_indices_tensor.contiguous(), splatting_weights_tensor.contiguous()
        n = self._check_slice_inputs(positions_raw, idx, w)
        d, v = self.pos_dim(), self.val_dim()
        out = torch.empty((n, (d + 1) * (v + 1)), dtype=torch.float32, device=self._dev())
        lib = _lib.load()
        _lib.check(lib.ln_gather_forward(_lib.ptr(self.values()), _lib.ptr(idx), _lib.ptr(w), n, d, v, _lib.ptr(out), self._stream()),
                   "ln_gather_forward")
        return out

    def gather_standalone_no_precomputation(self, positions_raw):  # Lattice.cu:835-876 (kernel cannot compile in the reference)
        raise NotImplementedError("gather_no_precomputation is dead code in the reference (LatticeGPU.cuh:2875 names an undeclared symbol)")

    def slice_classify_no_precomputation(self, *args, **kwargs):  # Lattice.cu:920-980 (kernel cannot compile in the reference)
        raise NotImplementedError("slice_classify_no_precomputation is dead code in the reference (LatticeGPU.cuh:3356)")

    def slice_backwards_standalone_with_precomputation(self, *args, **kwargs):  # Lattice.cu:1045-1065, kernel body commented out
        raise NotImplementedError("the homogeneous slice backward kernel is commented out in the reference (LatticeGPU.cuh:3467-3536)")

    def slice_classify_with_precomputation(self, positions_raw, delta_weights, linear_clasify_weight, linear_clasify_bias, nr_classes,
                                           splatting_indices_tensor, splatting_weights_tensor):  # Lattice.cu:982-1039
        idx, w = splatting_indices_tensor.contiguous(), splatting_weights_tensor.contiguous()
        n = self._check_slice_inputs(positions_raw, idx, w)
        dw, lw, lb = delta_weights.contiguous(), linear_clasify_weight.contiguous(), linear_clasify_bias.contiguous()
        logits = torch.empty((n, nr_classes), dtype=torch.float32, device=self._dev())
        lib = _lib.load()
        _lib.check(lib.ln_slice_classify_forward(_lib.ptr(self.values()), _lib.ptr(dw), _lib.ptr(lw), _lib.ptr(lb), _lib.ptr(idx),
                                                 _lib.ptr(w), n, self.pos_dim(), self.val_dim(), int(nr_classes), _lib.ptr(logits),
                                                 self._stream()), "ln_slice_classify_forward")
        return logits

    def slice_backwards_standalone_with_precomputation_no_homogeneous(self, positions_raw, grad_sliced_values, splatting_indices_tensor,
                                                                      splatting_weights_tensor, zeroed_accumulator=None):  # Lattice.cu:1067-1088
        idx, w = splatting_indices_tensor.contiguous(), splatting_weights_tensor.contiguous()
        n = self._check_slice_inputs(positions_raw, idx, w)
        if grad_sliced_values.dim() != 2 or not grad_sliced_values.is_contiguous():
            raise ValueError("grad_sliced_values should be contiguous nr_positions x val_dim")
        v = int(grad_sliced_values.shape[1])
        m = self.nr_lattice_vertices()
        gv = zeroed_accumulator
        if gv is None or tuple(gv.shape) != (m, v) or gv.dtype != torch.float32 or not gv.is_contiguous():
            gv = torch.zeros((m, v), dtype=torch.float32, device=self._dev())
        self._scatter_rows(grad_sliced_values, idx, w, gv, v, self.pos_dim() + 1, v)
        if grad_sliced_values.dtype == torch.float16:
            gv = gv.half()  # fp16 feature path: accumulated in fp32, returned in the gradient's dtype
        self.m_hash_table.m_values_tensor = gv  # result is read back through values() (lattice_funcs.py:507)

    def slice_classify_backwards_with_precomputation(self, grad_class_logits, positions_raw, initial_values, delta_weights,
                                                     linear_clasify_weight, linear_clasify_bias, nr_classes, grad_lattice_values,
                                                     grad_delta_weights, grad_linear_clasify_weight, grad_linear_clasify_bias,
                                                     splatting_indices_tensor, splatting_weights_tensor):  # Lattice.cu:1091-1115
        idx, w = splatting_indices_tensor.contiguous(), splatting_weights_tensor.contiguous()
        n = self._check_slice_inputs(positions_raw, idx, w)
        if grad_class_logits.dim() != 2 or not grad_class_logits.is_contiguous():
            raise ValueError("grad_class_logits should be contiguous nr_positions x nr_classes")
        for name, g in (("grad_lattice_values", grad_lattice_values), ("grad_delta_weights", grad_delta_weights),
                        ("grad_linear_clasify_weight", grad_linear_clasify_weight), ("grad_linear_clasify_bias", grad_linear_clasify_bias)):
            if not g.is_contiguous():
                raise ValueError(f"{name} must be contiguous (it is filled in place)")
        iv, dw, lw = initial_values.contiguous(), delta_weights.contiguous(), linear_clasify_weight.contiguous()
        lib = _lib.load()
        v, c, d = int(iv.shape[1]), int(nr_classes), self.pos_dim()
        dev = self._dev()
        grad_sliced = torch.empty((n, v), dtype=torch.float32, device=dev)
        w_eff = torch.empty((n * (d + 1),), dtype=torch.float32, device=dev)
        ws = self._workspace(lib.ln_slice_classify_backward_workspace_bytes(n, d, v, c))
        # classifier / delta-weight gradients + dL/d(sliced features); the lattice-value gradient is then the same
        # CSR segment reduce as every other scatter onto the vertices (no N(d+1)V atomics, LG:3714-3719)
        _lib.check(lib.ln_slice_classify_backward(_lib.ptr(grad_class_logits), _lib.ptr(iv), _lib.ptr(dw), _lib.ptr(lw), _lib.ptr(idx),
                                                  _lib.ptr(w), n, d, v, c, None, _lib.ptr(grad_delta_weights),
                                                  _lib.ptr(grad_linear_clasify_weight), _lib.ptr(grad_linear_clasify_bias),
                                                  _lib.ptr(grad_sliced), _lib.ptr(w_eff), _lib.ptr(ws), ws.numel(), self._stream()),
                   "ln_slice_classify_backward")
        self._scatter_rows(grad_sliced, idx, w_eff, grad_lattice_values, v, d + 1, v)

    def gather_backwards_standalone_with_precomputation(self, positions_raw, grad_sliced_values, splatting_indices_tensor,
                                                        splatting_weights_tensor):  # Lattice.cu:1117-1142
        idx, w = splatting_indices_tensor.contiguous(), splatting_weights_tensor.contiguous()
        n = self._check_slice_inputs(positions_raw, idx, w)
        if grad_sliced_values.dim() != 2 or not grad_sliced_values.is_contiguous():
            raise ValueError("grad_sliced_values should be contiguous nr_positions x ((val_dim+1)*(pos_dim+1))")
        d = self.pos_dim()
        v = grad_sliced_values.shape[1] // (d + 1) - 1
        m = self.nr_lattice_vertices()
        gv = torch.zeros((m, v), dtype=torch.float32, device=self._dev())
        self._scatter_rows(grad_sliced_values, idx, w, gv, v, 1, v + 1)  # one (V+1)-wide source row per token
        self.m_hash_table.m_values_tensor = gv

    # ---------------------------------------------------------------- getters / setters
    def clone_lattice(self) -> "Lattice":  # Lattice.cu:1146-1149
        return Lattice._clone_of(self)

    def increase_sigmas(self, stepsize: float):  # Lattice.cu:1150-1155
        self.m_sigmas = [s + stepsize for s in self.m_sigmas]
        self._sigmas_tensor = None

    def set_sigma(self, sigma: float):  # Lattice.cu:1383-1390
        if len(self.m_sigmas_val_and_extent) != 1:
            raise ValueError("set_sigma assumes exactly one sigma group")
        self.m_sigmas = [float(sigma)] * len(self.m_sigmas)
        self._sigmas_tensor = None

    def val_dim(self) -> int:
        return self.m_hash_table.val_dim()

    def pos_dim(self) -> int:
        d = self.m_hash_table.pos_dim()
        # before the first build the table has no buffers yet; the sigmas already say how many dimensions it will have
        return d if d >= 0 else len(self.m_sigmas)

    def capacity(self) -> int:
        return self.m_hash_table.capacity()

    def name(self) -> str:
        return self.m_name

    def set_name(self, name: str):
        self.m_name = name

    def lvl(self) -> int:
        return self.m_lvl

    def rows_device(self):
        """Device int32[1] holding this lattice's vertex count when the static-rows mode is on (its [rows, *] tensors are then
        taller than the lattice: row-mixing consumers such as GroupNorm need the real count), else None."""
        ht = self.m_hash_table
        return ht._counters[0:1] if (ht._static_rows is not None and ht._counters is not None) else None

    # calibration of the static-rows mode for a whole network: every build made while a trace is open records (level, vertices)
    _level_trace = None
    # builds issued in static-rows mode while a log is open: (level, row bound, pinned report word) — what a captured step keeps
    # to check its replays (CapturedNetworkStep.check)
    _static_build_log = None

    @staticmethod
    def start_static_build_log():
        Lattice._static_build_log = []

    @staticmethod
    def stop_static_build_log():
        log, Lattice._static_build_log = Lattice._static_build_log or [], None
        return log

    @staticmethod
    def decode_report(word: int):
        """(vertex count, status bits) of a build's 64-bit report word (LnTable.host_counters)."""
        word = int(word)
        return word & 0xFFFFFFFF, (word >> 32) & 0xFF

    @staticmethod
    def start_level_trace():
        Lattice._level_trace = []

    @staticmethod
    def stop_level_trace():
        """{level: largest vertex count seen} of the builds since start_level_trace() (host reads: eager mode only)."""
        trace, Lattice._level_trace = Lattice._level_trace or [], None
        out = {}
        for lvl, nr in trace:
            out[lvl] = max(out.get(lvl, 0), nr)
        return out

    def _trace_level(self):
        if Lattice._level_trace is not None:
            Lattice._level_trace.append((self.m_lvl, self.nr_lattice_vertices()))

    def set_static_rows(self, rows_bound, coarse_bounds=None):
        """Capture-safe mode for hipGraph / torch.cuda.graph capture of a whole step (extension; None switches it off).

        The reference reads the vertex count back to the host after every build (Lattice.cu:1320-1352) and sizes the value
        tensors with it; a stream capture cannot wait on the device.  With a static row bound B (>= the largest vertex
        count of the clouds that will be replayed; <= capacity) nr_lattice_vertices() returns B without touching the
        device, every [M, *] tensor of the path gets B rows, and rows M..B-1 behave as isolated vertices: zero values,
        no neighbours (the traversal marks them LN_NOT_VISITED), no splat index refers to them — so every kernel of the
        path computes exactly what it computes in eager mode on rows < M and zeros beyond.  The real count and the
        status bits of each build still land in the pinned host pair: call static_build_report() after synchronising
        to check them (M > B or a bucket overflow mean the replayed step is invalid and must be redone eagerly).

        Capture notes: run the step once or twice on a side stream first (it creates the table buffers, the pinned counter
        pair and the build workspace, none of which may be allocated during a capture), keep this object alive as long as
        the graph, and do not use it eagerly with LARGER clouds while the graph exists (the workspace the captured kernels
        point at would be re-allocated)."""
        ht = self.m_hash_table
        if rows_bound is None:
            ht._static_rows = None
            ht._static_levels = None
            ht.m_nr_filled_is_dirty = True
            return
        rows_bound = int(rows_bound)
        if rows_bound < 1 or rows_bound > ht.capacity():
            raise ValueError(f"static row bound {rows_bound} must be in [1, capacity={ht.capacity()}]")
        ht._static_rows = rows_bound
        # coarse_bounds[k]: bound of the lattice k + 1 levels coarser than this one (create_coarse_verts hands them down)
        ht._static_levels = None if coarse_bounds is None else {self.m_lvl + 1 + k: int(b) for k, b in enumerate(coarse_bounds)}

    def set_cloud_batch(self, points_per_cloud: Optional[int], quotient_step: int = 1 << 13):
        """The multi-cloud launch form for small clouds.  From now on the positions handed to this lattice are a BATCH of independent
        clouds of `points_per_cloud` points each (cloud c = rows c * points_per_cloud ... of the positions tensor; None / 0 switches it
        off).  The lattice of cloud c is translated by c * quotient_step lattice cells along the first coordinate (a translation of the
        permutohedral lattice onto itself: same simplices, ranks and barycentric weights as a build of the cloud alone), so the clouds
        share ONE table without sharing a vertex and every operator of the path — build, neighbour lists, convolutions forward and
        backward, slice, gather, the scatters — runs over the whole batch in one launch with per-cloud results; filter gradients are the
        sum over the clouds.  The clouds must stay within quotient_step / 2 lattice cells of the origin on the first axis (a cell is
        ~0.8 (d + 1) sigma wide... see README 'Key range'; 8192 cells and 64 clouds fit the 2^20 cells of d <= 3).  Coarser levels
        created from this lattice inherit the batch with the step halved per level.  GroupNorm-style statistics over the lattice values
        would mix the clouds: this is for the operator path, not for the reference's batch-1 network semantics."""
        d = self.pos_dim() if self.m_hash_table.is_initialized() else len(self.m_sigmas)
        if not points_per_cloud:
            self.m_hash_table._batch = (0, 0)
        else:
            if int(points_per_cloud) < 1 or int(quotient_step) < 2 or int(quotient_step) % (1 << 4):
                raise ValueError("points_per_cloud >= 1 and a quotient_step that is a multiple of 16 (halved per coarser level)")
            self.m_hash_table._batch = (int(points_per_cloud), int(quotient_step) * (d + 1))
        if self.m_hash_table.is_initialized():
            self.m_hash_table._storage.touch()

    def set_region_planes(self, planes, leaf_shares=None):
        """kd split planes of key space (7 ints: 1 + 2 + 4 thresholds in heap order, see LnCsr.planes) or None.  With planes, the builds
        of this lattice file the CSR segments of every vertex under one of 8 compact regions and the scatter kernels let XCD r walk
        region r; under set_slot_order("space") (the default) the planes also order the SLOTS — and with them the rows — of the table
        by space, from the next build that starts with a clear on (LnTable.slot_map).  `leaf_shares` (8 floats, optional): the share
        of the vertices each region is expected to hold (balanced_region_planes(..., return_shares=True)): the slot run of a region is
        sized with it; without it the regions get equal runs, which suits planes that balance VERTICES."""
        st = self.m_hash_table._storage
        if st is None:
            raise _lib.LatticeNetHipError("build the lattice once before setting region planes")
        if planes is None:
            st.planes = st.plane_values = st.leaf_shares = None
            return
        p = torch.as_tensor(planes, dtype=torch.int32).reshape(-1)
        if p.numel() != 7:
            raise ValueError("region planes: 7 ints (1 + 2 + 4 thresholds)")
        if leaf_shares is not None and (len(leaf_shares) != 8 or min(leaf_shares) < 0 or sum(leaf_shares) <= 0):
            raise ValueError("leaf_shares: 8 non-negative numbers")
        st.planes = torch.cat([p, torch.zeros(1, dtype=torch.int32)]).to(self._dev())
        st.plane_values = tuple(int(x) for x in p.tolist())
        st.leaf_shares = None if leaf_shares is None else tuple(float(x) for x in leaf_shares)
        _PLANES_KEEPALIVE.append(st.planes)  # captured graphs hold the raw pointer (32 bytes per calibration)

    def balanced_region_planes(self, idx: torch.Tensor, vertex_weight: float = 0.0, return_shares: bool = False):
        """Planes that split the vertices of the CURRENT build into 8 regions of equal load (host-side calibration helper): weighted
        medians of key[0], then key[1 % d] inside each half, then key[2 % d] inside each quarter.  The load of a vertex is its token
        count + vertex_weight x the mean token count: 0 (default) balances tokens — what the bucket pass of the build, the segment
        walks and the slice spend their time on.  return_shares: also the share of the vertices in each region (for set_region_planes)."""
        import numpy as np
        m = self.nr_lattice_vertices()
        keys = self.m_hash_table._storage.keys[:m].cpu().numpy().astype("int64")
        wts = self.vertex_point_counts(idx).cpu().numpy().astype("float64")
        if vertex_weight > 0 and m > 0:
            wts = wts + vertex_weight * float(wts.mean())
        d = keys.shape[1]
        planes = [0] * 7
        leaf = np.zeros((m,), np.int64)

        def wmedian(vals, w):
            if len(vals) == 0:
                return 0
            order = np.argsort(vals, kind="stable")
            cs = np.cumsum(w[order])
            return int(vals[order][min(np.searchsorted(cs, cs[-1] / 2.0), len(vals) - 1)])  # region test is key >= plane

        def split(sel, node, lvl):
            if lvl == 3:
                leaf[sel] = node - 7
                return
            ax = lvl % d
            planes[node] = wmedian(keys[sel, ax], wts[sel])
            hi = keys[sel, ax] >= planes[node]
            split(sel[~hi], 2 * node + 1, lvl + 1)
            split(sel[hi], 2 * node + 2, lvl + 1)

        split(np.arange(m), 0, 0)
        if not return_shares:
            return planes
        shares = np.bincount(leaf, minlength=8).astype(np.float64) / max(m, 1)
        return planes, [float(x) for x in shares]

    def calibrate_regions(self, idx: torch.Tensor, vertex_weight: float = 0.0):
        """balanced_region_planes + set_region_planes on the current build (whose splat indices are `idx`): the next build that clears
        runs over token-balanced regions with slot runs sized by the regions' vertex shares."""
        planes, shares = self.balanced_region_planes(idx, vertex_weight=vertex_weight, return_shares=True)
        self.set_region_planes(planes, shares)
        return planes

    def static_build_report(self):
        """(vertex count, status bits) of the last build as its scan kernel wrote them to pinned host memory.  Call after
        the stream (or graph replay) that ran the build has been synchronised.  Raises if the build is unusable under
        the static row bound."""
        ht = self.m_hash_table
        arr = getattr(ht, "_pinned_np", None)
        if arr is None:
            raise _lib.LatticeNetHipError("no build has run on this lattice yet")
        nr, status = Lattice.decode_report(arr[0])
        bound = ht._static_rows
        if status & _lib.LN_STATUS_BUCKET_OVERFLOW:
            raise _lib.LatticeNetHipError("the bucketed build overflowed inside a static-rows step: redo this cloud in eager mode "
                                          "(set_static_rows(None)), which replays the build on the atomic path")
        if status & _lib.LN_STATUS_TABLE_FULL:
            raise _lib.LatticeNetHipError(f"hash table overflow: capacity {ht.capacity()} is too small for this cloud")
        if status & _lib.LN_STATUS_KEY_RANGE:
            raise _lib.LatticeNetHipError("a lattice key does not fit the packed 64-bit slot format (README: 'Key range')")
        if bound is not None and nr > bound:
            raise _lib.LatticeNetHipError(f"this cloud has {nr} lattice vertices but the static row bound is {bound}: rows beyond the "
                                          "bound were dropped from the convolution; raise the bound and re-capture")
        return nr, status

    def nr_lattice_vertices(self) -> int:  # Lattice.cu:1320-1352
        ht = self.m_hash_table
        if ht._static_rows is not None:
            return ht._static_rows
        if ht.m_nr_filled_is_dirty:
            both = ht.read_counters()  # [nr_filled, status]: the path's one wait on the device
            nr, status = int(both[0]), int(both[1])
            if status & _lib.LN_STATUS_BUCKET_OVERFLOW:
                # one LDS-staged bucket of the fast build filled up (table loaded beyond ~0.85): redo the build with
                # global atomics, which spill past a full bucket, and re-issue what was queued behind it
                st = ht._storage
                replay, st.replay = st.replay, None
                if not replay:
                    raise _lib.LatticeNetHipError("bucketed build overflowed and cannot be replayed")
                for fn in replay:
                    fn()
                st.replay = None
                both = ht.read_counters()
                nr, status = int(both[0]), int(both[1])
            if status & _lib.LN_STATUS_TABLE_FULL:
                raise _lib.LatticeNetHipError(f"hash table overflow: capacity {ht.capacity()} is too small for this cloud "
                                              "(the reference would spin forever, HashTableGPU.cuh:443)")
            if status & _lib.LN_STATUS_KEY_RANGE:
                d = self.pos_dim()
                fmt = getattr(ht._storage, "key_format", _lib.LN_KEYS_LATTICE)
                if fmt == _lib.LN_KEYS_LATTICE:
                    bits = min(32, 60 // max(d, 1))
                    reach = 2 ** (bits - 1) * (d + 1)
                else:
                    bits = min(32, 63 // max(d, 1))
                    reach = 2 ** (bits - 1)
                raise _lib.LatticeNetHipError(
                    f"a lattice key does not fit the packed 64-bit slot format: pos_dim {d} leaves {bits} bits per coordinate, i.e. "
                    f"lattice coordinates within +-{reach} (about +-{reach / (d + 1) / 0.8165:.0f} sigmas from the "
                    "origin per axis); centre the positions or use larger sigmas (README: 'Key range')")
            ht.m_nr_filled = nr
            ht.m_nr_filled_is_dirty = False
            if ht._storage is not None:
                ht._storage.replay = None  # build accepted: drop the replay closures (they keep the build's tensors alive)
        if ht.m_nr_filled < 0 or ht.m_nr_filled >= 1e8:
            raise _lib.LatticeNetHipError(f"implausible vertex count {ht.m_nr_filled}")
        return ht.m_nr_filled

    def get_filter_extent(self, neighborhood_size: int) -> int:  # Lattice.cu:1353-1358
        if neighborhood_size != 1:
            raise ValueError("only a neighbourhood size of 1 is implemented")
        return 2 * (self.pos_dim() + 1) + 1

    @staticmethod
    def get_expected_filter_extent(neighborhood_size: int) -> int:  # Lattice.cu:1359-1364
        if neighborhood_size != 1:
            raise ValueError("only a neighbourhood size of 1 is implemented")
        return 2 * (Lattice.m_expected_position_dimensions + 1) + 1

    def sigmas_tensor(self) -> torch.Tensor:
        if self._sigmas_tensor is None:
            self._sigmas_tensor = torch.tensor(self.m_sigmas, dtype=torch.float32)
        return self._sigmas_tensor

    def positions(self):
        return self.m_positions

    def hash_table(self) -> HashTable:
        return self.m_hash_table

    def values(self) -> torch.Tensor:
        self.m_hash_table.flush()
        return self.m_hash_table.m_values_tensor

    def set_values(self, new_values: torch.Tensor):  # Lattice.cu:1394-1399
        self.m_hash_table.set_values(new_values)
        if new_values.shape[0] != self.nr_lattice_vertices():
            raise ValueError(f"values have {new_values.shape[0]} rows but the lattice has {self.nr_lattice_vertices()} vertices")

    def set_positions(self, positions_raw: torch.Tensor):
        self.m_positions = positions_raw
