// Shared device/host helpers for the gfx950 lattice backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "latticenet_hip.h"

#define LN_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define LN_EMPTY_TOK 0xFFFFFFFFu
#define LN_MAX_RETRIEVE_CONFLICTS 300  // HashTableGPU.cuh:494

#if defined(__HIPCC__)
#define LN_HD __host__ __device__ __forceinline__
#else
#define LN_HD inline
#endif

// ---- error plumbing (host) -------------------------------------------------------------
void ln_set_error(const char* fmt, ...);
int ln_check_launch(const char* what);

#define LN_REQUIRE(cond, code, ...)  \
    do {                             \
        if (!(cond)) {               \
            ln_set_error(__VA_ARGS__); \
            return (code);           \
        }                            \
    } while (0)

// Zero-fill of `bytes` bytes (a multiple of 4, 4-byte aligned) as a KERNEL on `st`.  Used instead of hipMemsetAsync everywhere:
// inside a captured hipGraph a memset node in front of ln_canonicalize's marking pass did not take effect on replay (the bitmap
// kept the previous replay's bits — a memory fault as soon as the replayed cloud differed from the captured one; ROCm 7.2,
// MI355X), a kernel node does.
int ln_zero_async(void* p, size_t bytes, hipStream_t st);
// out[i] = sum over s of partial[s * total + i], slabs added in order (ln_conv.hip)
int ln_reduce_slabs_async(const float* partial, int nslabs, int total, float* out, hipStream_t st);

// ---- per-kernel live timing (ln_profile_begin / ln_profile_end) ------------------------------
// Every launch goes through LN_LAUNCH.  When profiling is armed for NAME the kernel is launched with hipExtLaunchKernelGGL and a
// (start, stop) event pair: the pair brackets the DISPATCH itself (the kernel's own begin / end timestamps, what rocprofv3
// reports as its duration), not the gaps in front of it as hipEventRecord calls around the launch would.
#include <hip/hip_ext.h>
struct LnProfEvents {
    hipEvent_t start, stop;
    bool armed;
};
LnProfEvents ln_prof_next(const char* name);
#define LN_LAUNCH(NAME, KERNEL, GRID, BLOCK, LDS, STREAM, ...)                                                              \
    do {                                                                                                                    \
        const LnProfEvents ln_prof_ev_ = ln_prof_next(NAME);                                                                \
        if (ln_prof_ev_.armed)                                                                                              \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, ln_prof_ev_.start, ln_prof_ev_.stop, 0, __VA_ARGS__);  \
        else                                                                                                                \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);                                              \
    } while (0)

// ---- key packing -----------------------------------------------------------------------
// A lattice key (first d coordinates, int32) is packed into one 64-bit word so that a slot can be claimed and its key
// published by a single 64-bit CAS.  Bit 63 stays clear in both formats, so a packed key never equals LN_EMPTY_KEY.
//   LN_KEYS_RAW     : min(32, 63/d) bits per coordinate — any integer tuple (the key-based coarsening inserts halved fine keys,
//                     which are not lattice points: LatticeGPU.cuh:2376-2400), d = 5: +-2048, d = 6: +-512 lattice units.
//   LN_KEYS_LATTICE : for tables that only ever hold points of the permutohedral lattice (everything built from positions):
//                     all d+1 coordinates of such a point share one remainder r mod d+1, key[i] = (d+1) q[i] + r, and the
//                     word holds r (3 bits) and the d quotients with min(32, 60/d) bits each — d = 5: +-12288, d = 6: +-3584
//                     lattice units, 6-7x the raw range (d <= 3: 2^20 quotients, i.e. more than int32 positions can need).
//                     A key that is not a lattice point cannot be packed — and cannot be in such a table: lookups return -1.
template <int D>
struct KeyPack {
    static constexpr int BITS = (63 / D) > 32 ? 32 : (63 / D);
    static constexpr int64_t LO = -(int64_t(1) << (BITS - 1));
    static constexpr int64_t HI = (int64_t(1) << (BITS - 1)) - 1;
    static constexpr uint64_t MASK = (BITS == 64) ? ~0ull : ((uint64_t(1) << BITS) - 1);
    static constexpr int QBITS = (60 / D) > 32 ? 32 : (60 / D);
    static constexpr int64_t QLO = -(int64_t(1) << (QBITS - 1));
    static constexpr int64_t QHI = (int64_t(1) << (QBITS - 1)) - 1;
    static constexpr uint64_t QMASK = (uint64_t(1) << QBITS) - 1;

    static LN_HD int rem(int k) {  // k mod (D+1) in [0, D]
        const int r = k % (D + 1);
        return r < 0 ? r + (D + 1) : r;
    }
    static LN_HD bool in_range(const int* key, int fmt) {
        bool ok = true;
        if (fmt == LN_KEYS_LATTICE) {
            const int r = rem(key[0]);
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const int64_t qi = (int64_t(key[i]) - r) / (D + 1);
                ok = ok && rem(key[i]) == r && qi >= QLO && qi <= QHI;
            }
            return ok;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) ok = ok && (int64_t(key[i]) >= LO) && (int64_t(key[i]) <= HI);
        return ok;
    }
    static LN_HD uint64_t pack(const int* key, int fmt) {
        uint64_t p = 0;
        if (fmt == LN_KEYS_LATTICE) {
            const int r = rem(key[0]);
            p = uint64_t(r);
#pragma unroll
            for (int i = 0; i < D; ++i) p |= (uint64_t((int64_t(key[i]) - r) / (D + 1)) & QMASK) << (3 + i * QBITS);
            return p;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) p |= (uint64_t(int64_t(key[i])) & MASK) << (i * BITS);
        return p;
    }
    // Lattice format when the caller KNOWS the shared remainder r (the simplex code does: vertex `r` of a simplex has remainder r):
    // no modulo per coordinate, and the division by d+1 is exact.
    static LN_HD bool lattice_in_range(const int* key, int r) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const int64_t qi = (int64_t(key[i]) - r) / (D + 1);
            ok = ok && qi >= QLO && qi <= QHI;
        }
        return ok;
    }
    static LN_HD uint64_t lattice_pack(const int* key, int r) {
        uint64_t p = uint64_t(r);
#pragma unroll
        for (int i = 0; i < D; ++i) p |= (uint64_t((int64_t(key[i]) - r) / (D + 1)) & QMASK) << (3 + i * QBITS);
        return p;
    }
    static LN_HD void unpack(uint64_t p, int* key, int fmt) {
        if (fmt == LN_KEYS_LATTICE) {
            const int r = int(p & 7);
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const uint64_t f = (p >> (3 + i * QBITS)) & QMASK;
                const int64_t qi = int64_t(f << (64 - QBITS)) >> (64 - QBITS);  // sign extend
                key[i] = int(qi * (D + 1) + r);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            uint64_t f = (p >> (i * BITS)) & MASK;
            int64_t s = int64_t(f << (64 - BITS)) >> (64 - BITS);  // sign extend
            key[i] = int(s);
        }
    }
};

// HashTableGPU::hash (HashTableGPU.cuh:35-43): k += key[i]; k *= 2531011 with uint32 wrap.
template <int D>
LN_HD uint32_t ln_hash(const int* key) {
    uint32_t k = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        k += uint32_t(key[i]);
        k *= 2531011u;
    }
    return k;
}

// ---- probe sequence ----------------------------------------------------------------------
// The table is cut into buckets of ln_bucket_slots(capacity) consecutive slots (the unit one
// workgroup stages in LDS during a build).  A key whose hash lands in bucket [lo, lo+size) probes
// that bucket first (linear, wrapping inside the bucket) and only when the bucket holds no empty
// slot continues linearly through the rest of the table:  probe i -> lo + (off+i) % size for
// i < size, (lo + i) % capacity afterwards.  Every slot is visited exactly once in `capacity`
// probes, as with the reference's plain linear probing (HashTableGPU.cuh:479-482).
#ifndef LN_BKT_SLOTS
#define LN_BKT_SLOTS 512
#endif
#define LN_BKT_MAX 2048
#define LN_XCD_GROUPS 8
// At least LN_BKT_MIN_COUNT buckets (one bucket = one workgroup of the build: fewer buckets than CUs leave CUs idle) as long
// as they keep >= LN_BKT_MIN_SLOTS slots each.
#ifndef LN_BKT_MIN_COUNT
#define LN_BKT_MIN_COUNT 256
#endif
#ifndef LN_BKT_MIN_SLOTS
#define LN_BKT_MIN_SLOTS 256
#endif
LN_HD int ln_bucket_slots(int capacity) {
    int nb = (capacity + LN_BKT_SLOTS - 1) / LN_BKT_SLOTS;
    if (nb < LN_BKT_MIN_COUNT) {
        nb = capacity / LN_BKT_MIN_SLOTS;
        if (nb > LN_BKT_MIN_COUNT) nb = LN_BKT_MIN_COUNT;
    } else {
        // whole rounds of one bucket workgroup per CU: a partly filled last round of the bucket pass costs a whole round (C5: 336
        // buckets of 5.7 k tokens 76 us, 256 buckets of 7.5 k tokens 21 us less; C4: 614 -> 512 buckets, 267 -> 256 us per step)
        nb = (nb / LN_BKT_MIN_COUNT) * LN_BKT_MIN_COUNT;
    }
    if (nb > LN_BKT_MAX) nb = LN_BKT_MAX;
    if (nb < 1) nb = 1;
    if (nb >= 2 * LN_XCD_GROUPS) nb = (nb / LN_XCD_GROUPS) * LN_XCD_GROUPS;  // whole buckets per XCD group
    return (capacity + nb - 1) / nb;
}
LN_HD int ln_bucket_count(int capacity) {
    const int sb = ln_bucket_slots(capacity);
    return (capacity + sb - 1) / sb;
}
// ---- space-ordered slots ----------------------------------------------------------------------------------------------
// A table may carry a SLOT MAP (LnTable.slot_map, LN_SLOT_MAP_INTS device ints written by the host's calibration):
//   [0..6]   planes of a 3-level kd partition of KEY space into 8 leaves, heap order (node i: children 2i+1 below the plane, 2i+2
//            at or above it; level l compares key[l % d]) — the same 8 regions the segment lists of LnCsr are filed under;
//   [7]      buckets per leaf (bucket count / 8);
//   [8..16]  first slot of each leaf's run of slots (+ the end of the last one: <= capacity, the few slots behind it stay empty);
//   [17..24] slots per bucket inside each leaf.
// The slot function then puts the LEAF of a key into the high part of its slot and hashes only inside the leaf's run: a key of
// leaf r starts probing at start[r] + hash % (start[r + 1] - start[r]), inside the bucket that slot falls in.  Every leaf owns the
// same NUMBER of buckets and the planes balance the TOKENS of the leaves, so the bucket workgroups of a build see equal token
// loads; the leaves hold very different numbers of vertices (a LiDAR scan: a few hot vertices next to the sensor, thousands of
// thin ones far out), so the SIZE of a leaf's buckets follows its vertex share and the load factor — what the probe lengths of
// inserts and of every retrieval depend on — is the same everywhere.  (Measured with equal bucket sizes, round 6: token-balanced
// planes left buckets 76 % full — hash build 35 -> 43 us, neighbour traversal 23 -> 45 us; vertex-balanced planes left buckets
// with 4.5x the mean token count — build 45 us.)
// Buckets are numbered in slot order and rows bucket by bucket, so ROWS follow space: the vertices of one leaf own one contiguous
// row range, and with them the value rows every gather of the path reads (9 neighbour rows per vertex in the convolutions, d+1
// rows per point in the slice).  A MI355X dispatches workgroup b to XCD b % 8 and each XCD has its own 4 MB L2: kernels that let
// XCD x work on the rows of leaf x (ln_partition_tile) fetch a value row into ~1.1 L2s instead of into 4.8 of the 8
// (tools/probes/slot_order_model.py; profiles/r6_pmc_traffic.json).  Only row ids are reference-visible (PyBridge exposes
// m_keys / m_nr_filled); retrieval uses the same function, so "every vertex inserted, retrieve finds it" holds as with plain
// hashing (HashTableGPU.cuh:425-519).  Without a map: h0 = hash % capacity, as the reference.
// The map is 25 wave-uniform words: kernels load it with scalar loads into registers and walk the tree with selects — no
// memory round trip stands between a key and its first probe.
LN_HD uint32_t ln_stir(uint32_t k) {
    k ^= k >> 15;
    k *= 2246822519u;
    k ^= k >> 13;
    return k;
}
struct LnSlotMap {  // passed BY VALUE / reference through inlined code only: it has to stay in (scalar) registers
    bool on;        // false: hashed table, nothing else is meaningful
    int planes[7];
    int bpl;
    int start[LN_XCD_GROUPS + 1];
    int sb[LN_XCD_GROUPS];
};
// p = LnTable.slot_map (nullptr: hashed table -> .on = false)
LN_HD LnSlotMap ln_load_slot_map(const int* p) {
    LnSlotMap m;
    m.on = p != nullptr;
    const int* q = m.on ? p : nullptr;
#pragma unroll
    for (int i = 0; i < 7; ++i) m.planes[i] = m.on ? q[i] : 0;
    m.bpl = m.on ? q[7] : 1;
#pragma unroll
    for (int i = 0; i <= LN_XCD_GROUPS; ++i) m.start[i] = m.on ? q[8 + i] : 0;
#pragma unroll
    for (int i = 0; i < LN_XCD_GROUPS; ++i) m.sb[i] = m.on ? q[17 + i] : 1;
    return m;
}
// a[leaf] for an array that lives in registers (a dynamic index would send it to scratch memory): three levels of selects
LN_HD int ln_sel8(const int* a, int leaf) {
    const int b0 = leaf & 1, b1 = leaf & 2, b2 = leaf & 4;
    const int x0 = b0 ? a[1] : a[0], x1 = b0 ? a[3] : a[2], x2 = b0 ? a[5] : a[4], x3 = b0 ? a[7] : a[6];
    const int y0 = b1 ? x1 : x0, y1 = b1 ? x3 : x2;
    return b2 ? y1 : y0;
}
// leaf (= segment region) of a key under the 3-level kd partition `planes` (7 ints in registers, LDS or memory)
template <int D, typename P>
LN_HD int ln_leaf_of_key(const int* key, P planes) {
    const int r0 = (key[0] >= planes[0]) ? 1 : 0;
    const int p1 = r0 ? planes[2] : planes[1];
    const int r1 = (key[1 % D] >= p1) ? 1 : 0;
    const int p2a = r1 ? planes[4] : planes[3], p2b = r1 ? planes[6] : planes[5];
    const int r2 = (key[2 % D] >= (r0 ? p2b : p2a)) ? 1 : 0;
    return 4 * r0 + 2 * r1 + r2;
}
struct LnProbe {
    int lo, size, off, cap;
    LN_HD LnProbe(uint32_t hash, int capacity, int sb) {
        const int h0 = int(hash % uint32_t(capacity));
        cap = capacity;
        lo = (h0 / sb) * sb;
        size = (capacity - lo < sb) ? (capacity - lo) : sb;
        off = h0 - lo;
    }
    // space-ordered form: the key's leaf owns the slots [glo, glo + gsz) in buckets of sbl slots
    LN_HD LnProbe(int glo, int gsz, int sbl, uint32_t slot_hash, int capacity) {
        const int o = int(slot_hash % uint32_t(gsz));
        cap = capacity;
        size = sbl;
        off = o % sbl;
        lo = glo + o - off;
    }
    // !m.on: plain hashing
    template <int D>
    static LN_HD LnProbe of_key(const int* key, int capacity, int sb, const LnSlotMap& m) {
        if (!m.on) return LnProbe(ln_hash<D>(key), capacity, sb);
        const int leaf = ln_leaf_of_key<D>(key, m.planes);
        const int glo = ln_sel8(m.start, leaf), ghi = ln_sel8(m.start + 1, leaf);
        // (the raw hash is a poor slot hash: 2531011 = 7 * 361573, and e.g. 511 = 7 * 73)
        return LnProbe(glo, ghi - glo, ln_sel8(m.sb, leaf), ln_stir(ln_hash<D>(key)), capacity);
    }
    template <int D>
    static LN_HD LnProbe of_key(const int* key, const LnTable& t, int sb) {
        return of_key<D>(key, t.capacity, sb, ln_load_slot_map(t.slot_map));
    }
    // bucket of the key's first probe (the partition pass of the build): leaf * buckets per leaf + bucket inside the leaf
    template <int D>
    static LN_HD int bucket_of_key(const int* key, int capacity, int sb, const LnSlotMap& m) {
        if (!m.on) return int(ln_hash<D>(key) % uint32_t(capacity)) / sb;
        const int leaf = ln_leaf_of_key<D>(key, m.planes);
        const int glo = ln_sel8(m.start, leaf), ghi = ln_sel8(m.start + 1, leaf);
        return leaf * m.bpl + int(ln_stir(ln_hash<D>(key)) % uint32_t(ghi - glo)) / ln_sel8(m.sb, leaf);
    }
    LN_HD int slot(int i) const {
        if (i < size) {
            int o = off + i;
            if (o >= size) o -= size;
            return lo + o;
        }
        int s = lo + i;
        if (s >= cap) s -= cap;
        return s;
    }
};
// Slot range of bucket b (the unit one workgroup of the bucket pass stages in LDS) and the starting offset of a key inside it.
struct LnBucket {
    int lo, size, glo, gsz;  // slots [lo, lo + size); the leaf's run [glo, glo + gsz) (gsz = 0: hashed table)
    // `map` = LnTable.slot_map (memory: b is uniform over the workgroup, so these are a handful of scalar loads; a register copy of the
    // map indexed by a run-time leaf would be sent to scratch memory)
    LN_HD LnBucket(int b, int capacity, int sb, const int* map) {
        if (map == nullptr) {
            lo = b * sb;
            size = (capacity - lo < sb) ? (capacity - lo) : sb;
            glo = 0;
            gsz = 0;
        } else {
            const int bpl = map[7];
            const int leaf = b / bpl;
            glo = map[8 + leaf];
            gsz = map[9 + leaf] - glo;
            size = map[17 + leaf];
            lo = glo + (b - leaf * bpl) * size;
        }
    }
    // for a caller that knows the key lives in this bucket: no tree walk
    template <int D>
    LN_HD int offset_of(const int* key, int capacity) const {
        if (gsz == 0) return int(ln_hash<D>(key) % uint32_t(capacity)) - lo;
        return glo + int(ln_stir(ln_hash<D>(key)) % uint32_t(gsz)) - lo;
    }
};
// segment region (XCD group) of bucket b of a space-ordered table
LN_HD int ln_region_of_bucket(int b, int nbk) { return int((long long)b * LN_XCD_GROUPS / nbk); }

// Region of a lattice key under the 3-level kd split of key space described at LnCsr.planes (include/latticenet_hip.h).
template <int D>
LN_HD int ln_region_of_key(const int* key, const int* planes) {
    return ln_leaf_of_key<D>(key, planes);
}

#if defined(__HIPCC__)
// Workgroup -> tile map that hands every XCD ONE contiguous run of the G tiles of a launch (block b runs on XCD b % 8: observed
// dispatch behaviour, used for speed only — any placement is correct).  Bijective for every G.
__device__ __forceinline__ int ln_xcd_chunk_tile(int b, int G) {
    const int q = G >> 3, r = G & 7, x = b & 7, j = b >> 3;
    return x * q + (x < r ? x : r) + j;
}
// Workgroup -> row tile map of the vertex-tiled kernels over a space-ordered table.  `part` = LnTable.row_regions as the build
// left it: part[r] = first row of region r (the kd regions own contiguous row ranges), part[8] = rows in total; nullptr = identity.
// XCD x (workgroups b % 8 == x) takes the tiles of region x, so that the neighbour rows it gathers sit in ITS L2; regions hold
// different numbers of vertices (the planes balance tokens), so the workgroups an XCD has left over take the tiles the fuller
// regions have beyond their XCD's share, in a fixed order.  The boundaries are clamped into a monotone sequence first: whatever
// `part` holds (a stale or half-written array after a build that was replayed on another path) the map is a bijection of [0, G).
// All of it is wave-uniform scalar work (9 scalar loads, ~60 scalar instructions).
__device__ __forceinline__ int ln_partition_tile(int b, int G, const int* __restrict__ part, int tile_rows) {
    if (part == nullptr || G < 2 * LN_XCD_GROUPS) return b;
    int s[LN_XCD_GROUPS + 1];
    s[0] = 0;
#pragma unroll
    for (int r = 1; r < LN_XCD_GROUPS; ++r) {
        int v = (part[r] + (tile_rows >> 1)) / tile_rows;
        v = v < s[r - 1] ? s[r - 1] : v;
        s[r] = v > G ? G : v;
    }
    s[LN_XCD_GROUPS] = G;
    const int q = G >> 3, rem = G & 7, x = b & 7, j = b >> 3;
    int spare = 0;  // rank of this workgroup among the left-over workgroups (XCD-major), valid when j >= tiles of region x
    int mine = 0, first = 0;
#pragma unroll
    for (int y = 0; y < LN_XCD_GROUPS; ++y) {
        const int wy = q + (y < rem ? 1 : 0), ty = s[y + 1] - s[y];
        if (y < x) spare += wy > ty ? wy - ty : 0;
        if (y == x) {
            mine = ty;
            first = s[y];
        }
    }
    if (j < mine) return first + j;
    spare += j - mine;
#pragma unroll
    for (int y = 0; y < LN_XCD_GROUPS; ++y) {
        const int wy = q + (y < rem ? 1 : 0), ty = s[y + 1] - s[y];
        const int over = ty > wy ? ty - wy : 0;
        if (spare < over) return s[y] + wy + spare;
        spare -= over;
    }
    return b;  // unreachable: the left-over workgroups and the left-over tiles are equally many
}
#endif

#if defined(__HIPCC__)
// HashTableGPU::retrieve (HashTableGPU.cuh:491-519) on packed slots: stop at an empty slot or
// after 300 mismatching probes.
// `m`: the table's slot map, loaded by the caller at kernel entry (.on = false: hashed table) — the scalar loads of the map then overlap
// with the loads in front of the lookup instead of standing between the key and its first probe
template <int D>
__device__ __forceinline__ int ln_retrieve(const LnTable& t, const int* key, const LnSlotMap& m) {
    if (!KeyPack<D>::in_range(key, t.key_format)) return -1;  // cannot have been inserted
    const uint64_t pk = KeyPack<D>::pack(key, t.key_format);
    const LnProbe pr = LnProbe::of_key<D>(key, t.capacity, ln_bucket_slots(t.capacity), m);
    const int limit = t.capacity < LN_MAX_RETRIEVE_CONFLICTS ? t.capacity : LN_MAX_RETRIEVE_CONFLICTS;
    // One probe per round trip.  (Round 6 tried four slots per trip, examined in probe order: the traversal got SLOWER, 10.2 -> 13.1 us
    // over hashed slots and 17.5 -> 18.2 us over stirred ones — finished lanes no longer drop out of the later loads, and the kernel
    // pays per divergent memory access as much as per dependent round trip.)
    for (int conflicts = 0; conflicts < limit; ++conflicts) {
        const int h = pr.slot(conflicts);
        const uint64_t cur = t.slot_keys[h];
        const int row = t.entries[h];  // issued together with the key: a hit costs one memory round trip, not two
        if (cur == LN_EMPTY_KEY) return -1;
        if (cur == pk) return row;
    }
    return -1;
}
template <int D>
__device__ __forceinline__ int ln_retrieve(const LnTable& t, const int* key) {
    return ln_retrieve<D>(t, key, ln_load_slot_map(t.slot_map));
}
#endif

#if defined(__HIPCC__)
// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global load / store / atomic the wave has
// in flight (s_waitcnt vmcnt(0)), which a kernel that keeps global round trips in flight ACROSS its barriers must not do.
__device__ __forceinline__ void ln_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Inclusive prefix sum over the 64 lanes of a wave on the DPP network (no LDS round trips: __shfl_up is a ds_bpermute per step):
// Hillis-Steele inside every row of 16 lanes (row_shr 1, 2, 4, 8; lanes without a source add 0), then the row totals are handed
// on (row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3).
__device__ __forceinline__ int ln_wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return v;
}

// Exclusive prefix sum of one int per thread over a 256-thread workgroup (4 waves): shuffle scan inside
// each wave, wave totals through 4 LDS words.  `s_tmp` needs 5 ints; *total receives the block sum.
__device__ __forceinline__ int ln_block_excl_scan_256(int v, int* s_tmp, int* total) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_tmp[wave] = incl;
    __syncthreads();
    int wave_off = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < wave) wave_off += s_tmp[k];
    *total = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
    __syncthreads();
    return wave_off + incl - v;
}
// The same over WAVES waves (s_tmp: WAVES ints), with LDS-only barriers: global operations of the caller stay in flight across it.
template <int WAVES>
__device__ __forceinline__ int ln_block_excl_scan(int v, int* s_tmp, int* total) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int incl = ln_wave_incl_scan(v);
    if (lane == 63) s_tmp[wave] = incl;
    ln_lds_barrier();
    int wave_off = 0, sum = 0;
#pragma unroll
    for (int k = 0; k < WAVES; ++k) {
        const int w = s_tmp[k];
        sum += w;
        if (k < wave) wave_off += w;
    }
    *total = sum;
    ln_lds_barrier();
    return wave_off + incl - v;
}
#endif

#if defined(__HIPCC__)
// out[g] (+)= sum over slabs of partial[s * stride + g], g < total (outputs g >= split go to out2[g - split] when out2 is given).  16 outputs per workgroup, the slabs split over 16
// thread rows and combined through LDS (a single thread walking hundreds of slabs is latency-bound).  Launch with
// grid = ceil(total / 16), block = 256.
template <bool ACCUMULATE>
__device__ __forceinline__ void ln_sum_slabs_body(int block_x, const float* __restrict__ partial, int nslabs, long long stride, int total,
                                                  float* __restrict__ out, float* __restrict__ out2 = nullptr, int split = 0) {
    __shared__ float s_part[16][17];
    const int o = threadIdx.x & 15;
    const int part = threadIdx.x >> 4;
    const int g = block_x * 16 + o;
    float acc = 0.0f;
    if (g < total)
        for (int s0 = part; s0 < nslabs; s0 += 16 * 8) {  // eight loads in flight per thread (one at a time: ~10 us for 512 slabs)
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int sl = s0 + 16 * k;
                v[k] = sl < nslabs ? partial[(size_t)sl * stride + g] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
        }
    s_part[part][o] = acc;
    __syncthreads();
    if (part == 0 && g < total) {
        float r = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) r += s_part[k][o];
        float* dst = (out2 && g >= split) ? out2 + (g - split) : out + g;
        *dst = ACCUMULATE ? *dst + r : r;
    }
}

template <bool ACCUMULATE>
__global__ void __launch_bounds__(256)
    ln_k_sum_slabs(const float* __restrict__ partial, int nslabs, long long stride, int total, float* __restrict__ out) {
    ln_sum_slabs_body<ACCUMULATE>(blockIdx.x, partial, nslabs, stride, total, out);
}
// the same with the outputs g >= split going to out2[g - split] (one launch for a weight block and its bias row)
template <bool ACCUMULATE>
__global__ void __launch_bounds__(256)
    ln_k_sum_slabs2(const float* __restrict__ partial, int nslabs, long long stride, int total, float* __restrict__ out,
                    float* __restrict__ out2, int split) {
    ln_sum_slabs_body<ACCUMULATE>(blockIdx.x, partial, nslabs, stride, total, out, out2, split);
}
#endif

// Experiment knob (tools only): LN_DEBUG_MASK in the environment, read once.  0 in production.
int ln_debug_mask();

#define LN_SC_MAX_SLABS 768  // workgroups of a slice-classify backward launch = per-workgroup slabs of classifier gradients in its workspace
static inline int ln_div_up(long long a, long long b) { return int((a + b - 1) / b); }
