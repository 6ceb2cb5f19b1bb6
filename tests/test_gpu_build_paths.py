"""The two table builds behind ln_build_splat / ln_distribute — LDS-staged buckets (default after a clear) and
global atomics (LN_BUILD_ATOMIC_PATH; also the replay target when a bucket overflows) — must produce the same
rows, indices, weights, keys and slot adjacency, and both must match the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import lattice_oracle as O

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def N(t):
    return t.detach().cpu().numpy()


def make_lattice(sigma, capacity, d=3):
    from lattice_net_amd import Lattice
    return Lattice(sigmas=[float(sigma)] * d, capacity=int(capacity), device=dev())


def build(pos_np, sigma, cap, atomic, monkeypatch, vals_np=None):
    import lattice_net_amd.lattice as LM
    monkeypatch.setattr(LM, "_FORCE_ATOMIC_BUILD", bool(atomic))
    lat = make_lattice(sigma, cap, pos_np.shape[1])
    lat.begin_splat()
    if vals_np is None:
        idx, w = lat.just_create_verts(T(pos_np), True)
    else:
        idx, w = lat.splat_standalone(T(pos_np), T(vals_np))
    status_before_read = int(lat.m_hash_table._counters.tolist()[1])
    m = lat.nr_lattice_vertices()
    return lat, idx, w, m, status_before_read


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6])
def test_bucketed_and_atomic_builds_agree_with_the_oracle(d, monkeypatch):
    rng = np.random.default_rng(100 + d)
    n = 3000
    pos_np = (rng.random((n, d), dtype=np.float32) * 2 - 1).astype(np.float32)
    sigma, cap = 0.3, 60000
    t = O.OracleHashTable(cap, d)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((d,), sigma, np.float32)))
    for atomic in (False, True):
        lat, idx, w, m, _ = build(pos_np, sigma, cap, atomic, monkeypatch)
        assert m == t.nr_filled
        np.testing.assert_array_equal(N(idx), oidx)
        np.testing.assert_array_equal(N(w), ow)
        np.testing.assert_array_equal(N(lat.hash_table().m_keys_tensor[:m]), t.keys[:m])
        counts = N(lat.vertex_point_counts(idx))
        np.testing.assert_array_equal(counts, np.bincount(oidx[oidx >= 0], minlength=m))


def test_lidar_scan_same_rows_and_splat_values_on_both_paths(monkeypatch):
    from lattice_net_amd.synthetic import lidar_cloud
    pos_np = lidar_cloud(120000, 3)
    vals_np = np.random.default_rng(0).standard_normal((120000, 8)).astype(np.float32)
    res = [build(pos_np, 0.9, 100000, atomic, monkeypatch, vals_np) for atomic in (False, True)]
    (la, ia, wa, ma, _), (lb, ib, wb, mb, _) = res
    assert ma == mb and ma > 10000
    assert torch.equal(ia, ib) and torch.equal(wa, wb)
    assert torch.equal(la.hash_table().m_keys_tensor[:ma], lb.hash_table().m_keys_tensor[:mb])
    va, vb = N(la.values()[:ma]), N(lb.values()[:mb])
    np.testing.assert_allclose(va, vb, rtol=1e-5, atol=1e-5 * np.abs(vb).max())
    # neighbour lists come from probing the table: both slot layouts must answer every query the same way
    assert torch.equal(la.neighbours(None, 1, False), lb.neighbours(None, 1, False))


def test_bucket_overflow_is_replayed_on_the_atomic_path(monkeypatch):
    """Table loaded to ~0.99: some 512-slot bucket fills up, the build flags it, and nr_lattice_vertices() redoes
    the build (and the splat queued behind it) with global atomics."""
    rng = np.random.default_rng(5)
    pos_np = (rng.random((6000, 3), dtype=np.float32) * 4).astype(np.float32)
    sigma = 0.1
    probe = O.OracleHashTable(200000, 3)
    oidx, ow = O.build_splat(probe, O.scale_positions(pos_np, np.full((3,), sigma, np.float32)))
    cap = int(probe.nr_filled / 0.99)
    vals_np = rng.standard_normal((6000, 4)).astype(np.float32)
    lat, idx, w, m, status = build(pos_np, sigma, cap, False, monkeypatch, vals_np)
    from lattice_net_amd import _lib
    assert status & _lib.LN_STATUS_BUCKET_OVERFLOW, "test cloud no longer overflows a bucket: raise the load"
    assert int(lat.m_hash_table._counters.tolist()[1]) == 0
    assert m == probe.nr_filled
    np.testing.assert_array_equal(N(idx), oidx)
    np.testing.assert_array_equal(N(w), ow)
    expect = np.zeros((m, 4), np.float32)
    O.splat_accumulate(expect, vals_np, oidx, ow)
    got = N(lat.values()[:m])
    np.testing.assert_allclose(got, expect, rtol=1e-5, atol=1e-5 * np.abs(expect).max())
    # and the neighbour prefetch was redone against the rebuilt table.  At this load the 300-probe retrieval cap
    # (HashTableGPU.cuh:494) makes a few lookups depend on the race-dependent slot layout: a vertex that IS found
    # must be the right one
    lat2, *_ = build(pos_np, sigma, cap, True, monkeypatch, vals_np)
    a, b = N(lat.neighbours(None, 1, False)), N(lat2.neighbours(None, 1, False))
    both = (a >= 0) & (b >= 0)
    np.testing.assert_array_equal(a[both], b[both])
    assert (a == b).mean() > 0.9


def test_distribute_rows_identical_on_both_paths(monkeypatch):
    import lattice_net_amd.lattice as LM
    from lattice_net_amd.synthetic import cube_cloud
    pos_np = cube_cloud(5000, 2)
    vals_np = np.random.default_rng(1).standard_normal((5000, 3)).astype(np.float32)
    out = []
    for atomic in (False, True):
        monkeypatch.setattr(LM, "_FORCE_ATOMIC_BUILD", atomic)
        lat = make_lattice(0.1, 80000)
        new, dist, idx, w = lat.distribute(T(pos_np), T(vals_np))
        out.append((new.nr_lattice_vertices(), dist, idx, w, new.hash_table().m_keys_tensor.clone()))
    assert out[0][0] == out[1][0]
    for a, b in zip(out[0][1:], out[1][1:]):
        assert torch.equal(a, b)


def test_degenerate_cloud_overflows_a_bucket_region_and_is_replayed(monkeypatch):
    """All points identical: d+1 vertices receive every token, far beyond the 16x-mean bucket regions.  The build flags
    it and the replay on the atomic path produces the canonical result."""
    n = 60000
    pos_np = np.tile(np.array([[0.3, -0.2, 0.7]], np.float32), (n, 1))
    vals_np = np.ones((n, 2), np.float32)
    lat, idx, w, m, status = build(pos_np, 0.5, 50000, False, monkeypatch, vals_np)
    from lattice_net_amd import _lib
    assert status & _lib.LN_STATUS_BUCKET_OVERFLOW
    assert m == 4
    t = O.OracleHashTable(50000, 3)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np[:8], np.full((3,), 0.5, np.float32)))
    np.testing.assert_array_equal(N(idx).reshape(n, 4), np.tile(oidx.reshape(8, 4)[:1], (n, 1)))
    np.testing.assert_array_equal(N(w).reshape(n, 4)[:8], ow.reshape(8, 4))
    got = N(lat.values()[:4])
    np.testing.assert_allclose(got[:, 0], n * ow.reshape(8, 4)[0], rtol=1e-4)


def test_tiny_tables_and_empty_clouds(monkeypatch):
    import lattice_net_amd.lattice as LM
    monkeypatch.setattr(LM, "_FORCE_ATOMIC_BUILD", False)
    # capacity 2: too small for the bucket cursors -> atomic build behind the scenes; one point needs 4 slots -> overflow error
    lat = make_lattice(0.5, 2)
    lat.begin_splat()
    lat.just_create_verts(T(np.zeros((1, 3), np.float32)), False)
    from lattice_net_amd import LatticeNetHipError
    with pytest.raises(LatticeNetHipError, match="overflow"):
        lat.nr_lattice_vertices()
    # capacity 8 holds the 4 vertices of one point
    lat = make_lattice(0.5, 8)
    lat.begin_splat()
    idx, w = lat.just_create_verts(T(np.array([[0.1, 0.2, 0.3]], np.float32)), True)
    assert lat.nr_lattice_vertices() == 4 and sorted(N(idx).tolist()) == [0, 1, 2, 3]
    # empty cloud: nothing inserted, table cleared
    lat = make_lattice(0.5, 1000)
    lat.begin_splat()
    idx, w = lat.just_create_verts(T(np.zeros((0, 3), np.float32)), True)
    assert lat.nr_lattice_vertices() == 0 and idx.numel() == 0


def test_back_to_back_builds_report_the_latest_vertex_count():
    """Two builds into the same table without a read in between: the pinned {count, status, sequence} triple must not let
    the first build's late write satisfy the wait for the second."""
    from lattice_net_amd.synthetic import cube_cloud
    a, b = cube_cloud(20000, 1), cube_cloud(300, 2)
    lat = make_lattice(0.1, 400000)
    counts = []
    for _ in range(20):
        lat.begin_splat()
        lat.just_create_verts(T(a), False)
        lat.begin_splat()
        lat.just_create_verts(T(b), False)
        counts.append(lat.nr_lattice_vertices())
    t = O.OracleHashTable(400000, 3)
    O.build_splat(t, O.scale_positions(b, np.full((3,), 0.1, np.float32)))
    assert counts == [t.nr_filled] * 20


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_configurations_against_the_oracle(seed, monkeypatch):
    """Fuzz over dimension, cloud size, sigma and capacity (bucket counts from 1 to hundreds, ragged last buckets, loads up
    to ~0.8): indices, weights, keys, vertex count, neighbour list and splat values against the CPU oracle."""
    import lattice_net_amd.lattice as LM
    monkeypatch.setattr(LM, "_FORCE_ATOMIC_BUILD", False)
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.integers(1, 6))
    n = int(rng.integers(1, 6000))
    sigma = float(rng.choice([0.05, 0.1, 0.3, 1.0]))
    pos_np = ((rng.random((n, d), dtype=np.float32) - 0.5) * float(rng.choice([0.5, 2.0, 8.0]))).astype(np.float32)
    probe = O.OracleHashTable(n * (d + 1) + 8, d)
    oidx, ow = O.build_splat(probe, O.scale_positions(pos_np, np.full((d,), sigma, np.float32)))
    cap = int(probe.nr_filled / float(rng.choice([0.1, 0.3, 0.6, 0.8]))) + int(rng.integers(1, 700))
    v = int(rng.choice([1, 3, 4, 8]))
    vals_np = rng.standard_normal((n, v)).astype(np.float32)
    lat = make_lattice(sigma, cap, d)
    lat.begin_splat()
    idx, w = lat.splat_standalone(T(pos_np), T(vals_np))
    m = lat.nr_lattice_vertices()
    t = O.OracleHashTable(cap, d)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((d,), sigma, np.float32)))
    assert m == t.nr_filled, (d, n, sigma, cap)
    np.testing.assert_array_equal(N(idx), oidx)
    np.testing.assert_array_equal(N(w), ow)
    np.testing.assert_array_equal(N(lat.hash_table().m_keys_tensor[:m]), t.keys[:m])
    expect = np.zeros((m, v), np.float32)
    O.splat_accumulate(expect, vals_np, oidx, ow)
    np.testing.assert_allclose(N(lat.values()[:m]), expect, rtol=1e-5, atol=1e-5 * max(float(np.abs(expect).max()), 1e-30))
    np.testing.assert_array_equal(N(lat.neighbours(None, 1, False)), O.neighbour_rows(t.keys[:m], t, 1, 1, 1, False))


@pytest.mark.parametrize("cap,atomic", [(9_000_000, True), (9_000_000, False), (16_500_000, False)])
def test_very_large_tables_number_rows_like_a_small_one(cap, atomic, monkeypatch):
    """Row numbering (first occurrence in point order), weights and keys do not depend on the capacity, so a 9M / 16.5M-slot
    table must reproduce the 200k-slot build bit for bit.  9M slots on the atomic path takes the CSR offsets from the
    single-workgroup top-level scan (k_csr_scan_top); 16.5M slots is past what one bucket can stage in LDS, so the
    library takes the atomic path by itself."""
    rng = np.random.default_rng(77)
    pos_np = ((rng.random((40000, 3), dtype=np.float32) - 0.5) * 6).astype(np.float32)
    vals_np = rng.standard_normal((40000, 4)).astype(np.float32)
    import lattice_net_amd as L
    small, si, sw, sm, _ = build(pos_np, 0.2, 200000, False, monkeypatch, vals_np)
    prev = L.set_hash_capacity_policy("full")  # hash into ALL slots (the default would use 2 x tokens = 320 k of them)
    try:
        big, bi, bw, bm, _ = build(pos_np, 0.2, cap, atomic, monkeypatch, vals_np)
        assert big.hash_table()._storage.hashed() == cap
    finally:
        L.set_hash_capacity_policy(prev)
    assert bm == sm and sm > 20000
    assert torch.equal(bi, si) and torch.equal(bw, sw)
    assert torch.equal(big.hash_table().m_keys_tensor[:bm], small.hash_table().m_keys_tensor[:sm])
    np.testing.assert_allclose(N(big.values()[:bm]), N(small.values()[:sm]), rtol=1e-5, atol=1e-5)
    assert torch.equal(big.neighbours(None, 1, False), small.neighbours(None, 1, False))


@pytest.mark.parametrize("d,reach", [(5, 9000), (6, 3000)])
def test_lattice_key_format_covers_wide_clouds(d, reach):
    """pos_dim 5 / 6 (e.g. xyz+rgb positions): the packed key of a table built from positions stores the shared remainder and
    the quotients, so lattice coordinates of several thousand units fit (the raw format stops at +-2048 / +-512).  Indices,
    weights and keys must still match the oracle bit for bit; beyond the range the build reports it."""
    import lattice_net_amd as L

    rng = np.random.default_rng(50 + d)
    n, cap = 3000, 40000
    # keys scale linearly with the positions: measure them at one scale, then stretch the cloud so that the largest lattice
    # coordinate lands at `reach` units — beyond the raw format, inside the lattice format
    unit = rng.uniform(-1.0, 1.0, (n, d)).astype(np.float32)
    t0 = O.OracleHashTable(cap, d)
    O.build_splat(t0, O.scale_positions(unit * 100.0, np.ones((d,), np.float32)), False)
    spread = 100.0 * reach / float(np.abs(t0.keys[: t0.nr_filled]).max())
    pos_np = (unit * spread).astype(np.float32)
    t = O.OracleHashTable(cap, d)
    idx_ref, w_ref = O.build_splat(t, O.scale_positions(pos_np, np.ones((d,), np.float32)))
    kmax = int(np.abs(t.keys[: t.nr_filled]).max())
    assert (2048 if d == 5 else 512) < kmax < (12288 if d == 5 else 3584), kmax
    lat = L.Lattice(sigmas=[1.0] * d, capacity=cap, device=dev())
    lat.begin_splat()
    idx, w = lat.just_create_verts(T(pos_np), True)
    assert lat.nr_lattice_vertices() == t.nr_filled
    assert np.array_equal(N(idx), idx_ref)
    assert np.array_equal(N(w), w_ref)
    assert np.array_equal(N(lat.m_hash_table.m_keys_tensor)[: t.nr_filled], t.keys[: t.nr_filled])
    # neighbour lookups go through the same format
    nbr_ref = O.neighbour_rows(t.keys[: t.nr_filled], t, 1, 1, 1, False)
    assert np.array_equal(N(lat.neighbours(lat, 1, False)), nbr_ref)
    # far beyond the range: reported, not silently wrong
    far = L.Lattice(sigmas=[1.0] * d, capacity=cap, device=dev())
    far.begin_splat()
    far.just_create_verts(T(pos_np * 50.0), True)
    with pytest.raises(L.LatticeNetHipError, match="packed 64-bit"):
        far.nr_lattice_vertices()


def test_builds_hash_into_what_the_cloud_needs_and_grow_on_demand():
    """A build that starts from a cleared table hashes into min(capacity, max(16384, 2 x tokens)) slots of a larger table (same
    rows, keys, neighbours as with all slots in use); the range never shrinks; an incremental build first re-hashes the existing
    vertices into the whole table (ln_rehash), after which the cfg's capacity bounds what can be inserted."""
    import lattice_net_amd as L
    rng = np.random.default_rng(9)
    pos_np = ((rng.random((3000, 3), dtype=np.float32) - 0.5) * 8).astype(np.float32)
    more_np = ((rng.random((60000, 3), dtype=np.float32) - 0.5) * 30).astype(np.float32)
    cap = 2_000_000
    lat = make_lattice(0.3, cap)
    lat.begin_splat()
    idx, w = lat.just_create_verts(T(pos_np), True)
    m = lat.nr_lattice_vertices()
    st = lat.hash_table()._storage
    assert st.hashed() == max(16384, 2 * 12000) and lat.capacity() == cap
    t = O.OracleHashTable(cap, 3)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((3,), 0.3, np.float32)))
    assert m == t.nr_filled
    np.testing.assert_array_equal(N(idx), oidx)
    np.testing.assert_array_equal(N(lat.hash_table().m_keys_tensor[:m]), t.keys[:m])
    assert not N(lat.hash_table().m_keys_tensor[m:m + 4096]).any()
    np.testing.assert_array_equal(N(lat.neighbours(None, 1, False)), O.neighbour_rows(t.keys[:m], t, 1, 1, 1, False))
    # incremental build (no reset): 240 k more tokens than the 24 k-slot range could ever hold
    lat.begin_splat(reset_hashmap=False)
    idx2, w2 = lat.just_create_verts(T(more_np), True)
    m2 = lat.nr_lattice_vertices()
    assert st.hashed() == cap
    oidx2, ow2 = O.build_splat(t, O.scale_positions(more_np, np.full((3,), 0.3, np.float32)))
    assert m2 == t.nr_filled and m2 > 24000
    np.testing.assert_array_equal(N(idx2), oidx2)
    np.testing.assert_array_equal(N(lat.hash_table().m_keys_tensor[:m2]), t.keys[:m2])
    np.testing.assert_array_equal(N(lat.neighbours(None, 1, False)), O.neighbour_rows(t.keys[:m2], t, 1, 1, 1, False))
    # a fresh build of the small cloud afterwards keeps the grown range (it never shrinks) and still matches
    lat.begin_splat()
    idx3, _ = lat.just_create_verts(T(pos_np), True)
    assert lat.nr_lattice_vertices() == m and st.hashed() == cap
    np.testing.assert_array_equal(N(idx3), oidx)
    assert not N(lat.hash_table().m_keys_tensor[m:m2 + 16]).any(), "rows of the larger earlier build must be zero again"


def test_static_rows_builds_hash_into_a_range_that_follows_the_row_bound():
    """Static-rows mode: a build hashes into 2.3 x the row bound when that is less than 2 x its tokens (many tokens per vertex:
    ScanNet-like scenes) — the range SHRINKS once, on the first static build outside a capture, the slots given up are emptied,
    rows / keys / splat values still match the oracle, and leaving the mode lets the next eager build grow the range again."""
    rng = np.random.default_rng(11)
    # 30 k points on a coarse lattice: 120 k tokens on a few thousand vertices
    pos_np = ((rng.random((30000, 3), dtype=np.float32) - 0.5) * 6).astype(np.float32)
    vals_np = rng.standard_normal((30000, 8)).astype(np.float32)
    sig = 0.25
    cap = 1_000_000
    lat = make_lattice(sig, cap)
    lat.begin_splat()
    lat.splat_standalone(T(pos_np), T(vals_np))
    m = lat.nr_lattice_vertices()
    st = lat.hash_table()._storage
    assert st.hashed() == 240000 and m * 2.3 < 100000
    bound = ((int(m * 1.06) + 255) // 256) * 256
    lat.set_static_rows(bound)
    lat.begin_splat()
    idx, w = lat.splat_standalone(T(pos_np), T(vals_np))
    want = max(16384, int(2.3 * bound) + 1)
    assert st.hashed() == want
    assert lat.static_build_report()[0] == m
    assert (N(st.entries[want:240000]) == -1).all(), "slots outside the new range must read as empty"
    t = O.OracleHashTable(cap, 3)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((3,), sig, np.float32)))
    np.testing.assert_array_equal(N(idx), oidx)
    np.testing.assert_array_equal(N(lat.hash_table().m_keys_tensor[:m]), t.keys[:m])
    ovals = np.zeros((m, 8), np.float64)
    np.add.at(ovals, oidx, np.repeat(vals_np.astype(np.float64), 4, axis=0) * ow[:, None])
    got = N(lat.values())
    assert got.shape[0] == bound
    np.testing.assert_allclose(got[:m], ovals, rtol=1e-5, atol=1e-5 * np.abs(ovals).max())
    assert not got[m:].any()
    np.testing.assert_array_equal(N(lat.neighbours(None, 1, False))[:m], O.neighbour_rows(t.keys[:m], t, 1, 1, 1, False))
    # back to eager mode: the next build from a cleared table hashes into 2 x tokens again and still matches
    lat.set_static_rows(None)
    lat.begin_splat()
    idx2, _ = lat.splat_standalone(T(pos_np), T(vals_np))
    assert st.hashed() == 240000 and lat.nr_lattice_vertices() == m
    np.testing.assert_array_equal(N(idx2), oidx)


def test_cloud_far_beyond_the_static_bound_fills_the_shrunk_range_without_hanging():
    """A cloud with 80x the vertices of the static bound meets a hashed range (2.3 x bound slots) it cannot fit into: the build
    must come back (no endless probing), flag the step through the report word, hand out no row >= bound, and the same lattice
    must build the cloud correctly once static-rows mode is left."""
    rng = np.random.default_rng(3)
    small = ((rng.random((2000, 3), dtype=np.float32) - 0.5) * 4).astype(np.float32)
    big = ((rng.random((60000, 3), dtype=np.float32) - 0.5) * 40).astype(np.float32)
    lat = make_lattice(0.3, 1_000_000)
    lat.begin_splat()
    lat.splat_standalone(T(small), T(rng.standard_normal((2000, 8)).astype(np.float32)))
    m = lat.nr_lattice_vertices()
    bound = ((int(m * 1.06) + 255) // 256) * 256
    lat.set_static_rows(bound)
    lat.begin_splat()
    lat.splat_standalone(T(small), T(rng.standard_normal((2000, 8)).astype(np.float32)))
    torch.cuda.synchronize()
    assert lat.static_build_report() == (m, 0) and lat.hash_table()._storage.hashed() == 16384
    vb = T(rng.standard_normal((60000, 8)).astype(np.float32))
    lat.begin_splat()
    idx, _ = lat.splat_standalone(T(big), vb)
    torch.cuda.synchronize()
    import lattice_net_amd as L
    with pytest.raises(L.LatticeNetHipError):
        lat.static_build_report()
    assert int(idx.max()) < bound and int((idx < 0).sum()) > 0
    lat.set_static_rows(None)
    lat.begin_splat()
    idx2, _ = lat.splat_standalone(T(big), vb)
    m2 = lat.nr_lattice_vertices()
    t = O.OracleHashTable(1_000_000, 3)
    oidx, _ = O.build_splat(t, O.scale_positions(big, np.full((3,), 0.3, np.float32)))
    assert m2 == t.nr_filled and m2 > 50 * bound
    np.testing.assert_array_equal(N(idx2), oidx)


@pytest.mark.parametrize("d,n,sigma,cap", [(3, 120000, 0.9, 100000), (3, 2500, 0.05, 60000), (2, 40000, 0.3, 30000), (5, 6000, 0.4, 80000),
                                           (3, 200000, 0.08, 5000000)])
def test_bucket_pass_workgroup_size_is_invisible(d, n, sigma, cap, monkeypatch):
    """ln_build_concurrency (lattice.set_scans_in_flight): with several scans in flight the bucket pass of a build over small buckets runs
    on 512-thread workgroups instead of 1024 (ln_table.hip: k_bucket_rows<D, TH>).  Rows are numbered bucket by bucket either way: indices,
    weights, keys, vertex count, the CSR's per-vertex token counts and the neighbour list must be identical bit for bit — and equal to the
    oracle's in the canonical numbering."""
    import lattice_net_amd.lattice as LM
    from lattice_net_amd.synthetic import lidar_cloud
    rng = np.random.default_rng(7 * d + n)
    pos_np = lidar_cloud(n, 5) if (d == 3 and n == 120000) else (rng.random((n, d), dtype=np.float32) * 2 - 1).astype(np.float32)
    res = []
    for scans in (1, 4):
        prev = LM.set_scans_in_flight(scans)
        try:
            lat, idx, w, m, status = build(pos_np, sigma, cap, False, monkeypatch)
            res.append((idx.clone(), w.clone(), lat.hash_table().m_keys_tensor[:m].clone(), m, status,
                        lat.vertex_point_counts(idx).clone(), lat.neighbours(None, 1, False).clone()))
        finally:
            LM.set_scans_in_flight(prev)
    a, b = res
    assert a[3] == b[3] and a[4] == b[4] == 0
    for x, y in zip(a[:3] + a[5:], b[:3] + b[5:]):
        assert torch.equal(x, y)
    if n <= 40000:  # (the oracle's serial build: small cases only)
        t = O.OracleHashTable(cap, d)
        oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((d,), sigma, np.float32)))
        assert b[3] == t.nr_filled
        np.testing.assert_array_equal(N(b[0]), oidx)
        np.testing.assert_array_equal(N(b[1]), ow)
