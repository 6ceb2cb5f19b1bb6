"""Space-ordered slots (LnTable.planes, round 6) against the oracle (`pytest -m gpu`).

With kd planes bound to a table, a key starts probing in the run of buckets of its kd leaf and rows are numbered bucket by
bucket: the rows of the table follow space.  Nothing reference-visible may change: same vertex SET as the oracle, splat indices
equal under the row permutation that matches the keys, weights / sliced rows / filter gradients equal as they are, retrieval
(neighbour lists, slice_no_precomputation, incremental builds) finds every vertex — in the shipped numbering AND under the
canonical relabelling, for token-balanced, mixed and vertex-balanced planes and deliberately useless ones.  Also here: the row partition the
build hands the convolutions (LnTable.row_regions), the slice that walks the points in CSR order (bit-identical rows), and the
workgroup -> tile map of the vertex-tiled kernels (a bijection whatever the partition array holds)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import lattice_oracle as O

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def dev():
    return torch.device("cuda", 0)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def N(t):
    return t.detach().cpu().numpy()


@pytest.fixture(autouse=True)
def slot_order(canonical_row_order):
    from lattice_net_amd import lattice as lat
    prev = lat.set_row_order("slot")
    prev_s = lat.set_slot_order("space")
    yield
    lat.set_row_order(prev)
    lat.set_slot_order(prev_s)


def row_permutation(keys_gpu, keys_oracle):
    assert keys_gpu.shape == keys_oracle.shape
    og = np.lexsort(keys_gpu.T[::-1])
    oo = np.lexsort(keys_oracle.T[::-1])
    assert np.array_equal(keys_gpu[og], keys_oracle[oo]), "vertex sets differ"
    perm = np.empty(len(og), np.int64)
    perm[og] = oo
    return perm


def leaf_of_keys(keys, planes):
    node = np.zeros(len(keys), np.int64)
    pl = np.asarray(planes, np.int64)
    d = keys.shape[1]
    for lvl in range(3):
        node = 2 * node + 1 + (keys[:, lvl % d] >= pl[node]).astype(np.int64)
    return node - 7


def build_space_ordered(pos_np, sigma, cap, vertex_weight=0.0, planes=None, shares=True):
    """Lattice whose second build runs over a slot map calibrated on the first; returns (lattice, idx, w, planes).  shares=False: the
    planes without the vertex shares of their leaves (equal slot runs)."""
    from lattice_net_amd import Lattice
    lat = Lattice(sigmas=[sigma] * pos_np.shape[1], capacity=cap, device=dev())
    lat.begin_splat()
    idx, w = lat.just_create_verts(T(pos_np), True)
    lat.nr_lattice_vertices()
    if planes is not None:
        lat.set_region_planes(planes)
    elif shares:
        planes = lat.calibrate_regions(idx, vertex_weight=vertex_weight)
    else:
        planes = lat.balanced_region_planes(idx, vertex_weight=vertex_weight)
        lat.set_region_planes(planes)
    lat.begin_splat()
    idx, w = lat.just_create_verts(T(pos_np), True)
    return lat, idx, w, planes


def clouds():
    from lattice_net_amd import synthetic
    rng = np.random.default_rng(5)
    yield "cube4k", rng.uniform(-1, 1, (4000, 3)).astype(np.float32), 0.1, 60000
    yield "lidar20k", synthetic.lidar_cloud(20000, 3), 0.9, 30000
    yield "lidar120k", synthetic.lidar_cloud(120000, 0), 0.9, 100000
    yield "planes200k", synthetic.planes_cloud(200000, 1), 0.08, 5000000


@pytest.mark.parametrize("weights", ["tokens", "mixed", "vertices_equal_runs"])
@pytest.mark.parametrize("case", list(clouds()), ids=lambda c: c[0])
def test_space_ordered_build_equals_oracle_and_rows_follow_the_regions(case, weights):
    _, pos_np, sigma, cap = case
    lat, idx, w, planes = build_space_ordered(pos_np, sigma, cap, vertex_weight={"tokens": 0.0, "mixed": 1.0, "vertices_equal_runs": 1e6}[weights],
                                              shares=weights != "vertices_equal_runs")
    m = lat.nr_lattice_vertices()
    st = lat.m_hash_table._storage
    assert st.slot_map is not None and st.rows_follow_space, "the bucketed build over the calibrated slot map must not overflow"
    smap = N(st.slot_map)
    assert np.array_equal(smap[:7], planes) and smap[8] == 0 and smap[16] <= st.hashed() and np.all(smap[17:25] * smap[7] == np.diff(smap[8:17]))
    t = O.OracleHashTable(cap, 3)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((3,), sigma, np.float32)))
    assert m == t.nr_filled
    keys = N(lat.m_hash_table.m_keys_tensor)
    assert not keys[m:].any()
    perm = row_permutation(keys[:m], t.keys[:m])
    gi = N(idx).astype(np.int64)
    assert gi.min() >= 0 and gi.max() == m - 1
    assert np.array_equal(perm[gi], oidx)
    assert np.array_equal(N(w), ow)
    ent = N(lat.m_hash_table.m_entries_tensor)
    assert np.array_equal(np.sort(ent[ent >= 0]), np.arange(m))
    # rows follow space: the kd region of the key of row r never decreases with r, the partition the build left says where each of
    # the 8 regions starts, and every vertex sits in the slot run of its region
    region = leaf_of_keys(keys[:m].astype(np.int64), planes)
    assert np.all(np.diff(region) >= 0), "rows must be grouped by kd region"
    slot_of_row = np.empty(m, np.int64)
    slot_of_row[ent[ent >= 0]] = np.nonzero(ent >= 0)[0]
    assert np.all(slot_of_row >= smap[8:16][region]) and np.all(slot_of_row < smap[9:17][region])
    starts = N(st.row_regions)[:9]
    assert starts[0] == 0 and starts[8] == m
    for r in range(8):
        assert starts[r] == int(np.searchsorted(region, r, side="left")), (r, starts)


@pytest.mark.parametrize("d", [1, 2, 4, 5, 6])
def test_space_ordered_build_other_dimensions(d):
    rng = np.random.default_rng(100 + d)
    n, cap, sigma = 8000, 60000, 0.6
    pos_np = (rng.standard_normal((n, d)) * (3.0 if d < 5 else 1.2)).astype(np.float32)
    lat, idx, w, planes = build_space_ordered(pos_np, sigma, cap)
    assert lat.m_hash_table._storage.rows_follow_space
    m = lat.nr_lattice_vertices()
    t = O.OracleHashTable(cap, d)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((d,), sigma, np.float32)))
    assert m == t.nr_filled
    perm = row_permutation(N(lat.m_hash_table.m_keys_tensor)[:m], t.keys[:m])
    assert np.array_equal(perm[N(idx).astype(np.int64)], oidx)
    assert np.array_equal(N(w), ow)
    # retrieval over the space-ordered table: the same-level neighbour list, relabelled, is the oracle's
    nbr = O.neighbour_rows(t.keys[:m], t, 1, 1, 1, False)
    gn = N(lat.neighbours(lat, 1, False)).astype(np.int64)
    assert np.array_equal(np.where(gn >= 0, perm[np.maximum(gn, 0)], gn)[np.argsort(perm)], nbr)


def test_space_ordered_chain_matches_oracle_and_the_hashed_chain():
    """splat -> conv -> slice forward + backward over a space-ordered table (row partition in both convolution kernels, slice in
    CSR order) against the oracle; sliced rows and weights also against the same chain over hashed slots."""
    import lattice_net_amd as L
    from lattice_net_amd import synthetic
    n, v, f, sigma, cap = 60000, 32, 32, 0.9, 100000
    pos_np = synthetic.lidar_cloud(n, 1)
    rng = np.random.default_rng(1)
    vals_np = rng.standard_normal((n, v)).astype(np.float32)
    w_np = (rng.standard_normal((9 * v, f)) / np.sqrt(9 * v)).astype(np.float32)
    g_np = rng.standard_normal((n, f)).astype(np.float32)
    pos, vals = T(pos_np), T(vals_np)
    results = {}
    for mode in ("hash", "space"):
        W = T(w_np).requires_grad_(True)
        lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev())
        lv, wrap, idx, w = L.SplatLattice.apply(lat, pos, vals)
        lat.nr_lattice_vertices()
        if mode == "space":
            lat.calibrate_regions(idx)
            lv, wrap, idx, w = L.SplatLattice.apply(lat, pos, vals)
            assert lat.m_hash_table._storage.rows_follow_space
        m = lat.nr_lattice_vertices()
        lv = lv[:m].contiguous().requires_grad_(True)
        cv, cwrap = L.ConvIm2RowLattice.apply(lv, lat, W, 1)
        out = L.SliceLattice.apply(cv, cwrap.lattice, pos, idx, w)
        out.backward(T(g_np))
        torch.cuda.synchronize()
        results[mode] = dict(lat=lat, m=m, lv=N(lv), idx=N(idx), w=N(w), out=N(out), gw=N(W.grad), gv=N(lv.grad))
    r = results["space"]
    lat, m = r["lat"], r["m"]
    t = O.OracleHashTable(cap, 3)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((3,), sigma, np.float32)))
    assert t.nr_filled == m
    perm = row_permutation(N(lat.m_hash_table.m_keys_tensor)[:m], t.keys[:m])
    assert np.array_equal(perm[r["idx"].astype(np.int64)], oidx) and np.array_equal(r["w"], ow)
    ov = np.zeros((m, v), np.float32)
    O.splat_accumulate(ov, vals_np, oidx, ow)
    nbr = O.neighbour_rows(t.keys[:m], t, 1, 1, 1, False)
    oc = O.conv_forward(nbr, ov, w_np)
    oo = O.slice_with_precomputation(oc, oidx, ow, n)

    def close(a, b):
        np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=RTOL, atol=RTOL * float(np.max(np.abs(b))))

    close(r["lv"][np.argsort(perm)], ov)
    gn = N(lat.neighbours(lat, 1, False)).astype(np.int64)
    assert np.array_equal(np.where(gn >= 0, perm[np.maximum(gn, 0)], gn)[np.argsort(perm)], nbr)
    close(r["out"], oo)
    g_c = O.slice_backwards(g_np, oidx, ow, m)
    rows = O.im2row(nbr, ov).astype(np.float64)
    close(r["gw"], rows.T @ g_c.astype(np.float64))
    wb = w_np.reshape(9, v, f)
    gv = np.zeros((m, v), np.float64)
    for e in range(9):
        ok = nbr[:, e] >= 0
        np.add.at(gv, nbr[ok, e], g_c[ok].astype(np.float64) @ wb[e].T.astype(np.float64))
    close(r["gv"][np.argsort(perm)], gv)
    h = results["hash"]
    assert np.array_equal(h["w"], r["w"])
    close(h["out"], r["out"])
    close(h["gw"], r["gw"])


@pytest.mark.parametrize("v", [4, 8, 32, 64, 128, 20])
def test_ordered_slice_rows_are_bit_identical_to_the_plain_slice(v):
    """ln_slice_forward_ordered against ln_slice_forward on the same table, indices and weights: every output row bit for bit, and the
    accumulator it is asked to clear is cleared (widths the ordered kernel does not cover fall back to the plain one)."""
    import lattice_net_amd as L
    from lattice_net_amd import _lib, synthetic
    n, sigma, cap = 30000, 0.9, 60000
    pos_np = synthetic.lidar_cloud(n, 8)
    lat, idx, w, _ = build_space_ordered(pos_np, sigma, cap)
    m = lat.nr_lattice_vertices()
    st = lat.m_hash_table._storage
    rng = np.random.default_rng(v)
    values = T(rng.standard_normal((m, v)).astype(np.float32))
    lib = L.load_library()
    hit = st.csr_cache[(idx.data_ptr(), idx._version, idx.numel())]
    t = lat.m_hash_table.c_table()
    plain = torch.empty((n, v), dtype=torch.float32, device=dev())
    ordered = torch.full((n, v), float("nan"), dtype=torch.float32, device=dev())
    acc = torch.full((m * v,), 7.0, dtype=torch.float32, device=dev())
    _lib.check(lib.ln_slice_forward(_lib.ptr(values), _lib.ptr(idx), _lib.ptr(w), n, 3, v, _lib.ptr(plain), _lib.stream_ptr(dev())))
    _lib.check(lib.ln_slice_forward_ordered(C.byref(t), C.byref(hit[1]), _lib.ptr(values), _lib.ptr(idx), _lib.ptr(w), n, v, _lib.ptr(ordered),
                                            _lib.ptr(acc), acc.numel(), _lib.stream_ptr(dev())))
    torch.cuda.synchronize()
    assert np.array_equal(N(plain).view(np.uint32), N(ordered).view(np.uint32))
    assert not N(acc).any()
    assert np.array_equal(N(plain), O.slice_with_precomputation(N(values), N(idx), N(w), n))  # (oracle: the same order of operations)


def test_ordered_slice_reaches_points_whose_first_token_is_not_in_the_csr():
    """A point far outside the packable key range never reaches a bucket: its tokens are missing from the build's CSR and the status word
    says so — the ordered slice then finds such points by their idx[p * (d + 1)] < 0 and writes their (all-absent: zero) rows too."""
    import lattice_net_amd as L
    from lattice_net_amd import _lib, synthetic
    n, sigma, cap, v = 5000, 0.9, 30000, 32
    pos_np = synthetic.lidar_cloud(n, 9)
    pos_np[[7, 4000]] = [[3.0e6, 1.0, 1.0], [1.0, -3.0e6, 2.0]]
    lat, idx0, w0, _ = build_space_ordered(synthetic.lidar_cloud(n, 9), sigma, cap)
    m = lat.nr_lattice_vertices()
    lat.set_static_rows(m + 256)  # (static rows: the host does not read — and raise on — the status word)
    lat.begin_splat()
    idx, w = lat.just_create_verts(T(pos_np), True)
    torch.cuda.synchronize()
    gi = N(idx).reshape(n, 4)
    assert (gi[[7, 4000]] == -1).all() and (gi[[0, 1, 2]] >= 0).all()
    st = lat.m_hash_table._storage
    hit = st.csr_cache[(idx.data_ptr(), idx._version, idx.numel())]
    rows = m + 256
    values = T(np.random.default_rng(0).standard_normal((rows, v)).astype(np.float32))
    lib = L.load_library()
    t = lat.m_hash_table.c_table()
    plain = torch.empty((n, v), dtype=torch.float32, device=dev())
    ordered = torch.full((n, v), float("nan"), dtype=torch.float32, device=dev())
    _lib.check(lib.ln_slice_forward(_lib.ptr(values), _lib.ptr(idx), _lib.ptr(w), n, 3, v, _lib.ptr(plain), _lib.stream_ptr(dev())))
    _lib.check(lib.ln_slice_forward_ordered(C.byref(t), C.byref(hit[1]), _lib.ptr(values), _lib.ptr(idx), _lib.ptr(w), n, v, _lib.ptr(ordered),
                                            None, 0, _lib.stream_ptr(dev())))
    torch.cuda.synchronize()
    assert np.array_equal(N(plain).view(np.uint32), N(ordered).view(np.uint32))
    assert not N(ordered)[[7, 4000]].any()
    with pytest.raises(L.LatticeNetHipError):
        lat.static_build_report()
    lat.set_static_rows(None)


def test_useless_planes_are_survived_in_eager_mode():
    """Planes that put every key into ONE leaf overfill that leaf's buckets: the bucketed build reports it and the eager path replays
    the build (and what was queued behind it) on the atomic path over HASHED slots — the map is dropped, because spilling past full
    buckets would leave keys beyond the reach of the 300-probe retrieval (HashTableGPU.cuh:494).  Same lattice, retrieval included."""
    import lattice_net_amd as L
    from lattice_net_amd import synthetic
    n, v, sigma, cap = 20000, 32, 0.9, 40000
    pos_np = synthetic.lidar_cloud(n, 21)
    rng = np.random.default_rng(7)
    vals_np = rng.standard_normal((n, v)).astype(np.float32)
    t = O.OracleHashTable(cap, 3)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((3,), sigma, np.float32)))
    m = t.nr_filled
    ov = np.zeros((m, v), np.float32)
    O.splat_accumulate(ov, vals_np, oidx, ow)
    nbr = O.neighbour_rows(t.keys[:m], t, 1, 1, 1, False)
    for planes in ([10 ** 6] * 7, [-10 ** 6] * 7):
        lat, _, _, _ = build_space_ordered(pos_np, sigma, cap, planes=planes)
        lv, _, idx, w = L.SplatLattice.apply(lat, T(pos_np), T(vals_np))
        assert lat.nr_lattice_vertices() == m
        st = lat.m_hash_table._storage
        assert st.slot_map is None and not st.rows_follow_space, "the map that did not fit must be gone after the replay"
        perm = row_permutation(N(lat.m_hash_table.m_keys_tensor)[:m], t.keys[:m])
        assert np.array_equal(perm[N(idx).astype(np.int64)], oidx)
        np.testing.assert_allclose(N(lv)[:m][np.argsort(perm)], ov, rtol=RTOL, atol=RTOL * float(np.abs(ov).max()))
        gn = N(lat.neighbours(lat, 1, False)).astype(np.int64)
        assert np.array_equal(np.where(gn >= 0, perm[np.maximum(gn, 0)], gn)[np.argsort(perm)], nbr)
        lat.set_values(lv[:m].contiguous())
        out = lat.slice_standalone_no_precomputation(T(pos_np))[0]  # retrieval of every simplex vertex
        np.testing.assert_allclose(N(out), O.slice_with_precomputation(ov[np.argsort(np.argsort(perm))] if False else N(lv)[:m][np.argsort(perm)], oidx, ow, n),
                                   rtol=RTOL, atol=RTOL * float(np.abs(ov).max()))


def test_incremental_build_and_retrieval_keep_the_planes_of_the_contents():
    """Planes set AFTER a build do not touch the table's contents: retrieval and an incremental build (reset_hashmap = False) go on
    with the slot function the contents were inserted under; the next build that clears adopts the new planes."""
    import lattice_net_amd as L
    from lattice_net_amd import synthetic
    n, sigma, cap = 12000, 0.9, 60000
    a_np, b_np = synthetic.lidar_cloud(n, 31), synthetic.lidar_cloud(n, 32)
    lat, idx, w, planes = build_space_ordered(a_np, sigma, cap)
    m_a = lat.nr_lattice_vertices()
    lat.set_region_planes([5, -7, 9, 1, 2, 3, 4])  # for the NEXT fresh build
    st = lat.m_hash_table._storage
    assert np.array_equal(N(st.slot_map)[:7], np.asarray(planes))
    lat.begin_splat(reset_hashmap=False)
    idx_b, w_b = lat.just_create_verts(T(b_np), True)  # no clear in front of it: an incremental build (atomic path)
    m_ab = lat.nr_lattice_vertices()
    t = O.OracleHashTable(cap, 3)
    sig = np.full((3,), sigma, np.float32)
    O.build_splat(t, O.scale_positions(a_np, sig))
    oidx_b, ow_b = O.build_splat(t, O.scale_positions(b_np, sig))
    assert m_ab == t.nr_filled and m_ab > m_a
    perm = row_permutation(N(lat.m_hash_table.m_keys_tensor)[:m_ab], t.keys[:m_ab])
    assert np.array_equal(perm[N(idx_b).astype(np.int64)], oidx_b)
    nbr = O.neighbour_rows(t.keys[:m_ab], t, 1, 1, 1, False)
    gn = N(lat.neighbours(lat, 1, False)).astype(np.int64)
    assert np.array_equal(np.where(gn >= 0, perm[np.maximum(gn, 0)], gn)[np.argsort(perm)], nbr)
    lat.begin_splat()
    lat.just_create_verts(T(a_np), True)
    assert lat.nr_lattice_vertices() == m_a
    assert np.array_equal(N(st.slot_map)[:7], np.asarray([5, -7, 9, 1, 2, 3, 4]))


def test_canonical_numbering_over_space_ordered_slots_is_the_oracles():
    """set_row_order("canonical") over a space-ordered table: indices, keys and the neighbour list equal the oracle's bit for bit (the
    relabelling pass does not care how the slots are ordered), and the convolutions get no row partition."""
    from lattice_net_amd import lattice as LT, synthetic
    LT.set_row_order("canonical")
    pos_np, sigma, cap = synthetic.lidar_cloud(30000, 11), 0.9, 60000
    lat, idx, w, _ = build_space_ordered(pos_np, sigma, cap)
    m = lat.nr_lattice_vertices()
    assert lat.m_hash_table._storage.slot_map is not None and lat._row_partition() is None
    t = O.OracleHashTable(cap, 3)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((3,), sigma, np.float32)))
    assert m == t.nr_filled
    assert np.array_equal(N(idx), oidx) and np.array_equal(N(w), ow)
    assert np.array_equal(N(lat.m_hash_table.m_keys_tensor)[:m], t.keys[:m])
    assert np.array_equal(N(lat.neighbours(lat, 1, False)), O.neighbour_rows(t.keys[:m], t, 1, 1, 1, False))


def test_conv_row_partition_is_a_placement_hint_only():
    """The convolution kernels under arbitrary partition arrays (garbage, all zero, reversed): ln_partition_tile is a bijection whatever it
    reads, so forward and both gradients come out bit-identical to the run without a partition."""
    import lattice_net_amd as L
    from lattice_net_amd import _lib, synthetic
    n, v, f, sigma, cap = 40000, 32, 32, 0.9, 100000
    pos_np = synthetic.lidar_cloud(n, 13)
    lat, idx, w, _ = build_space_ordered(pos_np, sigma, cap)
    m = lat.nr_lattice_vertices()
    rng = np.random.default_rng(13)
    values = T(rng.standard_normal((m, v)).astype(np.float32))
    G = T(rng.standard_normal((m, f)).astype(np.float32))
    W = T((rng.standard_normal((9 * v, f)) / 17.0).astype(np.float32))
    nbr = lat.neighbours(lat, 1, False)
    lib = L.load_library()
    st = lat.m_hash_table._storage
    good = st.row_regions.clone()
    parts = [None, good, torch.zeros(16, dtype=torch.int32, device=dev()), torch.arange(16, 0, -1, dtype=torch.int32, device=dev()) * 9000,
             T(rng.integers(-10 ** 9, 10 ** 9, 16).astype(np.int32))]
    ref = None
    for part in parts:
        out = torch.empty((m, f), dtype=torch.float32, device=dev())
        gv = torch.empty((m, v), dtype=torch.float32, device=dev())
        gw = torch.empty((9 * v, f), dtype=torch.float32, device=dev())
        ws = torch.empty((int(lib.ln_conv_grad_filter_workspace_bytes(m, 9, v, f)) + 4096,), dtype=torch.uint8, device=dev())
        lib.ln_conv_row_partition(_lib.ptr(part))
        try:
            _lib.check(lib.ln_conv_forward(_lib.ptr(nbr), _lib.ptr(values), _lib.ptr(W), m, 9, v, f, 0, _lib.ptr(out), _lib.stream_ptr(dev())))
            _lib.check(lib.ln_conv_backward(_lib.ptr(nbr), _lib.ptr(nbr), _lib.ptr(values), _lib.ptr(G), _lib.ptr(W), m, m, 9, v, f, _lib.ptr(gv),
                                            _lib.ptr(gw), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev())))
        finally:
            lib.ln_conv_row_partition(None)
        torch.cuda.synchronize()
        got = (N(out), N(gv), N(gw))
        if ref is None:
            ref = got
        else:
            assert np.array_equal(ref[0].view(np.uint32), got[0].view(np.uint32))
            assert np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32))
            # (the filter gradient sums one slab per workgroup: the slab ORDER follows the tile map, so it agrees to rounding only)
            np.testing.assert_allclose(got[2], ref[2], rtol=1e-5, atol=1e-5 * float(np.abs(ref[2]).max()))


def test_slot_order_switch_restores_hashed_slots():
    """set_slot_order("hash"): the planes steer the segment regions only (round 2-5 behaviour) — same lattice, hashed slots."""
    from lattice_net_amd import lattice as LT, synthetic
    LT.set_slot_order("hash")
    pos_np, sigma, cap = synthetic.lidar_cloud(20000, 17), 0.9, 40000
    lat, idx, w, _ = build_space_ordered(pos_np, sigma, cap)
    st = lat.m_hash_table._storage
    assert st.slot_map is None and not st.rows_follow_space and st.planes is not None
    m = lat.nr_lattice_vertices()
    t = O.OracleHashTable(cap, 3)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((3,), sigma, np.float32)))
    perm = row_permutation(N(lat.m_hash_table.m_keys_tensor)[:m], t.keys[:m])
    assert np.array_equal(perm[N(idx).astype(np.int64)], oidx)


@pytest.mark.parametrize("order", ["slot", "canonical"])
def test_batch_of_clouds_in_one_table_gives_every_clouds_own_lattice(order):
    """Lattice.set_cloud_batch: B small clouds in ONE table, each cloud's lattice translated along the first key coordinate.  Per cloud:
    the vertex set is the oracle's (moved by the cloud's key offset), splat indices equal through the key matching, weights bit for
    bit; the whole chain splat -> conv -> slice forward + backward over the batch gives, cloud by cloud, what a lattice of that cloud
    alone gives (filter gradient = the sum over the clouds); the coarse level and the level-crossing neighbour lists stay inside a cloud."""
    import lattice_net_amd as L
    from lattice_net_amd import lattice as LT, synthetic
    LT.set_row_order(order)
    LT.set_slot_order("hash")
    B, n0, v, f, sigma, cap = 5, 2500, 32, 32, 0.05, 200000
    rng = np.random.default_rng(11)
    clouds = [synthetic.box_surface_cloud(n0, 100 + b) for b in range(B)]
    pos_np = np.ascontiguousarray(np.concatenate(clouds, 0))
    vals_np = rng.standard_normal((B * n0, v)).astype(np.float32)
    g_np = rng.standard_normal((B * n0, f)).astype(np.float32)
    w_np = (rng.standard_normal((9 * v, f)) / np.sqrt(9 * v)).astype(np.float32)
    lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev())
    lat.set_cloud_batch(n0)
    step = lat.m_hash_table._batch[1]
    W = T(w_np).requires_grad_(True)
    lv, _, idx, w = L.SplatLattice.apply(lat, T(pos_np), T(vals_np))
    m = lat.nr_lattice_vertices()
    lvm = lv[:m].contiguous().requires_grad_(True)
    cv, cw = L.ConvIm2RowLattice.apply(lvm, lat, W, 1)
    out = L.SliceLattice.apply(cv, cw.lattice, T(pos_np), idx, w)
    out.backward(T(g_np))
    torch.cuda.synchronize()
    keys = N(lat.m_hash_table.m_keys_tensor)[:m].astype(np.int64)
    gi = N(idx).astype(np.int64).reshape(B, n0 * 4)
    sig = np.full((3,), sigma, np.float32)
    m_sum, gw_sum = 0, np.zeros((9 * v, f), np.float64)

    def close(a, b):
        np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=RTOL, atol=RTOL * float(np.max(np.abs(b))))

    for b in range(B):
        t = O.OracleHashTable(cap, 3)
        oidx, ow = O.build_splat(t, O.scale_positions(clouds[b], sig))
        mb = t.nr_filled
        m_sum += mb
        okeys = t.keys[:mb].astype(np.int64).copy()
        okeys[:, 0] += b * step
        look = {tuple(k): r for r, k in enumerate(keys)}
        rows_of = np.array([look[tuple(k)] for k in okeys])  # raises KeyError if a vertex of the cloud is missing from the batch table
        assert np.array_equal(rows_of[oidx], gi[b])
        assert np.array_equal(N(w).reshape(B, n0 * 4)[b], ow)
        # the chain of the cloud ALONE (own lattice), against the batch's slice of it
        Wb = T(w_np).requires_grad_(True)
        lb = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev())
        pb, vb = T(clouds[b]), T(vals_np[b * n0:(b + 1) * n0])
        lvb, _, idxb, wb = L.SplatLattice.apply(lb, pb, vb)
        mbg = lb.nr_lattice_vertices()
        assert mbg == mb
        lvbm = lvb[:mb].contiguous().requires_grad_(True)
        cvb, cwb = L.ConvIm2RowLattice.apply(lvbm, lb, Wb, 1)
        outb = L.SliceLattice.apply(cvb, cwb.lattice, pb, idxb, wb)
        outb.backward(T(g_np[b * n0:(b + 1) * n0]))
        close(N(out)[b * n0:(b + 1) * n0], N(outb))
        gw_sum += N(Wb.grad).astype(np.float64)
    assert m == m_sum, "the clouds of a batch must not share a vertex"
    close(N(W.grad), gw_sum)
    # coarse level from the positions (create_coarse_verts_naive) and both level-crossing neighbour lists: every neighbour a row of the same cloud
    lat.set_values(lv[:m].contiguous())
    coarse = lat.create_coarse_verts_naive(T(pos_np))
    mc = coarse.nr_lattice_vertices()
    ck = N(coarse.m_hash_table.m_keys_tensor)[:mc].astype(np.int64)
    cstep = coarse.m_hash_table._batch[1]
    assert cstep * 2 == step
    cloud_f = np.floor_divide(keys[:, 0] + step // 2, step)
    cloud_c = np.floor_divide(ck[:, 0] + cstep // 2, cstep)
    assert set(cloud_f) == set(range(B)) and set(cloud_c) == set(range(B))
    nb_cf = N(coarse.neighbours(lat, 1, False)).astype(np.int64)
    ok = nb_cf >= 0
    assert ok.sum() > mc and np.array_equal(cloud_f[nb_cf[ok]], np.broadcast_to(cloud_c[:, None], nb_cf.shape)[ok])
    nb_fc = N(lat.neighbours(coarse, 1, False)).astype(np.int64)
    ok = nb_fc >= 0
    assert ok.sum() > m // 2 and np.array_equal(cloud_c[nb_fc[ok]], np.broadcast_to(cloud_f[:, None], nb_fc.shape)[ok])
    mc_sum = 0
    for b in range(B):
        tc = O.OracleHashTable(cap, 3)
        O.build_splat(tc, O.scale_positions(clouds[b], 2 * sig), write=False)
        mc_sum += tc.nr_filled
    assert mc == mc_sum
