"""Default vertex numbering of the bucketed build (hash-slot order) against the oracle (`pytest -m gpu`).

The reference numbers vertices in thread-arrival order (HashTableGPU.cuh:454), so its results are defined up to a
permutation of the rows.  Here: the slot-order build must produce exactly the oracle's vertex SET (keys), and its splat
indices must be the oracle's under the row permutation that matches the keys; everything that does not mention rows
(barycentric weights, sliced outputs, filter gradients) must agree directly; and ln_canonicalize must turn the table
into the oracle's numbering bit for bit."""
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
    yield
    lat.set_row_order(prev)


def row_permutation(keys_gpu, keys_oracle):
    """perm[gpu row] = oracle row; asserts that both hold the same set of keys."""
    assert keys_gpu.shape == keys_oracle.shape
    og = np.lexsort(keys_gpu.T[::-1])
    oo = np.lexsort(keys_oracle.T[::-1])
    assert np.array_equal(keys_gpu[og], keys_oracle[oo]), "vertex sets differ"
    perm = np.empty(len(og), np.int64)
    perm[og] = oo
    return perm


def clouds():
    from lattice_net_amd import synthetic
    rng = np.random.default_rng(5)
    yield "cube1k", rng.uniform(-1, 1, (1000, 3)).astype(np.float32), 0.2, 60000
    yield "lidar20k", synthetic.lidar_cloud(20000, 3), 0.9, 30000
    dup = np.repeat(rng.uniform(-2, 2, (50, 3)).astype(np.float32), 40, axis=0)  # 40 copies of 50 points: hot vertices
    yield "duplicates", dup, 0.5, 5000
    yield "lidar120k", synthetic.lidar_cloud(120000, 0), 0.9, 100000


@pytest.mark.parametrize("case", list(clouds()), ids=lambda c: c[0])
def test_slot_order_build_equals_oracle_up_to_row_permutation(case):
    from lattice_net_amd import Lattice
    _, pos_np, sigma, cap = case
    lat = Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev())
    lat.begin_splat()
    idx, w = lat.just_create_verts(T(pos_np), True)
    m = lat.nr_lattice_vertices()
    t = O.OracleHashTable(cap, 3)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((3,), sigma, np.float32)))
    assert m == t.nr_filled
    ht = lat.m_hash_table
    keys = N(ht.m_keys_tensor)
    assert not keys[m:].any(), "key rows beyond the vertex count must stay zero (HashTable::clear)"
    perm = row_permutation(keys[:m], t.keys[:m])
    gi = N(idx).astype(np.int64)
    assert gi.min() >= 0 and gi.max() == m - 1
    assert np.array_equal(perm[gi], oidx)
    assert np.array_equal(N(w), ow)  # weights do not depend on the numbering: bit-exact
    ent = N(ht.m_entries_tensor)
    assert np.array_equal(np.sort(ent[ent >= 0]), np.arange(m)), "entries must hold every row exactly once"
    # bucket-major numbering: the rows of a bucket (a run of consecutive slots) form one contiguous range, buckets in ascending order
    rows_in_slot_order = ent[ent >= 0]
    running_max = np.maximum.accumulate(rows_in_slot_order)
    assert np.all(running_max - rows_in_slot_order < 512), "rows must ascend bucket by bucket"


def test_default_numbering_is_identical_run_to_run():
    """Rows inside a bucket are ranked by their smallest token, not by the slot the LDS CAS race left them in: two builds of one
    cloud give identical indices, keys and neighbour lists."""
    from lattice_net_amd import Lattice, synthetic
    pos = T(synthetic.lidar_cloud(120000, 3))
    seen = []
    for _ in range(4):
        lat = Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev())
        lat.begin_splat()
        idx, w = lat.just_create_verts(pos, True)
        m = lat.nr_lattice_vertices()
        seen.append((N(idx).copy(), N(lat.m_hash_table.m_keys_tensor[:m]).copy(), N(lat.neighbours(lat, 1, False)).copy()))
    for other in seen[1:]:
        for a, b in zip(seen[0], other):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("case", list(clouds())[:3], ids=lambda c: c[0])
def test_canonicalize_after_slot_order_build_is_bit_exact(case):
    """ln_canonicalize called through the C ABI on a slot-order table: keys, entries and idx become the oracle's."""
    import lattice_net_amd as L
    from lattice_net_amd import _lib, lattice as LT
    _, pos_np, sigma, cap = case
    lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev())
    lat.begin_splat()
    idx, w = lat.just_create_verts(T(pos_np), True)
    m = lat.nr_lattice_vertices()
    lib = L.load_library()
    tokens = idx.numel()
    ws = torch.empty((LT._build_sizes(tokens, cap)[0],), dtype=torch.uint8, device=dev())
    t = lat.m_hash_table.c_table()
    _lib.check(lib.ln_canonicalize(C.byref(t), _lib.ptr(idx), tokens, None, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev())), "ln_canonicalize")
    lat.m_hash_table._storage.touch()  # csr = NULL: the build's cached CSR still names the old rows and must not be used again
    torch.cuda.synchronize()
    to = O.OracleHashTable(cap, 3)
    oidx, ow = O.build_splat(to, O.scale_positions(pos_np, np.full((3,), sigma, np.float32)))
    assert m == to.nr_filled
    assert np.array_equal(N(idx), oidx)
    assert np.array_equal(N(lat.m_hash_table.m_keys_tensor)[:m], to.keys[:m])
    ent = N(lat.m_hash_table.m_entries_tensor)
    assert np.array_equal(np.sort(ent[ent >= 0]), np.arange(m))


def test_slot_order_chain_matches_oracle():
    """splat -> conv -> slice forward + backward under the default numbering: the sliced output and the filter gradient do
    not mention rows and must equal the oracle's; lattice values and their gradient are compared through the permutation."""
    import lattice_net_amd as L
    from lattice_net_amd import synthetic
    n, v, f, sigma, cap = 6000, 32, 32, 0.9, 20000
    pos_np = synthetic.lidar_cloud(n, 1)
    rng = np.random.default_rng(1)
    vals_np = rng.standard_normal((n, v)).astype(np.float32)
    w_np = (rng.standard_normal((9 * v, f)) / np.sqrt(9 * v)).astype(np.float32)
    g_np = rng.standard_normal((n, f)).astype(np.float32)
    pos, vals = T(pos_np), T(vals_np)
    W = T(w_np).requires_grad_(True)
    lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev())
    lv, wrap, idx, w = L.SplatLattice.apply(lat, pos, vals)
    m = lat.nr_lattice_vertices()
    lv = lv[:m].contiguous().requires_grad_(True)
    cv, cwrap = L.ConvIm2RowLattice.apply(lv, lat, W, 1)
    out = L.SliceLattice.apply(cv, cwrap.lattice, pos, idx, w)
    out.backward(T(g_np))
    torch.cuda.synchronize()

    t = O.OracleHashTable(cap, 3)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((3,), sigma, np.float32)))
    assert t.nr_filled == m
    perm = row_permutation(N(lat.m_hash_table.m_keys_tensor)[:m], t.keys[:m])
    ov = np.zeros((m, v), np.float32)
    O.splat_accumulate(ov, vals_np, oidx, ow)
    nbr = O.neighbour_rows(t.keys[:m], t, 1, 1, 1, False)
    oc = O.conv_forward(nbr, ov, w_np)
    oo = O.slice_with_precomputation(oc, oidx, ow, n)

    def close(a, b):
        np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=RTOL, atol=RTOL * float(np.max(np.abs(b))))

    close(N(lv)[np.argsort(perm)], ov)  # gpu row r is oracle row perm[r]
    # the neighbour list, relabelled, is the oracle's (bit-exact)
    gn = N(lat.neighbours(lat, 1, False)).astype(np.int64)
    gn_as_oracle = np.where(gn >= 0, perm[np.maximum(gn, 0)], gn)[np.argsort(perm)]
    assert np.array_equal(gn_as_oracle, nbr)
    close(N(out), oo)
    g_c = O.slice_backwards(g_np, oidx, ow, m)
    rows = O.im2row(nbr, ov).astype(np.float64)
    close(N(W.grad), rows.T @ g_c.astype(np.float64))
    wb = w_np.reshape(9, v, f)
    gv = np.zeros((m, v), np.float64)  # adjoint of the gather-GEMM, fp64
    for e in range(9):
        ok = nbr[:, e] >= 0
        np.add.at(gv, nbr[ok, e], g_c[ok].astype(np.float64) @ wb[e].T.astype(np.float64))
    close(N(lv.grad)[np.argsort(perm)], gv)


def _splat_values_case(pos_np, vals_np, sigma, cap, half=False):
    import lattice_net_amd as L
    n, v = vals_np.shape
    lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev())
    vals = T(vals_np)
    if half:
        vals = vals.half()
        vals_np = N(vals).astype(np.float32)
    lv, wrap, idx, w = L.SplatLattice.apply(lat, T(pos_np), vals)
    m = lat.nr_lattice_vertices()
    t = O.OracleHashTable(cap, 3)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((3,), sigma, np.float32)))
    assert m == t.nr_filled
    perm = row_permutation(N(lat.m_hash_table.m_keys_tensor)[:m], t.keys[:m])
    ov = np.zeros((m, v), np.float64)
    np.add.at(ov, oidx[oidx >= 0], (vals_np.astype(np.float64)[:, None, :] * ow.reshape(n, 4, 1).astype(np.float64)).reshape(n * 4, v)[oidx >= 0])
    got = N(lv).astype(np.float64)
    assert not got[m:].any(), "rows beyond the vertex count must stay zero"
    np.testing.assert_allclose(got[:m][np.argsort(perm)], ov, rtol=RTOL, atol=RTOL * float(np.abs(ov).max()))
    return lat


@pytest.mark.parametrize("v", [4, 32, 64, 6, 20])
def test_splat_values_match_oracle(v, golden):
    """Splat values under the slot-order numbering (float4 and scalar lanes of the segment reduce) against an fp64 evaluation
    of splatCacheNaive (LatticeGPU.cuh:937-971) on the oracle's indices, rows matched through the keys."""
    g = golden("F9_lidar")
    pos_np = g["pos_raw"]
    rng = np.random.default_rng(v)
    _splat_values_case(pos_np, rng.standard_normal((pos_np.shape[0], v)).astype(np.float32), float(g["sigma"]), int(g["capacity"]))


def test_splat_values_match_golden(golden):
    g = golden("F1_config1")
    lat = _splat_values_case(g["pos_raw"], g["vals"], float(g["sigma"]), int(g["capacity"]))
    m = int(g["nr_filled"])
    perm = row_permutation(N(lat.m_hash_table.m_keys_tensor)[:m], g["keys"])
    got = N(lat.values())[:m][np.argsort(perm)]  # the reference kernel's own accumulate (serial run), rows matched through the keys
    np.testing.assert_allclose(got, g["values"], rtol=RTOL, atol=RTOL * float(np.abs(g["values"]).max()))


def test_splat_values_fp16_rows():
    from lattice_net_amd import synthetic
    rng = np.random.default_rng(2)
    pos_np = synthetic.lidar_cloud(30000, 2)
    _splat_values_case(pos_np, rng.standard_normal((30000, 32)).astype(np.float32), 0.9, 60000, half=True)


@pytest.mark.parametrize("n_same", [500, 6000])
def test_splat_values_hot_vertices(n_same):
    """Thousands of tokens on four vertices: multi-segment runs (shuffle combine + atomics), and at 6000 identical points
    buckets that hold more tokens than the bucket pass keeps in registers."""
    rng = np.random.default_rng(3)
    pos_np = np.concatenate([np.tile(np.array([[0.31, -0.17, 0.05]], np.float32), (n_same, 1)),
                             rng.uniform(-3, 3, (800, 3)).astype(np.float32)], 0)
    vals_np = rng.uniform(0.5, 1.5, (pos_np.shape[0], 8)).astype(np.float32)  # same sign: no cancellation in the hot sums
    _splat_values_case(pos_np, vals_np, 0.5, 5000)


def test_splat_values_full_size_c3():
    from lattice_net_amd import synthetic
    rng = np.random.default_rng(4)
    pos_np = synthetic.lidar_cloud(120000, 0)
    lat = _splat_values_case(pos_np, rng.standard_normal((120000, 32)).astype(np.float32), 0.9, 100000)
    assert lat.nr_lattice_vertices() == 46538


def test_slot_order_two_levels_distribute_and_heads_match_oracle():
    """The shipped default numbering on the rest of the operator surface (the golden / parity suite runs under the canonical
    relabelling): distribute, the coarse vertex set, both level-crossing neighbour lists with and without flip, the coarsening
    convolution forward + backward, gather and slice_classify — rows matched through the keys, everything that does not mention rows
    bit for bit (distributed rows, weights, gathered rows, logits)."""
    import lattice_net_amd as L
    from lattice_net_amd import synthetic
    from lattice_net_amd.lattice_funcs import CoarsenLattice, GatherLattice, SliceClassifyLattice
    n, sigma, cap, v, f, c = 9000, 0.9, 30000, 32, 64, 7
    pos_np = synthetic.lidar_cloud(n, 5)
    rng = np.random.default_rng(6)
    pos = T(pos_np)
    lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev())
    lat.begin_splat()
    fine, rows, idx, w = lat.distribute(pos, T(np.zeros((n, 1), np.float32)))
    m = fine.nr_lattice_vertices()
    sig = np.full((3,), sigma, np.float32)
    tf = O.OracleHashTable(cap, 3)
    orow, oidx, ow = O.distribute(tf, O.scale_positions(pos_np, sig), np.zeros((n, 1), np.float32))
    assert m == tf.nr_filled
    pf = row_permutation(N(fine.m_hash_table.m_keys_tensor)[:m], tf.keys[:m])
    assert np.array_equal(pf[N(idx).astype(np.int64)], oidx)
    assert np.array_equal(N(w), ow) and np.array_equal(N(rows), orow)
    coarse = fine.create_coarse_verts_naive(pos)
    mc = coarse.nr_lattice_vertices()
    tc = O.OracleHashTable(cap, 3)
    O.build_splat(tc, O.scale_positions(pos_np, 2 * sig), write=False)
    assert mc == tc.nr_filled
    pc = row_permutation(N(coarse.m_hash_table.m_keys_tensor)[:mc], tc.keys[:mc])

    def as_oracle(gn, perm_q, perm_nb):  # a gpu neighbour list in the oracle's numbering of both lattices
        gn = gn.astype(np.int64)
        return np.where(gn >= 0, perm_nb[np.maximum(gn, 0)], gn)[np.argsort(perm_q)]
    nbr_cf = {}
    for flip in (False, True):
        nbr_cf[flip] = O.neighbour_rows(tc.keys[:mc], tf, 2, 1, 1, flip)
        assert np.array_equal(as_oracle(N(coarse.neighbours(fine, 1, flip)), pc, pf), nbr_cf[flip])
        assert np.array_equal(as_oracle(N(fine.neighbours(coarse, 1, flip)), pf, pc), O.neighbour_rows(tf.keys[:m], tc, 1, 2, 1, flip))

    def close(a, b):
        np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=RTOL, atol=RTOL * float(np.max(np.abs(b))))
    # coarsening convolution (coarse query x fine table), forward + both gradients in fp64 on the oracle's list
    lv_o = rng.standard_normal((m, v)).astype(np.float32)          # in the ORACLE's row order
    fb_np = (rng.standard_normal((9 * v, f)) / np.sqrt(9 * v)).astype(np.float32)
    g_o = rng.standard_normal((mc, f)).astype(np.float32)
    lv = T(lv_o[pf]).requires_grad_(True)                          # gpu row r holds oracle row pf[r]
    fb = T(fb_np).requires_grad_(True)
    y, _ = CoarsenLattice.apply(lv, fine, fb, coarse)
    y.backward(T(g_o[pc]))
    rowsf = O.im2row(nbr_cf[False], lv_o).astype(np.float64)
    close(N(y)[np.argsort(pc)], rowsf @ fb_np.astype(np.float64))
    close(N(fb.grad), rowsf.T @ g_o.astype(np.float64))
    gv = np.zeros((m, v), np.float64)
    wb = fb_np.reshape(9, v, f).astype(np.float64)
    for e in range(9):
        ok = nbr_cf[False][:, e] >= 0
        np.add.at(gv, nbr_cf[False][ok, e], g_o[ok].astype(np.float64) @ wb[e].T)
    close(N(lv.grad)[np.argsort(pf)], gv)
    # gather and slice_classify on the fine lattice
    v8 = rng.standard_normal((m, 8)).astype(np.float32)
    ga = GatherLattice.apply(T(v8[pf]), fine, pos, idx, w)
    assert np.array_equal(N(ga), O.gather_with_precomputation(v8, oidx, ow, n))
    vals_o = rng.standard_normal((m, 64)).astype(np.float32)
    dw_np = (0.1 * rng.standard_normal((n, 4))).astype(np.float32)
    lw_np, lb_np = rng.standard_normal((c, 64)).astype(np.float32), rng.standard_normal((c,)).astype(np.float32)
    gl_np = rng.standard_normal((n, c)).astype(np.float32)
    vals, dw, lw, lb = (T(x).requires_grad_(True) for x in (vals_o[pf], dw_np, lw_np, lb_np))
    logits = SliceClassifyLattice.apply(vals, fine, pos, dw, lw, lb, c, idx, w)
    assert np.array_equal(N(logits), O.slice_classify(vals_o, dw_np, lw_np, lb_np, oidx, ow, n))
    logits.backward(T(gl_np))
    ogv, ogd, ogw, ogb = O.slice_classify_backwards(gl_np, vals_o, dw_np, lw_np, lb_np, oidx, ow, n)
    close(N(vals.grad)[np.argsort(pf)], ogv)
    close(N(dw.grad), ogd)
    close(N(lw.grad), ogw)
    close(N(lb.grad), ogb)


def test_canonicalize_rows_keeps_the_cached_csr_consistent():
    """Lattice.canonicalize_rows relabels the build's cached CSR together with the table: a scatter through it afterwards lands on
    the new rows (splat values equal the oracle's in the oracle's own numbering)."""
    import lattice_net_amd as L
    from lattice_net_amd import synthetic
    n, v, sigma, cap = 8000, 32, 0.9, 20000
    pos_np = synthetic.lidar_cloud(n, 7)
    vals_np = np.random.default_rng(7).standard_normal((n, v)).astype(np.float32)
    lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev())
    lat.begin_splat()
    idx, w = lat.just_create_verts(T(pos_np), True)
    m = lat.nr_lattice_vertices()
    lat.canonicalize_rows(idx)
    t = O.OracleHashTable(cap, 3)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((3,), sigma, np.float32)))
    assert np.array_equal(N(idx), oidx) and np.array_equal(N(lat.m_hash_table.m_keys_tensor)[:m], t.keys[:m])
    out = torch.zeros((m, v), device=dev())
    lat._scatter_rows(T(vals_np), idx, w, out, v, 4, v)  # through the cached (relabelled) CSR
    ov = np.zeros((m, v), np.float32)
    O.splat_accumulate(ov, vals_np, oidx, ow)
    np.testing.assert_allclose(N(out), ov, rtol=RTOL, atol=RTOL * float(np.abs(ov).max()))
    assert np.array_equal(N(lat.neighbours(lat, 1, False)), O.neighbour_rows(t.keys[:m], t, 1, 1, 1, False))
