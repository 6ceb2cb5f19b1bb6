"""Deterministic mode of the backend (lattice.set_deterministic, round 6; `pytest -m gpu`).

The default kernels fix WHAT is summed onto a vertex, not the ORDER: a token's place in its vertex's list comes from an atomic counter
of the build (LDS in the bucket pass, global on the atomic path and in ln_csr_build), and rows with several segments combine through
float atomics.  In deterministic mode every build sorts the token lists (LN_BUILD_SORTED_CSR / ln_csr_sort) and every segment reduce
walks a row with one lane group in list order (LnCsr.dense & 2): results are bitwise identical run to run — and still the oracle's."""
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


@pytest.fixture(autouse=True)
def deterministic():
    from lattice_net_amd import lattice as LT
    prev = LT.set_deterministic(True)
    yield
    LT.set_deterministic(prev)


def bits(t):
    return N(t).view(np.uint32).copy()


@pytest.mark.parametrize("path", ["bucketed", "atomic"])
@pytest.mark.parametrize("v", [32, 96, 5])
def test_splat_and_slice_gradient_are_bitwise_reproducible_and_match_the_oracle(path, v):
    import lattice_net_amd as L
    from lattice_net_amd import lattice as LT, synthetic
    n, sigma, cap = 60000, 0.9, 100000
    rng = np.random.default_rng(v)
    pos_np = np.concatenate([synthetic.lidar_cloud(n - 3000, 4), np.tile(np.array([[0.31, -0.17, 0.05]], np.float32), (3000, 1))], 0)  # + four hot vertices
    vals_np = rng.standard_normal((n, v)).astype(np.float32)
    g_np = rng.standard_normal((n, v)).astype(np.float32)
    pos, vals, G = T(pos_np), T(vals_np), T(g_np)
    old = LT._FORCE_ATOMIC_BUILD
    LT._FORCE_ATOMIC_BUILD = path == "atomic"
    try:
        seen = []
        for _ in range(3):
            lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev())
            lv, _, idx, w = L.SplatLattice.apply(lat, pos, vals)
            m = lat.nr_lattice_vertices()
            lvm = lv[:m].clone().requires_grad_(True)
            out = L.SliceLattice.apply(lvm, lat, pos, idx, w)
            out.backward(G)
            torch.cuda.synchronize()
            st = lat.m_hash_table._storage
            hit = st.csr_cache[(idx.data_ptr(), idx._version, idx.numel())]
            seen.append((bits(lv[:m]), bits(lvm.grad), N(idx).copy(), N(hit[0]).copy()))
    finally:
        LT._FORCE_ATOMIC_BUILD = old
    for other in seen[1:]:
        assert np.array_equal(seen[0][0], other[0]) and np.array_equal(seen[0][1], other[1]) and np.array_equal(seen[0][2], other[2])
    t = O.OracleHashTable(cap, 3)
    oidx, ow = O.build_splat(t, O.scale_positions(pos_np, np.full((3,), sigma, np.float32)))
    assert m == t.nr_filled and np.array_equal(seen[0][2], oidx)  # (tests/conftest.py: canonical numbering)
    ov = np.zeros((m, v), np.float64)
    np.add.at(ov, oidx, (vals_np.astype(np.float64)[:, None, :] * ow.reshape(n, 4, 1).astype(np.float64)).reshape(n * 4, v))
    oabs = np.zeros((m, v), np.float64)
    np.add.at(oabs, oidx, (np.abs(vals_np).astype(np.float64)[:, None, :] * np.abs(ow).reshape(n, 4, 1)).reshape(n * 4, v))
    got = seen[0][0].view(np.float32).astype(np.float64)
    assert np.all(np.abs(got - ov) <= 1e-5 * oabs + 1e-30)
    gb = O.slice_backwards(g_np, oidx, ow, m).astype(np.float64)
    gabs = O.slice_backwards(np.abs(g_np), oidx, np.abs(ow), m).astype(np.float64)
    assert np.all(np.abs(seen[0][1].view(np.float32).astype(np.float64) - gb) <= 1e-5 * gabs + 1e-30)


def test_token_lists_are_sorted_after_a_deterministic_build():
    """The build's slot CSR and the CSR ln_csr_build derives from an index tensor: every group's tokens ascend."""
    import lattice_net_amd as L
    from lattice_net_amd import synthetic
    n, sigma, cap = 30000, 0.9, 60000
    pos = T(synthetic.lidar_cloud(n, 6))
    lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev())
    lat.begin_splat()
    idx, w = lat.just_create_verts(pos, True)
    m = lat.nr_lattice_vertices()
    st = lat.m_hash_table._storage
    hashed = st.hashed()

    def check(buf, groups, tokens):
        a = N(buf)
        max_seg = (a.size - (groups + 1) - max(tokens, 1) - 10) // 32
        o1 = 32 * max_seg
        grp_start = a[o1:o1 + groups + 1]
        csr_tok = a[o1 + groups + 1:o1 + groups + 1 + tokens]
        assert grp_start[0] == 0 and grp_start[-1] == tokens and np.all(np.diff(grp_start) >= 0)
        nonempty = 0
        for g in np.nonzero(np.diff(grp_start) > 1)[0]:
            seg = csr_tok[grp_start[g]:grp_start[g + 1]]
            assert np.all(np.diff(seg) > 0), (g, seg[:8])
            nonempty += 1
        assert nonempty > 1000
        return csr_tok

    hit = st.csr_cache[(idx.data_ptr(), idx._version, idx.numel())]
    toks = check(hit[0], hashed, idx.numel())
    assert np.array_equal(np.sort(toks), np.arange(idx.numel()))  # every token exactly once
    idx2 = idx.clone()  # an index tensor the table has no CSR for: ln_csr_build + ln_csr_sort
    entry = lat._csr(idx2)
    toks2 = check(entry[0], lat.m_hash_table.capacity(), idx2.numel())
    assert np.array_equal(np.sort(toks2), np.arange(idx2.numel()))
