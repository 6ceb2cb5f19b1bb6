"""GPU tests of the operator surface beyond the raw kernels: autograd Functions, nn.Modules, Lattice
object semantics, and the remaining BASELINE.json configs (C2 ShapeNet-like U-Net level chain, C4
ScanNet-like 200k-point scene with a 5M-slot table) — all against the CPU oracle."""
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


def make_lattice(sigma, capacity, d=3):
    from lattice_net_amd import Lattice
    return Lattice(sigmas=[float(sigma)] * d, capacity=int(capacity), device=dev())


def close(a, b, scale=None, rtol=RTOL):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    s = float(np.max(np.abs(b))) if scale is None else scale
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * max(s, 1e-30))


def oracle_table(pos_np, sigma, cap):
    t = O.OracleHashTable(cap, pos_np.shape[1])
    idx, w = O.build_splat(t, O.scale_positions(pos_np, np.full((pos_np.shape[1],), sigma, np.float32)))
    return t, idx, w


def test_slice_and_gather_autograd_against_oracle():
    from lattice_net_amd import GatherLattice, SliceLattice
    from lattice_net_amd.synthetic import cube_cloud
    pos_np = cube_cloud(2500, 11)
    lat = make_lattice(0.2, 40000)
    lat.begin_splat()
    idx, w = lat.just_create_verts(T(pos_np), True)
    m = lat.nr_lattice_vertices()
    t, oidx, ow = oracle_table(pos_np, 0.2, 40000)
    rng = np.random.default_rng(0)
    v = 8
    vals_np = rng.standard_normal((m, v)).astype(np.float32)
    g_np = rng.standard_normal((2500, v)).astype(np.float32)
    vals = T(vals_np).requires_grad_(True)
    out = SliceLattice.apply(vals, lat, T(pos_np), idx, w)
    np.testing.assert_array_equal(N(out), O.slice_with_precomputation(vals_np, oidx, ow, 2500))
    out.backward(T(g_np))
    close(N(vals.grad), O.slice_backwards(g_np, oidx, ow, m))
    # slicing without precomputed indices (new positions): backward builds the CSR from the returned indices
    q_np = cube_cloud(700, 12, -1.05, 1.05)
    vals2 = T(vals_np).requires_grad_(True)
    out2 = SliceLattice.apply(vals2, lat, T(q_np))
    o2, i2, w2 = O.slice_no_precomputation(t, vals_np, O.scale_positions(q_np, np.full((3,), 0.2, np.float32)))
    np.testing.assert_array_equal(N(out2), o2)
    g2 = rng.standard_normal((700, v)).astype(np.float32)
    out2.backward(T(g2))
    close(N(vals2.grad), O.slice_backwards(g2, i2, np.where(i2 >= 0, w2, 0).astype(np.float32), m))
    # gather
    vals3 = T(vals_np).requires_grad_(True)
    ga = GatherLattice.apply(vals3, lat, T(pos_np), idx, w)
    np.testing.assert_array_equal(N(ga), O.gather_with_precomputation(vals_np, oidx, ow, 2500))
    gg = rng.standard_normal(tuple(ga.shape)).astype(np.float32)
    ga.backward(T(gg))
    close(N(vals3.grad), O.gather_backwards(gg, oidx, ow, m, 3))


@pytest.mark.parametrize("n,v,c", [(1500, 8, 20), (5000, 96, 20), (777, 128, 13), (300, 5, 3), (2500, 128, 50), (4000, 64, 20), (3111, 32, 32), (129, 64, 1),
                                   (2049, 160, 24)])
def test_slice_classify_autograd_against_oracle(n, v, c):
    """(5000, 96, 20) is the head of the SemanticKITTI LNN.  V % 32 == 0 with up to 32 classes runs the wave-tiled kernels of
    ln_classify.hip (forward: 64-point tiles, all accumulator widths 8 / 16 / 24 / 32; backward up to V = 128, so (2049, 160, 24)
    pairs the wave-tiled forward with the general backward); the other shapes take the general kernels of ln_rows.hip; ragged
    last tiles everywhere."""
    from lattice_net_amd import SliceClassifyLattice
    from lattice_net_amd.synthetic import cube_cloud
    pos_np = cube_cloud(n, 21)
    lat = make_lattice(0.25, 30000)
    lat.begin_splat()
    idx, w = lat.just_create_verts(T(pos_np), True)
    m = lat.nr_lattice_vertices()
    _, oidx, ow = oracle_table(pos_np, 0.25, 30000)
    rng = np.random.default_rng(2)
    vals_np = rng.standard_normal((m, v)).astype(np.float32)
    dw_np = (0.1 * rng.standard_normal((n, 4))).astype(np.float32)
    lw_np = rng.standard_normal((c, v)).astype(np.float32)
    lb_np = rng.standard_normal((c,)).astype(np.float32)
    gl_np = rng.standard_normal((n, c)).astype(np.float32)
    vals, dw, lw, lb = (T(x).requires_grad_(True) for x in (vals_np, dw_np, lw_np, lb_np))
    logits = SliceClassifyLattice.apply(vals, lat, T(pos_np), dw, lw, lb, c, idx, w)
    np.testing.assert_array_equal(N(logits), O.slice_classify(vals_np, dw_np, lw_np, lb_np, oidx, ow, n))
    logits.backward(T(gl_np))
    gv, gd, gw, gb = O.slice_classify_backwards(gl_np, vals_np, dw_np, lw_np, lb_np, oidx, ow, n)
    close(N(vals.grad), gv)
    close(N(dw.grad), gd)
    close(N(lw.grad), gw)
    close(N(lb.grad), gb)
    # the C entry point can also scatter the lattice-value gradient itself (callers without a CSR adjacency)
    import ctypes as C
    from lattice_net_amd import _lib
    lib = _lib.load()
    g_vals = torch.zeros((m, v), device=dev())
    g_dw, g_lw, g_lb = torch.zeros((n, 4), device=dev()), torch.zeros((c, v), device=dev()), torch.zeros((c,), device=dev())
    gs, weff = torch.empty((n, v), device=dev()), torch.empty((n * 4,), device=dev())
    ws = torch.empty((lib.ln_slice_classify_backward_workspace_bytes(n, 3, v, c),), dtype=torch.uint8, device=dev())
    rc = lib.ln_slice_classify_backward(_lib.ptr(T(gl_np)), _lib.ptr(vals.detach()), _lib.ptr(dw.detach()), _lib.ptr(lw.detach()),
                                        _lib.ptr(idx), _lib.ptr(w), n, 3, v, c, _lib.ptr(g_vals), _lib.ptr(g_dw), _lib.ptr(g_lw),
                                        _lib.ptr(g_lb), _lib.ptr(gs), _lib.ptr(weff), _lib.ptr(ws), ws.numel(),
                                        _lib.stream_ptr(dev()))
    assert rc == 0, lib.ln_last_error_string()
    close(N(g_vals), gv)
    close(N(g_lw), gw)
    np.testing.assert_array_equal(N(weff), ow + dw_np.reshape(-1))


def test_modules_level_chain_c2_shapenet_like():
    """C2: ~2.5k-point surface cloud, sigma 0.05, capacity 60000; distribute -> conv -> coarsen x3 -> finefy -> slice, fwd+bwd."""
    from lattice_net_amd import Lattice
    from lattice_net_amd.lattice_modules import (CoarsenLatticeModule, ConvLatticeIm2RowModule, DistributeLatticeModule,
                                                 FinefyLatticeModule, SliceLatticeModule)
    from lattice_net_amd.synthetic import box_surface_cloud
    torch.manual_seed(0)
    pos_np = box_surface_cloud(2500, 5)
    pos = T(pos_np)
    vals = torch.ones((2500, 1), device=dev())
    lat = Lattice(sigmas=[0.05] * 3, capacity=60000, device=dev())
    dist_lat, distributed, idx, w = DistributeLatticeModule()(lat, pos, vals)
    m0 = dist_lat.nr_lattice_vertices()
    t, oidx, ow = oracle_table(pos_np, 0.05, 60000)
    assert m0 == t.nr_filled
    np.testing.assert_array_equal(N(idx), oidx)
    # post-processing of mods:66-94: per-vertex mean of the scaled positions removed, vertex 0 = invalid bucket
    raw, _, _ = O.distribute(O.OracleHashTable(60000, 3), O.scale_positions(pos_np, np.full((3,), 0.05, np.float32)),
                             np.ones((2500, 1), np.float32))
    sums = np.zeros((m0, 3), np.float64)
    cnt = np.zeros((m0,), np.float64)
    np.add.at(sums, oidx, raw[:, :3].astype(np.float64))
    np.add.at(cnt, oidx, 1.0)
    mean = sums / cnt[:, None]
    mean[0] = 0
    exp = raw.astype(np.float64).copy()
    exp[:, :3] -= mean[oidx]
    exp[oidx == 0] = 0
    close(N(distributed), exp, scale=np.abs(exp).max())
    # feature lift so the convs have something to chew on
    feat = torch.randn((m0, 16), device=dev(), requires_grad=True)
    dist_lat.set_values(feat.detach())
    conv = ConvLatticeIm2RowModule(16, 16)
    lv, ls = conv(feat, dist_lat)
    levels = [(lv, ls)]
    chans = 16
    coarsens = []
    for _ in range(3):
        cm = CoarsenLatticeModule(chans, chans * 2)
        coarsens.append(cm)
        lv, ls = cm(lv, ls)
        chans *= 2
        levels.append((lv, ls))
        assert ls.lvl() == len(levels)
    assert levels[1][1].nr_lattice_vertices() < m0
    for k in range(3, 0, -1):
        fm = FinefyLatticeModule(chans, chans // 2)
        lv, ls = fm(lv, levels[k][1], levels[k - 1][1])
        chans //= 2
        assert lv.shape == (levels[k - 1][1].nr_lattice_vertices(), chans)
    out = SliceLatticeModule()(lv, ls, pos, idx, w)
    assert out.shape == (2500, 16) and torch.isfinite(out).all()
    out.square().mean().backward()
    for mod in [conv] + coarsens:
        assert mod.weight.grad is not None and torch.isfinite(mod.weight.grad).all() and mod.weight.grad.abs().sum() > 0
    assert feat.grad is not None and torch.isfinite(feat.grad).all()
    # coarse level keys against the oracle (sigma doubles per level, L.cu:718-722)
    tc, _, _ = oracle_table(pos_np, 0.1, 60000)
    l1 = levels[1][1]
    np.testing.assert_array_equal(N(l1.hash_table().m_keys_tensor[: l1.nr_lattice_vertices()]), tc.keys[: tc.nr_filled])


def test_c4_scannet_like_scene_large_table():
    """C4: 200k points on planes in an 8x3x8 m box, sigma 0.08, capacity 5,000,000, V=32 (one cloud per GPU)."""
    from lattice_net_amd.synthetic import planes_cloud
    pos_np = planes_cloud(200000, 4)
    lat = make_lattice(0.08, 5000000)
    lat.begin_splat()
    vals_np = np.random.default_rng(4).standard_normal((200000, 4)).astype(np.float32)
    idx, w = lat.splat_standalone(T(pos_np), T(vals_np))
    m = lat.nr_lattice_vertices()
    t, oidx, ow = oracle_table(pos_np, 0.08, 5000000)
    assert m == t.nr_filled
    np.testing.assert_array_equal(N(idx), oidx)
    np.testing.assert_array_equal(N(w), ow)
    nbr = O.neighbour_rows(t.keys[:m], t, 1, 1, 1, False)
    lat.set_values(lat.values()[:m].contiguous())
    np.testing.assert_array_equal(N(lat.neighbours(lat, 1, False)), nbr)
    ov = np.zeros((m, 4), np.float32)
    O.splat_accumulate(ov, vals_np, oidx, ow)
    absv = np.zeros((m, 4), np.float32)
    O.splat_accumulate(absv, np.abs(vals_np), oidx, ow)
    assert np.all(np.abs(N(lat.values()).astype(np.float64) - ov) <= 1e-5 * absv + 1e-30)


def test_lattice_object_semantics():
    from lattice_net_amd.synthetic import cube_cloud
    pos = T(cube_cloud(600, 31))
    lat = make_lattice(0.3, 10000)
    lat.begin_splat()
    lat.splat_standalone(pos, torch.ones((600, 2), device=dev()))
    m = lat.nr_lattice_vertices()
    assert lat.pos_dim() == 3 and lat.val_dim() == 2 and lat.capacity() == 10000 and lat.get_filter_extent(1) == 9
    assert tuple(lat.values().shape) == (10000, 2)            # splat leaves CAP rows (HashTable.cu:32)
    with pytest.raises(ValueError, match="rows"):
        lat.set_values(lat.values())                          # set_values checks rows == nr_lattice_vertices (L.cu:1398)
    lat.set_values(lat.values()[:m].contiguous())
    clone = lat.clone_lattice()                               # structure shared shallowly (L.cu:88-92)
    assert clone.hash_table().m_keys_tensor.data_ptr() == lat.hash_table().m_keys_tensor.data_ptr()
    clone.set_values(torch.zeros((m, 7), device=dev()))
    assert clone.val_dim() == 7 and lat.val_dim() == 2        # val_dim is defined by the values tensor (HashTable.cu:102)
    assert lat.positions() is pos
    coarse = lat.create_coarse_verts_naive(pos)
    assert coarse.lvl() == 2 and coarse.m_sigmas == pytest.approx([0.6] * 3)
    with pytest.raises(ValueError, match="differ by at most 1"):
        coarse.create_coarse_verts_naive(pos).neighbours(lat, 1, False)
    with pytest.raises(ValueError, match="filter extent"):
        lat.im2row(lat, 7, 1, False)


def test_expand_matches_oracle_on_the_same_jitter():
    """Lattice::expand (Lattice.cu:292-348): the positions repeated `point_multiplier` times plus Gaussian noise are inserted into a
    copy of the table.  The jitter is reproducible under torch.manual_seed (same generator, same call shape), so the expanded
    table is compared with the oracle's incremental build on exactly those positions: keys and row ids bit-exact."""
    from lattice_net_amd.synthetic import cube_cloud
    pos_np = cube_cloud(500, 41)
    pos = T(pos_np)
    lat = make_lattice(0.3, 20000)
    lat.begin_splat()
    lat.splat_standalone(pos, torch.ones((500, 3), device=dev()))
    m = lat.nr_lattice_vertices()
    lat.set_values(lat.values()[:m].contiguous())
    keys_before = N(lat.hash_table().m_keys_tensor[:m]).copy()
    torch.manual_seed(0)
    ex = lat.expand(pos, 4, 0.2, True)
    m2 = ex.nr_lattice_vertices()
    assert m2 > m and lat.nr_lattice_vertices() == m
    np.testing.assert_array_equal(N(ex.hash_table().m_keys_tensor[:m]), keys_before)  # existing rows keep their ids
    assert tuple(ex.values().shape) == (m2, 3)
    np.testing.assert_array_equal(N(ex.values()[:m]), N(lat.values()))
    assert float(ex.values()[m:].abs().sum()) == 0.0
    # the same jitter again (Lattice.cu:311-318: repeat, then + randn * stddev), then the oracle's serial insertion order
    torch.manual_seed(0)
    rep = pos.repeat(4, 1)
    jittered = N(rep + torch.randn_like(rep) * 0.2)
    sig = np.full((3,), 0.3, np.float32)
    t = O.OracleHashTable(20000, 3)
    O.build_splat(t, O.scale_positions(pos_np, sig))
    assert t.nr_filled == m and np.array_equal(t.keys[:m], keys_before)
    O.build_splat(t, O.scale_positions(jittered, sig), write=False)
    assert t.nr_filled == m2
    np.testing.assert_array_equal(N(ex.hash_table().m_keys_tensor[:m2]), t.keys[:m2])
    # and the expanded table answers lookups like the oracle's: slice_no_precomputation indices of fresh query points
    q = cube_cloud(300, 42)
    ex.set_values(torch.zeros((m2, 3), device=dev()))
    _, qi, qw = ex.slice_standalone_no_precomputation(T(q))
    _, oi, ow = O.slice_no_precomputation(t, np.zeros((m2, 3), np.float32), O.scale_positions(q, sig))
    np.testing.assert_array_equal(N(qi), oi)
    np.testing.assert_array_equal(N(qw), ow)


def test_create_from_cfg_and_splat(tmp_path):
    from lattice_net_amd import Lattice, SplatLattice
    from lattice_net_amd.synthetic import lidar_cloud
    cfg = tmp_path / "kitti.cfg"
    cfg.write_text('lattice_gpu: {\n    hash_table_capacity: 100000 //comment\n    nr_sigmas: 1\n    sigma_0: "0.9 3" //x\n}\n')
    lat = Lattice.create(str(cfg), "lattice")
    pos_np = lidar_cloud(5000, 2)
    lv, wrap, idx, w = SplatLattice.apply(lat, T(pos_np), torch.ones((5000, 1), device=dev()))
    t, oidx, ow = oracle_table(pos_np, 0.9, 100000)
    assert wrap.lattice is lat and lat.nr_lattice_vertices() == t.nr_filled
    np.testing.assert_array_equal(N(idx), oidx)


def test_scatter_max_and_pointnet_aggregation():
    """the aggregation in front of the hot path (torch_scatter.scatter_max / scatter_add replacements)."""
    from lattice_net_amd import Lattice, ScatterMaxLattice
    from lattice_net_amd.lattice_modules import DistributeLatticeModule, PointNetModule
    from lattice_net_amd.synthetic import lidar_cloud
    pos_np = lidar_cloud(6000, 8)
    lat = Lattice(sigmas=[0.9] * 3, capacity=40000, device=dev())
    dist_lat, distributed, idx, w = DistributeLatticeModule()(lat, T(pos_np), torch.ones((6000, 1), device=dev()))
    m = dist_lat.nr_lattice_vertices()
    _, oidx, _ = oracle_table(pos_np, 0.9, 40000)
    rng = np.random.default_rng(3)
    feat_np = rng.standard_normal((24000, 7)).astype(np.float32)
    feat_np[::5] = np.round(feat_np[::5])  # force ties
    feat = T(feat_np).requires_grad_(True)
    vmax, arg = ScatterMaxLattice.apply(feat, dist_lat, idx)
    omax, oarg = O.scatter_max(feat_np, oidx, m)
    np.testing.assert_array_equal(N(vmax), omax)
    np.testing.assert_array_equal(N(arg), oarg)
    np.testing.assert_array_equal(N(dist_lat.vertex_point_counts(idx)), O.vertex_point_counts(oidx, m))
    g_np = rng.standard_normal((m, 7)).astype(np.float32)
    vmax.backward(T(g_np))
    exp = np.zeros_like(feat_np)
    for c in range(7):
        exp[oarg[:, c], c] = g_np[:, c]
    np.testing.assert_array_equal(N(feat.grad), exp)
    # the module end to end: shapes, masking rules, gradients reach the per-token MLP
    torch.manual_seed(0)
    pn = PointNetModule([16, 32], 32)
    lv, ls = pn(dist_lat, distributed, idx)
    assert lv.shape == (m, 32) and ls.val_dim() == 32 and torch.isfinite(lv).all()
    lv.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in pn.parameters())
    assert pn.layers[0].weight_v.grad.abs().sum() > 0 and pn.layers[0].weight_g.grad.abs().sum() > 0


@pytest.mark.parametrize("v,f", [(64, 64), (32, 32), (32, 64), (64, 32), (96, 64), (128, 128), (16, 48), (20, 24)])
def test_conv_fp16_feature_path_matches_fp64_reference(v, f):
    """BASELINE config 5 (C5): fp16 features, fp32 accumulation.  Forward, gradient wrt values and wrt the filter bank of
    ConvIm2RowLattice on half tensors against an fp64 evaluation of the same fp16 inputs."""
    from lattice_net_amd import ConvIm2RowLattice
    from lattice_net_amd.synthetic import cube_cloud
    pos = cube_cloud(6000, 31)
    lat = make_lattice(0.12, 120000)
    lat.begin_splat()
    lat.just_create_verts(T(pos), False)
    m = lat.nr_lattice_vertices()
    rng = np.random.default_rng(v * 7 + f)
    vals = torch.tensor(rng.standard_normal((m, v)), dtype=torch.float16, device=dev(), requires_grad=True)
    W = torch.tensor(rng.standard_normal((9 * v, f)) / np.sqrt(9 * v), dtype=torch.float16, device=dev(), requires_grad=True)
    G = torch.tensor(rng.standard_normal((m, f)), dtype=torch.float16, device=dev())
    out, wrap = ConvIm2RowLattice.apply(vals, lat, W, 1)
    assert out.dtype == torch.float16 and out.shape == (m, f)
    (out.float() * G.float()).sum().backward()
    assert vals.grad.dtype == torch.float16 and W.grad.dtype == torch.float16
    t, _, _ = oracle_table(pos, 0.12, 120000)
    nbr = torch.from_numpy(O.neighbour_rows(t.keys[:m], t, 1, 1, 1, False).astype(np.int64))
    v64 = vals.detach().cpu().double().requires_grad_(True)
    w64 = W.detach().cpu().double().requires_grad_(True)
    padded = torch.cat([v64, torch.zeros((1, v), dtype=torch.float64)], 0)
    rows = padded[torch.where(nbr >= 0, nbr, torch.full_like(nbr, m))].reshape(m, 9 * v)
    ref = rows @ w64
    (ref * G.cpu().double()).sum().backward()
    # results are rounded to fp16 once (rel 2^-11); the accumulation itself is fp32
    close(N(out.detach().float()), ref.detach().numpy(), scale=float(ref.abs().max()), rtol=2e-3)
    close(N(vals.grad.float()), v64.grad.numpy(), scale=float(v64.grad.abs().max()), rtol=2e-3)
    close(N(W.grad.float()), w64.grad.numpy(), scale=float(w64.grad.abs().max()), rtol=2e-3)


@pytest.mark.parametrize("v,f,n,sigma", [(64, 64, 6000, 0.12), (32, 32, 6000, 0.12), (64, 32, 6000, 0.12), (128, 128, 6000, 0.12),
                                         (64, 64, 30000, 0.05), (32, 64, 30000, 0.05)])
def test_conv_fp16_small_integer_operands_are_exact(v, f, n, sigma):
    """The fp16 kernels (k_conv_f16 / k_conv_f16_tiled / k_grad_filter_mfma_f16) with small-integer operands: every partial sum is an
    integer far below 2^24 (fp32 accumulation) and every result below 2048 (exact in fp16), so forward and both gradients must equal
    the float64 result bit for bit — a wrong operand lane cannot hide in a tolerance."""
    from lattice_net_amd import ConvIm2RowLattice
    from lattice_net_amd.synthetic import cube_cloud
    lat = make_lattice(sigma, 200000)
    lat.begin_splat()
    lat.just_create_verts(T(cube_cloud(n, 31)), False)
    m = lat.nr_lattice_vertices()
    rng = np.random.default_rng(v * 11 + f + n)
    vals_np = rng.integers(-2, 3, (m, v)).astype(np.float64)
    W_np = (rng.integers(-1, 2, (9 * v, f)) * (rng.random((9 * v, f)) < 0.5)).astype(np.float64)
    G_np = (rng.integers(-1, 2, (m, f)) * (rng.random((m, 1)) < 0.125)).astype(np.float64)
    vals = torch.tensor(vals_np, dtype=torch.float16, device=dev(), requires_grad=True)
    W = torch.tensor(W_np, dtype=torch.float16, device=dev(), requires_grad=True)
    out, _ = ConvIm2RowLattice.apply(vals, lat, W, 1)
    (out.float() * torch.tensor(G_np, dtype=torch.float32, device=dev())).sum().backward()
    nbr = torch.from_numpy(N(lat.neighbours(lat, 1, False)).astype(np.int64))
    padded = torch.cat([torch.from_numpy(vals_np), torch.zeros((1, v), dtype=torch.float64)], 0)
    rows = padded[torch.where(nbr >= 0, nbr, torch.full_like(nbr, m))].reshape(m, 9 * v).numpy()
    ref, gw = rows @ W_np, rows.T @ G_np
    gr = torch.from_numpy((G_np @ W_np.T).reshape(m, 9, v))
    gx = torch.zeros((m, v), dtype=torch.float64)
    for e in range(9):
        ok = nbr[:, e] >= 0
        gx.index_add_(0, nbr[ok, e], gr[ok, e])
    assert max(np.abs(ref).max(), np.abs(gw).max(), float(gx.abs().max())) < 2048
    assert np.array_equal(N(out.detach().float()).astype(np.float64), ref)
    assert np.array_equal(N(W.grad.float()).astype(np.float64), gw)
    assert np.array_equal(N(vals.grad.float()).astype(np.float64), gx.numpy())


def test_fp16_feature_path_splat_conv_slice_end_to_end():
    """C5-style chain on half features: splat of fp16 point features (fp32 table), fp16 convolution, fp16 slice, and the
    backward pass through all three, against an fp64 evaluation of the same fp16 inputs."""
    from lattice_net_amd import ConvIm2RowLattice, SliceLattice, SplatLattice
    from lattice_net_amd.synthetic import cube_cloud
    n, v, f = 8000, 32, 32
    pos_np = cube_cloud(n, 41)
    rng = np.random.default_rng(3)
    vals = torch.tensor(rng.standard_normal((n, v)), dtype=torch.float16, device=dev())
    W = torch.tensor(rng.standard_normal((9 * v, f)) / np.sqrt(9 * v), dtype=torch.float16, device=dev(), requires_grad=True)
    G = torch.tensor(rng.standard_normal((n, f)), dtype=torch.float16, device=dev())
    lat = make_lattice(0.15, 120000)
    pos = T(pos_np)
    lv, wrap, idx, w = SplatLattice.apply(lat, pos, vals)
    m = lat.nr_lattice_vertices()
    assert lv.dtype == torch.float32  # accumulated in fp32
    t, oidx, ow = oracle_table(pos_np, 0.15, 120000)
    expect = np.zeros((m, v), np.float64)
    np.add.at(expect, oidx, np.repeat(vals.cpu().double().numpy(), 4, axis=0) * ow[:, None])
    close(N(lv[:m]), expect, rtol=1e-5)
    lvh = lv[:m].half().requires_grad_(True)
    cv, cwrap = ConvIm2RowLattice.apply(lvh, lat, W, 1)
    out = SliceLattice.apply(cv, cwrap.lattice, pos, idx, w)
    assert cv.dtype == torch.float16 and out.dtype == torch.float16 and out.shape == (n, f)
    out.backward(G)
    assert lvh.grad.dtype == torch.float16 and W.grad.dtype == torch.float16
    # fp64 reference of conv + slice and their gradients on the same fp16 inputs
    nbr = torch.from_numpy(O.neighbour_rows(t.keys[:m], t, 1, 1, 1, False).astype(np.int64))
    v64 = lvh.detach().cpu().double().requires_grad_(True)
    w64 = W.detach().cpu().double().requires_grad_(True)
    padded = torch.cat([v64, torch.zeros((1, v), dtype=torch.float64)], 0)
    rows = padded[torch.where(nbr >= 0, nbr, torch.full_like(nbr, m))].reshape(m, 9 * v)
    conv_ref = rows @ w64
    wi = torch.from_numpy(ow.astype(np.float64)).reshape(n, 4, 1)
    ii = torch.from_numpy(oidx.astype(np.int64)).reshape(n, 4)
    out_ref = (conv_ref[ii] * wi).sum(1)
    (out_ref * G.cpu().double()).sum().backward()
    close(N(out.float()), out_ref.detach().numpy(), scale=float(out_ref.abs().max()), rtol=4e-3)
    close(N(lvh.grad.float()), v64.grad.numpy(), scale=float(v64.grad.abs().max()), rtol=4e-3)
    close(N(W.grad.float()), w64.grad.numpy(), scale=float(w64.grad.abs().max()), rtol=4e-3)


def test_c5_full_size_fp16_feature_chain():
    """BASELINE.json configs[4] as ONE run: 4 aggregated scans (480 k points), capacity 400 000, V = F = 64, fp16 point features /
    lattice values / filter bank with fp32 accumulation — splat -> convolution -> slice, forward and backward, against an fp64
    evaluation of the same fp16 inputs on the oracle's indices and neighbour list."""
    from lattice_net_amd import ConvIm2RowLattice, SliceLattice, SplatLattice
    from lattice_net_amd.synthetic import lidar_cloud
    parts = [lidar_cloud(120000, s) + np.array([6.0 * s, 0.0, 0.0], np.float32) for s in range(4)]
    pos_np = np.ascontiguousarray(np.concatenate(parts, 0))
    n, v, f = pos_np.shape[0], 64, 64
    rng = np.random.default_rng(5)
    vals = torch.tensor(rng.standard_normal((n, v)), dtype=torch.float16, device=dev())
    W = torch.tensor(rng.standard_normal((9 * v, f)) / np.sqrt(9 * v), dtype=torch.float16, device=dev(), requires_grad=True)
    G = torch.tensor(rng.standard_normal((n, f)), dtype=torch.float16, device=dev())
    lat = make_lattice(0.9, 400000)
    pos = T(pos_np)
    lv, wrap, idx, w = SplatLattice.apply(lat, pos, vals)
    m = lat.nr_lattice_vertices()
    t, oidx, ow = oracle_table(pos_np, 0.9, 400000)
    assert m == t.nr_filled
    np.testing.assert_array_equal(N(idx), oidx)
    np.testing.assert_array_equal(N(w), ow)
    expect = np.zeros((m, v), np.float64)
    np.add.at(expect, oidx, np.repeat(vals.cpu().double().numpy(), 4, axis=0) * ow[:, None])
    close(N(lv[:m]), expect, rtol=1e-5)  # fp16 rows accumulated in fp32
    lvh = lv[:m].half().requires_grad_(True)
    cv, cwrap = ConvIm2RowLattice.apply(lvh, lat, W, 1)
    out = SliceLattice.apply(cv, cwrap.lattice, pos, idx, w)
    assert cv.dtype == torch.float16 and out.dtype == torch.float16 and out.shape == (n, f)
    out.backward(G)
    torch.cuda.synchronize()
    nbr_np = O.neighbour_rows(t.keys[:m], t, 1, 1, 1, False)
    np.testing.assert_array_equal(N(lat.neighbours(lat, 1, False)), nbr_np)
    nbr = torch.from_numpy(nbr_np.astype(np.int64))
    v64 = lvh.detach().cpu().double().requires_grad_(True)
    w64 = W.detach().cpu().double().requires_grad_(True)
    padded = torch.cat([v64, torch.zeros((1, v), dtype=torch.float64)], 0)
    rows = padded[torch.where(nbr >= 0, nbr, torch.full_like(nbr, m))].reshape(m, 9 * v)
    conv_ref = rows @ w64
    wi = torch.from_numpy(ow.astype(np.float64)).reshape(n, 4, 1)
    ii = torch.from_numpy(oidx.astype(np.int64)).reshape(n, 4)
    out_ref = (conv_ref[ii] * wi).sum(1)
    (out_ref * G.cpu().double()).sum().backward()
    # fp16 outputs: half an ulp of fp16 (4.9e-4 relative) on values of the size of the scale, plus the rounding of the fp16
    # convolution output that the slice reads; the run-to-run order of the hot-vertex atomics is below both
    close(N(cv.float()), conv_ref.detach().numpy(), scale=float(conv_ref.abs().max()), rtol=2e-3)
    close(N(out.float()), out_ref.detach().numpy(), scale=float(out_ref.abs().max()), rtol=4e-3)
    close(N(lvh.grad.float()), v64.grad.numpy(), scale=float(v64.grad.abs().max()), rtol=4e-3)
    close(N(W.grad.float()), w64.grad.numpy(), scale=float(w64.grad.abs().max()), rtol=4e-3)


@pytest.mark.parametrize("d,v", [(3, 64), (3, 12), (2, 20), (4, 8), (5, 40)])
def test_fp16_slice_forward_and_backward_by_width_and_dimension(d, v):
    """k_slice_forward_f16 in both word sizes (16 bytes when the width is a multiple of 8, else 8), for several lattice
    dimensions, with absent vertices (idx = -1) among the inputs, and its backward through the accumulator the forward
    launch zeroes (ln_slice_forward_f16_prepare_backward) — against fp64 on the same fp16 inputs."""
    from lattice_net_amd import SliceLattice
    rng = np.random.default_rng(100 * d + v)
    n = 3000
    pos_np = ((rng.random((n, d), dtype=np.float32) - 0.5) * 3).astype(np.float32)
    lat = make_lattice(0.3, 200000, d=d)
    lat.begin_splat()
    idx, w = lat.just_create_verts(T(pos_np), True)
    m = lat.nr_lattice_vertices()
    idx = idx.clone()
    drop = torch.from_numpy(rng.random(idx.numel()) < 0.05).to(idx.device)
    idx[drop] = -1  # vertices the table could not hold read as absent (Lattice.cu:771-786)
    vals = torch.tensor(rng.standard_normal((m, v)), dtype=torch.float16, device=dev(), requires_grad=True)
    G = torch.tensor(rng.standard_normal((n, v)), dtype=torch.float16, device=dev())
    for _ in range(2):  # (the second pass meets a dirty accumulator from the first one's backward)
        vals.grad = None
        lat.set_values(vals.detach())
        out = SliceLattice.apply(vals, lat, T(pos_np), idx, w)
        assert out.dtype == torch.float16 and out.shape == (n, v)
        out.backward(G)
        assert vals.grad.dtype == torch.float16
    i64 = idx.cpu().long().reshape(n, d + 1)
    ok = (i64 >= 0).double().unsqueeze(-1)
    w64 = w.cpu().double().reshape(n, d + 1, 1) * ok
    v64 = vals.detach().cpu().double().requires_grad_(True)
    ref = (v64[i64.clamp(min=0)] * w64).sum(1)
    (ref * G.cpu().double()).sum().backward()
    close(N(out.detach().float()), ref.detach().numpy(), scale=float(ref.abs().max()), rtol=2e-3)
    close(N(vals.grad.float()), v64.grad.numpy(), scale=float(v64.grad.abs().max()), rtol=2e-3)


@pytest.mark.parametrize("d,n,v,c", [(2, 3000, 64, 9), (2, 1111, 32, 20), (4, 2000, 64, 12), (2, 2500, 96, 32)])
def test_slice_classify_other_lattice_dimensions(d, n, v, c):
    """pos_dim 2 takes the wave-tiled kernels of ln_classify.hip in their (d + 1) = 3 instantiation (tokens per tile: 192 / 48, not a
    whole number of waves); pos_dim 4 the general kernels.  Logits bit for bit, gradients 1e-5, against the oracle."""
    from lattice_net_amd import SliceClassifyLattice
    rng = np.random.default_rng(100 * d + v)
    pos_np = rng.uniform(-1.0, 1.0, (n, d)).astype(np.float32)
    lat = make_lattice(0.2, 30000, d=d)
    lat.begin_splat()
    idx, w = lat.just_create_verts(T(pos_np), True)
    m = lat.nr_lattice_vertices()
    t, oidx, ow = oracle_table(pos_np, 0.2, 30000)
    assert t.nr_filled == m and np.array_equal(N(idx), oidx)
    vals_np = rng.standard_normal((m, v)).astype(np.float32)
    dw_np = (0.1 * rng.standard_normal((n, d + 1))).astype(np.float32)
    lw_np, lb_np = rng.standard_normal((c, v)).astype(np.float32), rng.standard_normal((c,)).astype(np.float32)
    gl_np = rng.standard_normal((n, c)).astype(np.float32)
    vals, dw, lw, lb = (T(x).requires_grad_(True) for x in (vals_np, dw_np, lw_np, lb_np))
    logits = SliceClassifyLattice.apply(vals, lat, T(pos_np), dw, lw, lb, c, idx, w)
    np.testing.assert_array_equal(N(logits), O.slice_classify(vals_np, dw_np, lw_np, lb_np, oidx, ow, n))
    logits.backward(T(gl_np))
    gv, gd, gw, gb = O.slice_classify_backwards(gl_np, vals_np, dw_np, lw_np, lb_np, oidx, ow, n)
    close(N(vals.grad), gv)
    close(N(dw.grad), gd)
    close(N(lw.grad), gw)
    close(N(lb.grad), gb)
