"""The fused glue kernels of csrc/ln_glue.hip / ln_csr.hip (weight normalisation, DistributeLatticeModule's per-token tail, the
vertex-side reduction of PointNetModule, the one-launch table arena) against the torch operator chains they replace — the chains
are the reference's own formulation (lattice_modules.py:72-94, 688-712; utils.py:72-158)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


ALL_FUSED = ("weight_norm", "distribute", "pointnet")


def dev():
    return torch.device("cuda", 0)


@pytest.mark.parametrize("rows,cols,g_dim", [(16, 5, 0), (32, 16, 0), (64, 32, 0), (297, 32, 1), (1152, 128, 1), (7, 1, 0), (1, 9, 1),
                                             (1000, 3, 0), (3, 1000, 1)])
def test_weight_norm_matches_the_torch_chain(rows, cols, g_dim):
    from lattice_net_amd.lattice_modules import WeightNormFunction
    gen = torch.Generator().manual_seed(rows * 31 + cols)
    v64 = torch.randn((rows, cols), generator=gen, dtype=torch.float64)
    g_shape = (rows, 1) if g_dim == 0 else (1, cols)
    g64 = torch.rand(g_shape, generator=gen, dtype=torch.float64) + 0.5
    gw64 = torch.randn((rows, cols), generator=gen, dtype=torch.float64)
    v64.requires_grad_(True)
    g64.requires_grad_(True)
    w64 = v64 * (g64 / v64.norm())
    w64.backward(gw64)
    v = v64.detach().float().to(dev()).requires_grad_(True)
    g = g64.detach().float().to(dev()).requires_grad_(True)
    w = WeightNormFunction.apply(v, g, g_dim)
    w.backward(gw64.float().to(dev()))
    for got, exp in ((w, w64), (v.grad, v64.grad), (g.grad, g64.grad)):
        exp = exp.detach().numpy()
        np.testing.assert_allclose(got.detach().cpu().numpy(), exp, rtol=0, atol=2e-6 * max(float(np.abs(exp).max()), 1e-30))
    assert g.grad.shape == g.shape


def test_weight_normed_layers_take_the_fused_form_and_agree_with_the_chain():
    import lattice_net_amd.lattice_modules as M
    torch.manual_seed(1)
    lin = M.LinearWN(12, 24, device=dev())
    x = torch.randn((50, 12), device=dev())
    outs = {}
    for fused in (True, False):
        M.FUSED_GLUE = set(ALL_FUSED) if fused else set(ALL_FUSED) - {"weight_norm"}
        try:
            lin.zero_grad()
            y = lin(x)
            y.square().sum().backward()
            outs[fused] = (y.detach().clone(), lin.weight_v.grad.clone(), lin.weight_g.grad.clone())
        finally:
            M.FUSED_GLUE = set(ALL_FUSED)
    for a, b in zip(outs[True], outs[False]):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max()))


def _distribute(n=5000, seed=3, sigma=0.9, values=2):
    from lattice_net_amd import Lattice
    from lattice_net_amd.lattice_modules import DistributeLatticeModule
    from lattice_net_amd.synthetic import lidar_cloud
    pos = torch.from_numpy(lidar_cloud(n, seed)).to(dev())
    vals = torch.from_numpy(np.random.default_rng(seed).standard_normal((n, values)).astype(np.float32)).to(dev())
    lat = Lattice(sigmas=[sigma] * 3, capacity=8 * n, device=dev())
    return DistributeLatticeModule()(lat, pos, vals)


def test_distribute_tail_matches_the_torch_chain():
    import lattice_net_amd.lattice_modules as M
    got = _distribute()
    M.FUSED_GLUE = set(ALL_FUSED) - {"distribute"}
    try:
        exp = _distribute()
    finally:
        M.FUSED_GLUE = set(ALL_FUSED)
    assert got[0].nr_lattice_vertices() == exp[0].nr_lattice_vertices()
    assert torch.equal(got[2], exp[2]) and torch.equal(got[3], exp[3])
    idx = got[2].cpu().numpy()
    assert (idx <= 0).any()                                     # tokens of vertex 0 exist: their rows are zero
    assert not got[1].cpu().numpy()[idx <= 0].any()
    # (the position sums of vertices with several segments are combined with float atomics: the last bit may differ run to run)
    a, b = got[1].cpu().numpy(), exp[1].cpu().numpy()
    assert np.array_equal(a == 0, b == 0) or np.abs(a - b).max() < 1e-6
    np.testing.assert_allclose(a, b, rtol=0, atol=2e-6 * float(np.abs(b).max()))
    assert np.array_equal(a[:, 3:], b[:, 3:])                   # everything but the centred positions: copied bit for bit


def test_pointnet_reduction_is_bitwise_the_torch_chain():
    import lattice_net_amd.lattice_modules as M
    dist_lat, distributed, idx, _ = _distribute(n=6000, seed=8)
    torch.manual_seed(0)
    pn = M.PointNetModule([16, 32], 32, nr_input_channels=distributed.shape[1] - 1)
    seen = {}
    hook = pn.last_conv.register_forward_pre_hook(lambda mod, args: seen.__setitem__("in", args[0].detach().clone()))
    outs = {}
    for fused in (True, False):
        M.FUSED_GLUE = set(ALL_FUSED) if fused else set(ALL_FUSED) - {"pointnet"}
        try:
            pn.zero_grad()
            lat = dist_lat  # (PointNetModule only sets values on it)
            lv, _ = pn(lat, distributed, idx)
            lv.square().mean().backward()
            outs[fused] = (seen["in"], lv.detach().clone(), [p.grad.clone() for p in pn.parameters()])
        finally:
            M.FUSED_GLUE = set(ALL_FUSED)
    hook.remove()
    a, b = outs[True], outs[False]
    assert torch.equal(a[0], b[0])                               # the reduced rows entering the convolution: identical
    m = a[0].shape[0]
    counts = dist_lat.vertex_point_counts(idx).cpu().numpy()
    dropped = (counts < 4) | (np.arange(m) == 0)
    assert dropped.any() and not a[0].cpu().numpy()[dropped].any()
    assert torch.equal(a[1], b[1])
    for ga, gb in zip(a[2], b[2]):
        torch.testing.assert_close(ga, gb, rtol=1e-5, atol=1e-6 * float(gb.abs().max()) + 1e-12)


def test_fresh_table_buffers_come_initialised_from_one_arena():
    from lattice_net_amd.lattice import _TableStorage
    s = _TableStorage(1001, 3, dev(), spare_row_width=5)
    assert s.keys.shape == (1001, 3) and not s.keys.any()
    assert s.entries.shape == (1001,) and bool((s.entries == -1).all())
    assert s.slot_cnt.shape == (1001,) and not s.slot_cnt.any()
    assert s.fresh_counters.shape == (2,) and not s.fresh_counters.any()
    assert s.fresh_row.shape == (1, 5) and s.fresh_row.dtype == torch.float32 and not s.fresh_row.any()
    for t in (s.keys, s.entries, s.slot_cnt, s.fresh_counters, s.fresh_row):
        assert t.data_ptr() % 16 == 0
    c = s.clone()
    assert c.keys.data_ptr() != s.keys.data_ptr() and bool((c.entries == -1).all())


@pytest.mark.parametrize("n,c,ignore", [(120000, 20, None), (5000, 20, 0), (777, 5, 3), (64, 4, 1), (1, 2, None)])
def test_fused_nll_matches_the_gather_formulation(n, c, ignore):
    import lattice_net_amd.losses as L
    gen = torch.Generator().manual_seed(n + c)
    logits = torch.randn((n, c), generator=gen)
    target = torch.randint(0, c, (n,), generator=gen)
    if n == 64:
        target[:] = ignore                                       # every label ignored: loss 0, gradient 0
    out = {}
    for fused in (True, False):
        L.FUSED_NLL = fused
        try:
            x = logits.to(dev()).requires_grad_(True)
            lp = torch.log_softmax(x.double() if not fused else x, dim=1)
            loss = L.nll_loss_gather(lp, target.to(dev()), ignore)
            (loss * 3.0).backward()
            out[fused] = (float(loss), x.grad.double().cpu().numpy())
        finally:
            L.FUSED_NLL = True
    assert abs(out[True][0] - out[False][0]) <= 2e-6 * max(abs(out[False][0]), 1e-30) + 1e-12
    np.testing.assert_allclose(out[True][1], out[False][1], rtol=0, atol=2e-6 * max(float(np.abs(out[False][1]).max()), 1e-30))


def test_split_bank_of_an_unchanged_filter_is_reused_and_invalidated_by_in_place_changes():
    import ctypes as C
    import lattice_net_amd as L
    from lattice_net_amd.synthetic import lidar_cloud
    lib = L.load_library()
    pos = torch.from_numpy(lidar_cloud(60000, 2)).to(dev())
    lat = L.Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev())
    lat.begin_splat()
    lat.just_create_verts(pos, False)
    m = lat.nr_lattice_vertices()
    vals = torch.randn((m, 128), device=dev())
    bank = torch.randn((9 * 128, 128), device=dev()) * 0.05
    lat.set_values(vals)

    def conv_and_count():
        assert lib.ln_profile_begin(b"k_conv_split_bank", 16) == 0
        y = lat.convolve_im2row_standalone(bank, 1, lat, False).values().clone()
        torch.cuda.synchronize()
        ms, cnt = C.c_double(0), C.c_int(0)
        assert lib.ln_profile_end(C.byref(ms), C.byref(cnt)) == 0
        return y, cnt.value

    from lattice_net_amd import lattice as LT
    _, na = conv_and_count()
    _, nb = conv_and_count()
    assert na >= 1 and nb >= 1                                      # the cache is opt-in (round 6, advisor): off by default, every call splits
    prev = LT.set_bank_cache(True)
    try:
        _bank_cache_checks(lat, bank, vals, lib, conv_and_count)
    finally:
        LT.set_bank_cache(prev)


def _bank_cache_checks(lat, bank, vals, lib, conv_and_count):
    import ctypes as C
    y0, n0 = conv_and_count()
    y1, n1 = conv_and_count()
    assert n0 >= 1 and n1 == 0 and torch.equal(y0, y1)             # second call: no split launch, same result
    side = torch.cuda.Stream()                                       # an entry remembers the stream that produced it: a call on another
    side.wait_stream(torch.cuda.current_stream())                    # stream does not take the hit (the bank may still be being written there)
    with torch.cuda.stream(side):
        _, ns = conv_and_count()
    torch.cuda.current_stream().wait_stream(side)
    assert ns >= 1
    y0, _ = conv_and_count()                                         # (back on the first stream: the side stream's entry replaced ours)
    _, n1 = conv_and_count()
    assert n1 == 0
    bank.mul_(2.0)                                                   # in place: the version counter moves, the bank is split again
    y2, n2 = conv_and_count()
    assert n2 >= 1
    torch.testing.assert_close(y2, 2.0 * y0, rtol=1e-5, atol=1e-5 * float(y0.abs().max()))
    wanted = bank.clone().requires_grad_(True)                       # a trainable filter is never served from the cache, not even under
    lat.set_values(vals)                                             # no_grad: fused optimizers update parameters without moving the version counter
    with torch.no_grad():
        assert lib.ln_profile_begin(b"k_conv_split_bank", 16) == 0
        for _ in range(2):
            lat.convolve_im2row_standalone(wanted, 1, lat, False)
        torch.cuda.synchronize()
        ms, cnt = C.c_double(0), C.c_int(0)
        assert lib.ln_profile_end(C.byref(ms), C.byref(cnt)) == 0
    assert cnt.value >= 2


def test_fp16_features_with_fp32_master_weights_return_the_filter_gradient_in_fp32():
    import lattice_net_amd as L
    from lattice_net_amd.synthetic import lidar_cloud
    pos = torch.from_numpy(lidar_cloud(30000, 5)).to(dev())
    lat = L.Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev())
    lat.begin_splat()
    lat.just_create_verts(pos, False)
    m = lat.nr_lattice_vertices()
    torch.manual_seed(2)
    vals = torch.randn((m, 64), device=dev()).half()
    bank = torch.randn((9 * 64, 64), device=dev()) * 0.05
    g = torch.randn((m, 64), device=dev()).half()
    res = {}
    for mixed in (True, False):
        lv = vals.clone().requires_grad_(True)
        w = bank.clone().requires_grad_(True)
        y, _ = L.ConvIm2RowLattice.apply(lv, lat, w if mixed else w.half(), 1)
        y.backward(g)
        res[mixed] = (y.detach(), lv.grad, w.grad)
    assert res[True][2].dtype == torch.float32 and res[True][1].dtype == torch.float16
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    a, b = res[True][2], res[False][2]                             # fp32 sums vs the same sums rounded to fp16 and back
    assert float((a - b).abs().max()) <= 1e-3 * float(a.abs().max())
    assert float((a.half().float() - b).abs().max()) == 0.0


def test_segment_reduce_variants_agree_and_match_a_float64_scatter():
    """The workgroup-combining form of the CSR segment reduce (LnCsr.dense / wide rows) against the plain form and against an fp64
    index_add, on a cloud with hot vertices (hundreds of tokens each) — at 32 channels (8 lanes per segment) and 96 (32 lanes)."""
    import lattice_net_amd as L
    import lattice_net_amd.lattice as LL
    from lattice_net_amd.synthetic import lidar_cloud
    n = 60000
    pos = torch.from_numpy(lidar_cloud(n, 11)).to(dev())
    lat = L.Lattice(sigmas=[1.4] * 3, capacity=100000, device=dev())      # coarse cells: many tokens per vertex
    lat.begin_splat()
    idx, w = lat.splat_standalone(pos, torch.zeros((n, 1), device=dev()))
    m = lat.nr_lattice_vertices()
    counts = np.bincount(idx.cpu().numpy()[idx.cpu().numpy() >= 0], minlength=m)
    assert counts.max() > 200 and 4 * n / m > 10                           # many tokens per vertex, with hot vertices (the test sets the hint itself)
    for v in (32, 96):
        vals = torch.randn((n, v), device=dev())
        ref = torch.zeros((m, v), dtype=torch.float64, device=dev())
        ok = idx >= 0
        ref.index_add_(0, idx[ok].long(), (vals.double().repeat_interleave(4, 0) * w.double().unsqueeze(1))[ok])
        got = {}
        saved = LL._DENSE_TOKENS_PER_VERTEX
        for name, thr in (("combined", 0.0), ("plain", 1e9)):
            LL._DENSE_TOKENS_PER_VERTEX = thr
            try:
                dst = torch.zeros((m, v), dtype=torch.float32, device=dev())
                lat._scatter_rows(vals, idx, w, dst, v, 4, v)
                got[name] = dst
            finally:
                LL._DENSE_TOKENS_PER_VERTEX = saved
        scale = float(ref.abs().max())
        for name, dst in got.items():
            assert float((dst.double() - ref).abs().max()) <= 2e-6 * scale, name
        assert float((got["combined"] - got["plain"]).abs().max()) <= 2e-6 * scale


def test_segment_max_with_hot_vertices_is_exact():
    """k_csr_segment_max in its run-combining form (C = 32, 64: 8 / 16 lanes per segment) and its plain form (C = 7) against the
    NumPy oracle in canonical row order, with ties, on a cloud whose hottest vertices span several waves."""
    from oracle import lattice_oracle as O
    import lattice_net_amd as L
    import lattice_net_amd.lattice as LL
    from lattice_net_amd import ScatterMaxLattice
    from lattice_net_amd.synthetic import lidar_cloud
    n = 40000
    pos_np = lidar_cloud(n, 12)
    previous_order = LL.set_row_order("canonical")
    try:
        lat = L.Lattice(sigmas=[1.4] * 3, capacity=100000, device=dev())
        lat.begin_splat()
        idx, _ = lat.splat_standalone(torch.from_numpy(pos_np).to(dev()), torch.zeros((n, 1), device=dev()))
        m = lat.nr_lattice_vertices()
        tab = O.OracleHashTable(100000, 3)
        oidx, _ = O.build_splat(tab, O.scale_positions(pos_np, np.full((3,), 1.4, np.float32)))
        assert np.array_equal(oidx.ravel(), idx.cpu().numpy().ravel())
        assert np.bincount(oidx[oidx >= 0].ravel()).max() > 600          # more than 32 segments: spans waves and workgroups
        rng = np.random.default_rng(5)
        for c in (32, 64, 7):
            f = rng.standard_normal((4 * n, c)).astype(np.float32)
            f[::3] = np.round(f[::3])
            vmax, arg = ScatterMaxLattice.apply(torch.from_numpy(f).to(dev()), lat, idx)
            omax, oarg = O.scatter_max(f, oidx, m)
            assert np.array_equal(vmax.cpu().numpy(), omax) and np.array_equal(arg.cpu().numpy(), oarg), c
            assert np.array_equal(lat.vertex_point_counts(idx).cpu().numpy(), O.vertex_point_counts(oidx, m))
    finally:
        LL.set_row_order(previous_order)
