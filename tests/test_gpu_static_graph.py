"""Static-rows mode and hipGraph replay of the whole hot-path step (run on a real MI355X: `pytest -m gpu`).

The reference reads the vertex count back after every build (Lattice.cu:1320-1352); `Lattice.set_static_rows` replaces
that readback by a fixed row bound so that splat -> conv -> slice, forward + backward, can be captured into ONE hipGraph.
Checked here, through the C ABI, against the CPU oracle (oracle/lattice_oracle.py):
  * a replayed graph reproduces the oracle's forward output and filter gradient (1e-5 relative, BASELINE.json);
  * replaying after the positions / features were overwritten IN PLACE with another cloud gives that cloud's result
    (the graph redoes the hash build, it does not cache a lattice);
  * rows beyond the real vertex count stay zero; a bound that is too small is reported.
"""
import numpy as np
import pytest
import torch

from oracle import lattice_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def dev():
    return torch.device("cuda", 0)


def oracle_step(pos_np, vals_np, w_np, g_np, sigma, cap):
    n, v = vals_np.shape
    t = O.OracleHashTable(cap, 3)
    idx, w = O.build_splat(t, O.scale_positions(pos_np, np.full((3,), sigma, np.float32)))
    m = t.nr_filled
    lv = np.zeros((m, v), np.float32)
    O.splat_accumulate(lv, vals_np, idx, w)
    nbr = O.neighbour_rows(t.keys[:m], t, 1, 1, 1, False)
    cv = O.conv_forward(nbr, lv, w_np)
    out = O.slice_with_precomputation(cv, idx, w, n)
    g_c = O.slice_backwards(g_np, idx, w, m)
    g_w = O.im2row(nbr, lv).astype(np.float64).T @ g_c.astype(np.float64)
    return m, idx, out, g_w


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) / max(float(np.max(np.abs(b))), 1e-30)


def test_graph_replay_matches_oracle_and_follows_new_inputs():
    import lattice_net_amd as L
    from lattice_net_amd.synthetic import lidar_cloud

    torch.autograd.set_multithreading_enabled(False)
    n, v, f, sigma, cap = 6000, 32, 32, 0.9, 30000
    rng = np.random.default_rng(3)
    clouds = [lidar_cloud(n, 11), lidar_cloud(n, 12)]
    feats = [rng.standard_normal((n, v)).astype(np.float32) for _ in range(2)]
    w_np = (rng.standard_normal((9 * v, f)) / np.sqrt(9 * v)).astype(np.float32)
    g_np = rng.standard_normal((n, f)).astype(np.float32)
    refs = [oracle_step(clouds[k], feats[k], w_np, g_np, sigma, cap) for k in range(2)]

    pos = torch.from_numpy(clouds[0]).to(dev())
    vals = torch.from_numpy(feats[0]).to(dev())
    G = torch.from_numpy(g_np).to(dev())
    W = torch.from_numpy(w_np).to(dev()).requires_grad_(True)
    lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev())
    bound = ((max(r[0] for r in refs) + 300 + 255) // 256) * 256
    lat.set_static_rows(bound)
    st = {}

    def step():
        W.grad = None
        lv, _, idx, w = L.SplatLattice.apply(lat, pos, vals)
        m = lat.nr_lattice_vertices()
        assert m == bound
        lv = lv[:m].requires_grad_(True)
        cv, cw = L.ConvIm2RowLattice.apply(lv, lat, W, 1)
        out = L.SliceLattice.apply(cv, cw.lattice, pos, idx, w)
        out.backward(G)
        st.update(out=out, idx=idx, cv=cv, gv=lv.grad)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    st.clear()  # no reference to the warm-up's autograd graph (and with it W's AccumulateGrad node) survives into the capture
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for k in (0, 1, 0):
        pos.copy_(torch.from_numpy(clouds[k]))
        vals.copy_(torch.from_numpy(feats[k]))
        graph.replay()
        torch.cuda.synchronize()
        m_ref, idx_ref, out_ref, gw_ref = refs[k]
        nr, status = lat.static_build_report()
        assert (nr, status) == (m_ref, 0)
        assert np.array_equal(st["idx"].cpu().numpy(), idx_ref), "splat indices differ from the oracle"
        assert rel(st["out"].detach().cpu().numpy(), out_ref) < RTOL
        assert rel(W.grad.cpu().numpy(), gw_ref) < RTOL
        cv = st["cv"].detach().cpu().numpy()
        assert cv.shape[0] == bound and not cv[m_ref:].any(), "rows beyond the vertex count must stay zero"
        assert not st["gv"].cpu().numpy()[m_ref:].any()


def test_static_bound_too_small_is_reported():
    import lattice_net_amd as L
    from lattice_net_amd.synthetic import lidar_cloud

    n, sigma, cap = 3000, 0.9, 20000
    pos = torch.from_numpy(lidar_cloud(n, 5)).to(dev())
    vals = torch.ones((n, 8), device=dev())
    lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev())
    L.SplatLattice.apply(lat, pos, vals)
    m = lat.nr_lattice_vertices()
    lat.set_static_rows(m // 2)
    # the whole chain must stay inside its (too short) tensors: vertices beyond the bound are left un-inserted (LnTable.row_limit)
    W = torch.randn((9 * 8, 16), device=dev(), requires_grad=True)
    lv, _, idx, w = L.SplatLattice.apply(lat, pos, vals)
    assert lat.nr_lattice_vertices() == m // 2 and lv.shape[0] == m // 2
    lvb = lv.clone().requires_grad_(True)
    cv, cw = L.ConvIm2RowLattice.apply(lvb, lat, W, 1)
    out = L.SliceLattice.apply(cv, cw.lattice, pos, idx, w)
    out.sum().backward()
    torch.cuda.synchronize()
    assert int(idx.max()) < m // 2 and int((idx < 0).sum()) > 0
    assert torch.isfinite(out).all() and torch.isfinite(lvb.grad).all() and torch.isfinite(W.grad).all()
    with pytest.raises(L.LatticeNetHipError, match="static row bound"):
        lat.static_build_report()
    lat.set_static_rows(None)
    L.SplatLattice.apply(lat, pos, vals)
    assert lat.nr_lattice_vertices() == m


def test_region_planes_change_placement_not_results():
    """kd region planes (LnCsr.planes) only steer which XCD walks which segments: indices stay bit-exact with the oracle,
    splatted values / slice backward within 1e-5, for balanced planes and for deliberately lopsided ones."""
    import lattice_net_amd as L
    from lattice_net_amd.synthetic import lidar_cloud

    n, v, sigma, cap = 20000, 32, 0.9, 40000
    rng = np.random.default_rng(7)
    pos_np = lidar_cloud(n, 21)
    vals_np = rng.standard_normal((n, v)).astype(np.float32)
    g_np = rng.standard_normal((n, v)).astype(np.float32)
    t = O.OracleHashTable(cap, 3)
    idx_ref, w_ref = O.build_splat(t, O.scale_positions(pos_np, np.full((3,), sigma, np.float32)))
    m = t.nr_filled
    lv_ref = np.zeros((m, v), np.float32)
    O.splat_accumulate(lv_ref, vals_np, idx_ref, w_ref)
    gb_ref = O.slice_backwards(g_np, idx_ref, w_ref, m)

    pos = torch.from_numpy(pos_np).to(dev())
    vals = torch.from_numpy(vals_np).to(dev())
    G = torch.from_numpy(g_np).to(dev())
    lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev())
    _, _, idx, _ = L.SplatLattice.apply(lat, pos, vals)
    balanced = lat.balanced_region_planes(idx)
    for planes in (balanced, [10 ** 6] * 7, [-10 ** 6] * 7, [0, -50, 50, 3, -3, 7, -7]):
        lat.set_region_planes(planes)
        lv, _, idx, w = L.SplatLattice.apply(lat, pos, vals)
        assert lat.nr_lattice_vertices() == m
        assert np.array_equal(idx.cpu().numpy(), idx_ref)
        assert rel(lv[:m].cpu().numpy(), lv_ref) < RTOL
        lvm = lv[:m].clone().requires_grad_(True)
        out = L.SliceLattice.apply(lvm, lat, pos, idx, w)
        out.backward(G)
        assert rel(lvm.grad.cpu().numpy(), gb_ref) < RTOL
    lat.set_region_planes(None)
    lv, _, idx, w = L.SplatLattice.apply(lat, pos, vals)
    assert rel(lv[:m].cpu().numpy(), lv_ref) < RTOL


@pytest.mark.parametrize("d", [1, 2, 4, 5])
def test_region_planes_other_dimensions(d):
    """The kd region function uses key[0], key[1 % d], key[2 % d]: any d builds the same lattice with or without planes."""
    import lattice_net_amd as L

    n, v, cap = 6000, 8, 40000
    rng = np.random.default_rng(100 + d)
    pos_np = (rng.standard_normal((n, d)) * 2.0).astype(np.float32)
    vals_np = rng.standard_normal((n, v)).astype(np.float32)
    sigma = 0.6
    t = O.OracleHashTable(cap, d)
    idx_ref, w_ref = O.build_splat(t, O.scale_positions(pos_np, np.full((d,), sigma, np.float32)))
    m = t.nr_filled
    lv_ref = np.zeros((m, v), np.float32)
    O.splat_accumulate(lv_ref, vals_np, idx_ref, w_ref)
    pos = torch.from_numpy(pos_np).to(dev())
    vals = torch.from_numpy(vals_np).to(dev())
    lat = L.Lattice(sigmas=[sigma] * d, capacity=cap, device=dev())
    _, _, idx, _ = L.SplatLattice.apply(lat, pos, vals)
    assert lat.nr_lattice_vertices() == m
    lat.set_region_planes(lat.balanced_region_planes(idx))
    lv, _, idx, w = L.SplatLattice.apply(lat, pos, vals)
    assert lat.nr_lattice_vertices() == m
    assert np.array_equal(idx.cpu().numpy(), idx_ref)
    assert rel(lv[:m].cpu().numpy(), lv_ref) < RTOL


def test_captured_step_helper_two_scans_in_flight():
    """lattice_net_amd.CapturedStep: calibration + capture in one call; two captured scans replayed concurrently on their own
    streams give each scan's oracle result."""
    import lattice_net_amd as L
    from lattice_net_amd.synthetic import lidar_cloud

    torch.autograd.set_multithreading_enabled(False)
    n, v, f, sigma, cap = 5000, 32, 32, 0.9, 30000
    rng = np.random.default_rng(9)
    w_np = (rng.standard_normal((9 * v, f)) / np.sqrt(9 * v)).astype(np.float32)
    W = torch.from_numpy(w_np).to(dev()).requires_grad_(True)
    scans = []
    for k in range(2):
        pos_np = lidar_cloud(n, 40 + k)
        vals_np = rng.standard_normal((n, v)).astype(np.float32)
        g_np = rng.standard_normal((n, f)).astype(np.float32)
        sc = {"ref": oracle_step(pos_np, vals_np, w_np, g_np, sigma, cap), "st": {},
              "pos": torch.from_numpy(pos_np).to(dev()), "vals": torch.from_numpy(vals_np).to(dev()), "G": torch.from_numpy(g_np).to(dev()),
              "lat": L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev())}

        def step(sc=sc):
            W.grad = None
            lv, _, idx, w = L.SplatLattice.apply(sc["lat"], sc["pos"], sc["vals"])
            lv = lv[:sc["lat"].nr_lattice_vertices()].requires_grad_(True)
            cv, cw = L.ConvIm2RowLattice.apply(lv, sc["lat"], W, 1)
            out = L.SliceLattice.apply(cv, cw.lattice, sc["pos"], idx, w)
            out.backward(sc["G"])
            sc["st"].update(out=out, idx=idx, gw=W.grad)

        sc["cap"] = L.CapturedStep(step, [sc["lat"]], region_indices=lambda sc=sc: sc["st"]["idx"], stream=torch.cuda.Stream(),
                                   before_capture=sc["st"].clear)
        scans.append(sc)
    torch.cuda.synchronize()
    for _ in range(3):
        for sc in scans:
            sc["cap"].launch()
    torch.cuda.synchronize()
    for sc in scans:
        m_ref, idx_ref, out_ref, gw_ref = sc["ref"]
        assert sc["cap"].check() == [m_ref]
        assert np.array_equal(sc["st"]["idx"].cpu().numpy(), idx_ref)
        assert rel(sc["st"]["out"].detach().cpu().numpy(), out_ref) < RTOL
        assert rel(sc["st"]["gw"].cpu().numpy(), gw_ref) < RTOL


def test_concurrent_streams_are_distinct_and_start_with_the_current_stream():
    """capture.concurrent_streams: k streams, the caller's current stream first, no stream twice; the streams it returns
    pass its own overlap probe again (two spin kernels side by side take about as long as one)."""
    import time
    from lattice_net_amd.capture import concurrent_streams
    streams = concurrent_streams(3)
    assert len(streams) == 3 and streams[0] == torch.cuda.current_stream()
    assert len({s.cuda_stream for s in streams}) == 3
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000)
    cycles = 400_000
    ev0.record()
    torch.cuda._sleep(cycles)
    ev1.record()
    torch.cuda.synchronize()
    single_us = ev0.elapsed_time(ev1) * 1e3
    best = 1e30
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in streams[1:]:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cycles)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e6)
    assert best < 1.7 * single_us + 100.0, (best, single_us)
