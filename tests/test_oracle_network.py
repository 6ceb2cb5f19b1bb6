"""CPU checks of the whole-network checker itself (tests/oracle_lattice.py): the LNN definition evaluated in float64 on an
oracle-backed lattice must be differentiable consistently — the analytic gradients that flow through the backward formulas
of the reference (lattice_funcs.py:298-313, 375-387, 440-452; LatticeGPU.cuh:3648-3814) are compared with central
differences of the loss.  Runs without a GPU."""
import tempfile
import textwrap

import numpy as np
import torch

CFG = textwrap.dedent("""
    model: {
        positions_mode: "xyz"
        values_mode: "none"
        pointnet_layers: [16,32]
        pointnet_start_nr_channels: 32
        nr_downsamples: 2
        nr_blocks_down_stage: [1,1]
        nr_blocks_bottleneck: 1
        nr_blocks_up_stage: [1,1]
        nr_levels_down_with_normal_resnet: 1
        nr_levels_up_with_normal_resnet: 1
        compression_factor: 1.0
        dropout_last_layer: 0.0
    }
    lattice_gpu: {
        hash_table_capacity: 60000
        nr_sigmas: 1
        sigma_0: "0.08 3"
    }
""")


def make_oracle_case(n=1500, nr_classes=6, seed=0):
    """(network in float64 on the CPU, oracle lattice, positions f32, values f64, target)."""
    from lattice_net_amd import ModelParams
    from lattice_net_amd.models import LNN
    from lattice_net_amd.synthetic import box_surface_cloud
    from tests.oracle_lattice import OracleLattice
    with tempfile.NamedTemporaryFile("w", suffix=".cfg") as f:
        f.write(CFG)
        f.flush()
        mp = ModelParams.create(f.name)
    torch.manual_seed(seed)
    lattice = OracleLattice([0.08] * 3, 60000)  # (before the network: the modules size their banks from the lattice dimension)
    net = LNN(nr_classes, mp, device="cpu").double()
    pos = torch.from_numpy(box_surface_cloud(n, seed))
    vals = torch.zeros((n, 1), dtype=torch.float64)
    target = torch.from_numpy(np.random.default_rng(seed).integers(0, nr_classes, n))
    return net, lattice, pos, vals, target


def test_oracle_network_gradients_match_central_differences():
    net, lattice, pos, vals, target = make_oracle_case(n=600)

    def loss_of():
        logsoftmax, _ = net(lattice, pos, vals)
        return torch.nn.functional.nll_loss(logsoftmax, target)

    loss = loss_of()
    loss.backward()
    named = dict(net.named_parameters())
    assert all(p.grad is not None for p in named.values())
    rng = np.random.default_rng(1)
    # one filter bank / linear weight of every stage of the network
    picks = ["point_net.layers.0.weight_v", "point_net.last_conv.weight_v", "resnet_blocks_per_down_lvl_list.0.0.conv1.conv.weight",
             "coarsens_list.0.coarse.weight", "coarsens_list.1.coarse.weight", "resnet_blocks_bottleneck.0.conv.conv.weight",
             "finefy_list.0.fine.weight", "finefy_list.1.fine.weight", "resnet_blocks_per_up_lvl_list.1.0.conv2.conv.weight",
             "slice_fast_cuda.linear_deltaW.weight", "slice_fast_cuda.gamma", "slice_fast_cuda.linear_clasify.weight",
             "resnet_blocks_per_down_lvl_list.0.0.conv1.norm.gn.weight"]
    eps = 1e-6
    worst = 0.0
    for name in picks:
        p = named[name]
        flat = p.data.view(-1)
        g = p.grad.view(-1)
        # the entries with the largest analytic gradient (away from the noise floor of the difference quotient)
        for i in np.argsort(-g.abs().numpy())[:2]:
            old = float(flat[i])
            flat[i] = old + eps
            with torch.no_grad():
                up = float(loss_of())
            flat[i] = old - eps
            with torch.no_grad():
                down = float(loss_of())
            flat[i] = old
            fd = (up - down) / (2 * eps)
            rel = abs(fd - float(g[i])) / max(abs(fd), 1e-12)
            worst = max(worst, rel)
            assert rel < 2e-4, f"{name}[{i}]: autograd {float(g[i]):.6e} vs central difference {fd:.6e}"
    assert worst < 2e-4


def test_oracle_lattice_matches_numpy_oracle_ops():
    """The torch restatements inside OracleLattice against the NumPy oracle functions they stand in for (fp32-level agreement)."""
    from oracle import lattice_oracle as O
    from tests.oracle_lattice import OracleLattice
    rng = np.random.default_rng(3)
    n, v, f = 400, 5, 7
    pos = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32))
    lat0 = OracleLattice([0.3] * 3, 20000)
    lat, rows, idx, w = lat0.distribute(pos, torch.zeros((n, 1), dtype=torch.float64))
    m = lat.nr_lattice_vertices()
    values = torch.from_numpy(rng.standard_normal((m, v)))
    lat.set_values(values)
    bank = torch.from_numpy(rng.standard_normal((9 * v, f)))
    nbr = O.neighbour_rows(lat.table.keys[:m], lat.table, 1, 1, 1, False)
    out = lat.convolve_im2row_standalone(bank, 1, lat, False).values()
    ref = O.conv_forward(nbr, values.numpy().astype(np.float32), bank.numpy().astype(np.float32))
    np.testing.assert_allclose(out.numpy(), ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
    g = lat.gather_standalone_with_precomputation(pos, idx, w)
    ref = O.gather_with_precomputation(values.numpy().astype(np.float32), idx.numpy(), w.numpy().astype(np.float32), n)
    np.testing.assert_allclose(g.numpy(), ref, rtol=2e-6, atol=2e-6)
    dw = torch.from_numpy(rng.standard_normal((n, 4)) * 0.1)
    lw, lb = torch.from_numpy(rng.standard_normal((f, v))), torch.from_numpy(rng.standard_normal(f))
    logits = lat.slice_classify_with_precomputation(pos, dw, lw, lb, f, idx, w)
    ref = O.slice_classify(values.numpy().astype(np.float32), dw.numpy().astype(np.float32), lw.numpy().astype(np.float32),
                           lb.numpy().astype(np.float32), idx.numpy(), w.numpy().astype(np.float32), n)
    np.testing.assert_allclose(logits.numpy(), ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
    gl = torch.from_numpy(rng.standard_normal((n, f)))
    gv, gd, gw, gb = torch.zeros_like(values), torch.zeros_like(dw), torch.zeros_like(lw), torch.zeros_like(lb)
    lat.slice_classify_backwards_with_precomputation(gl, pos, values, dw, lw, lb, f, gv, gd, gw, gb, idx, w)
    r = O.slice_classify_backwards(gl.numpy().astype(np.float32), values.numpy().astype(np.float32), dw.numpy().astype(np.float32),
                                   lw.numpy().astype(np.float32), lb.numpy().astype(np.float32), idx.numpy(), w.numpy().astype(np.float32), n)
    for a, b in zip((gv, gd, gw, gb), r):
        np.testing.assert_allclose(a.numpy(), b, rtol=2e-5, atol=2e-5 * np.abs(b).max())
    src = torch.from_numpy(rng.standard_normal((4 * n, 3)))
    mx, arg = lat.scatter_max(src, idx)
    rmx, rarg = O.scatter_max(src.numpy().astype(np.float32), idx.numpy(), m)
    np.testing.assert_allclose(mx.numpy(), rmx, rtol=1e-6, atol=1e-6)
    assert np.array_equal(arg.numpy(), rarg)
