"""cfg reader for the `model:` block and the LNN assembly (channel bookkeeping and state_dict key
names as in reference latticenet_py/lattice/models.py:70-197).  CPU only: no kernels run here."""
import textwrap

import numpy as np

import pytest
import torch

from lattice_net_amd.model_params import ModelParams, read_cfg_block

KITTI_TINY = textwrap.dedent("""
    core: { loguru_verbosity: 3 }
    model: {
        //SHOULD BE USED WITH A SIGMA OF 0.6
        // pointnet_layers: [16,32,64]
        positions_mode: "xyz"
        values_mode: "none"
        pointnet_layers: [16,32]
        pointnet_start_nr_channels: 32
        nr_downsamples: 2
        nr_blocks_down_stage: [1,1,1]
        nr_blocks_bottleneck: 1
        nr_blocks_up_stage: [1,1,1]
        nr_levels_down_with_normal_resnet: 3
        nr_levels_up_with_normal_resnet: 3
        compression_factor: 1.0
        dropout_last_layer: 0.0
        experiment: "none" // a comment with a "quote"
    }
    lattice_gpu: {
        hash_table_capacity: 100000 //good for kitti
        nr_sigmas: 1
        sigma_0: "0.9 3" //sigma of X affecting Y dimensions
    }
""")


@pytest.fixture
def cfg_path(tmp_path):
    p = tmp_path / "net.cfg"
    p.write_text(KITTI_TINY)
    return str(p)


def test_model_block_is_parsed_with_both_pointnet_spellings(cfg_path, tmp_path):
    mp = ModelParams.create(cfg_path)
    assert mp.positions_mode() == "xyz" and mp.values_mode() == "none"
    assert mp.pointnet_channels_per_layer() == [16, 32]
    assert mp.pointnet_start_nr_channels() == 32
    assert mp.nr_downsamples() == 2
    assert mp.nr_blocks_down_stage() == [1, 1, 1] and mp.nr_blocks_up_stage() == [1, 1, 1]
    assert mp.nr_blocks_bottleneck() == 1
    assert mp.nr_levels_down_with_normal_resnet() == 3 and mp.nr_levels_up_with_normal_resnet() == 3
    assert mp.compression_factor() == 1.0 and mp.dropout_last_layer() == 0.0
    p2 = tmp_path / "other.cfg"
    p2.write_text(KITTI_TINY.replace("pointnet_layers: [16,32]", "pointnet_channels_per_layer: [8, 16, 24]"))
    assert ModelParams.create(str(p2)).pointnet_channels_per_layer() == [8, 16, 24]


def test_cfg_reader_handles_comments_strings_and_missing_blocks(cfg_path):
    lat = read_cfg_block(cfg_path, "lattice_gpu")
    assert lat == {"hash_table_capacity": 100000, "nr_sigmas": 1, "sigma_0": "0.9 3"}
    with pytest.raises(ValueError, match="no `train` block"):
        read_cfg_block(cfg_path, "train")
    with pytest.raises(KeyError, match="values_mode"):
        import os
        bad = os.path.join(os.path.dirname(cfg_path), "bad.cfg")
        open(bad, "w").write(KITTI_TINY.replace('values_mode: "none"', ""))
        ModelParams.create(bad)


def test_lattice_create_reads_the_same_file(cfg_path):
    from lattice_net_amd import Lattice
    lat = Lattice.create(cfg_path, "lattice")
    assert lat.capacity() == 100000 and lat.name() == "lattice" and list(lat.sigmas_tensor().cpu().numpy()) == [pytest.approx(0.9)] * 3
    assert Lattice.get_expected_filter_extent(1) == 9


def test_lnn_assembly_channels_and_state_dict_names(cfg_path):
    from lattice_net_amd.models import LNN
    torch.manual_seed(0)
    net = LNN(20, ModelParams.create(cfg_path), device="cpu")
    # 32 -> coarsen 64 -> coarsen 128 | finefy 64 (+64 skip = 128) | finefy 64 (+32 skip = 96) -> head on 96 channels
    assert [c.coarse.weight.shape for c in net.coarsens_list] == [(9 * 32, 64), (9 * 64, 128)]
    assert [f.fine.weight.shape for f in net.finefy_list] == [(9 * 128, 64), (9 * 128, 64)]
    assert net.slice_fast_cuda.in_channels == 96
    assert [s.linear.weight.shape for s in net.slice_fast_cuda.stepdown] == [(96, 96), (48, 96)]
    assert net.slice_fast_cuda.bottleneck.linear.weight.shape == (8, 48)
    assert net.slice_fast_cuda.linear_clasify.weight.shape == (20, 96)
    assert net.slice_fast_cuda.linear_deltaW.weight.shape == (1, 9)
    # group norm: 32 groups when divisible, else C/2 (mods:585-599)
    assert net.slice_fast_cuda.stepdown[1].norm.gn.num_groups == 32 and net.slice_fast_cuda.bottleneck.norm.gn.num_groups == 24
    keys = set(net.state_dict().keys())
    for k in ["point_net.layers.0.weight_g", "point_net.layers.0.weight_v", "point_net.layers.1.bias",
              "point_net.last_conv.weight_g", "point_net.last_conv.weight_v", "point_net.last_conv.bias",
              "resnet_blocks_per_down_lvl_list.0.0.conv1.norm.gn.weight", "resnet_blocks_per_down_lvl_list.0.0.conv1.conv.weight",
              "resnet_blocks_per_down_lvl_list.1.0.conv2.conv.weight", "coarsens_list.1.coarse.weight",
              "resnet_blocks_bottleneck.0.contract.linear.weight", "resnet_blocks_bottleneck.0.conv.conv.weight",
              "resnet_blocks_bottleneck.0.expand.norm.gn.bias", "finefy_list.0.norm.gn.weight", "finefy_list.1.fine.weight",
              "resnet_blocks_per_up_lvl_list.1.0.conv2.conv.bias", "slice_fast_cuda.stepdown.0.linear.weight",
              "slice_fast_cuda.bottleneck.norm.gn.weight", "slice_fast_cuda.linear_deltaW.bias", "slice_fast_cuda.gamma",
              "slice_fast_cuda.beta", "slice_fast_cuda.linear_clasify.weight"]:
        assert k in keys, k
    assert net.point_net.layers[0].weight_v.shape == (16, 4) and net.point_net.layers[0].weight_g.shape == (16, 1)  # xyz + 1 dummy value
    assert net.point_net.last_conv.weight_v.shape == (9 * 64, 32) and net.point_net.last_conv.weight_g.shape == (1, 32)
    # only the last convolution of the decoder carries a bias (models.py:176)
    assert "resnet_blocks_per_up_lvl_list.0.0.conv2.conv.bias" not in keys
    assert "resnet_blocks_per_down_lvl_list.0.0.conv1.conv.bias" not in keys


def test_blocks_reject_non_matrix_values():
    from lattice_net_amd.lattice_blocks import DropoutLattice, GroupNormLatticeModule
    with pytest.raises(ValueError):
        DropoutLattice(0.1)(torch.zeros(3))
    with pytest.raises(ValueError):
        GroupNormLatticeModule(8, device="cpu")(torch.zeros(2, 3, 8), None)
    gn = GroupNormLatticeModule(8, device="cpu")
    with torch.no_grad():
        gn.gn.weight.uniform_(0.5, 2.0)
        gn.gn.bias.uniform_(-1.0, 1.0)
    x = (torch.randn(50, 8) * 3 + 1).requires_grad_(True)
    y, _ = gn(x, None, do_set_values=False)
    ref = torch.nn.functional.group_norm(x.t().unsqueeze(0), gn.gn.num_groups, gn.gn.weight, gn.gn.bias).squeeze(0).t()
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5)
    gy = torch.randn_like(y)
    g1 = torch.autograd.grad(y, [x, gn.gn.weight, gn.gn.bias], gy, retain_graph=True)
    g2 = torch.autograd.grad(ref, [x, gn.gn.weight, gn.gn.bias], gy)
    for a, b in zip(g1, g2):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)


def test_weight_norm_layers_follow_the_reference_parametrisation():
    """weight = weight_v * weight_g / ||weight_v||_F, g per output unit, initialised to ||v|| (utils.py:72-158, 291)."""
    from lattice_net_amd.lattice_modules import LinearWN
    torch.manual_seed(0)
    lin = LinearWN(5, 7)
    assert torch.allclose(lin.weight, lin.weight_v)  # g == ||v|| at construction
    with torch.no_grad():
        lin.weight_g.mul_(torch.linspace(0.5, 2.0, 7).unsqueeze(1))
    x = torch.randn(11, 5)
    w = lin.weight_v * (lin.weight_g / lin.weight_v.norm())
    assert torch.allclose(lin(x), x @ w.t() + lin.bias)
    lin(x).sum().backward()
    assert lin.weight_g.grad is not None and lin.weight_v.grad is not None
    # a checkpoint written by the reference (torch WeightNorm parameter names) loads with strict key matching
    sd = {"weight_g": torch.ones(7, 1), "weight_v": torch.randn(7, 5), "bias": torch.zeros(7)}
    lin.load_state_dict(sd, strict=True)


def test_reference_style_checkpoint_round_trip(cfg_path, tmp_path):
    """a state_dict saved from one LNN loads into a freshly built one before any forward pass."""
    from lattice_net_amd.models import LNN
    torch.manual_seed(1)
    a = LNN(20, ModelParams.create(cfg_path), device="cpu")
    path = tmp_path / "model_e_1.pt"
    torch.save(a.state_dict(), path)
    torch.manual_seed(2)
    b = LNN(20, ModelParams.create(cfg_path), device="cpu")
    missing, unexpected = b.load_state_dict(torch.load(path), strict=True)
    assert not missing and not unexpected
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)


def test_state_dict_key_space_is_the_reference_classes_attribute_names(cfg_path):
    """every component of every state_dict key of the assembled LNN is a name the reference's class of the same name
    assigns on `self` (tests/golden/reference_attribute_names.json, extracted from the reference's source by
    tests/golden/make_reference_attribute_names.py), a ModuleList index, or a parameter name of a stock torch module.  A
    checkpoint written by the reference addresses its tensors by exactly these paths."""
    import json
    import os
    from lattice_net_amd.models import LNN
    names = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_attribute_names.json")))
    torch_params = {"Linear": {"weight", "bias"}, "GroupNorm": {"weight", "bias"}}

    def allowed_names(cls):
        e = names[cls]
        if "weight_norm_of" in e:  # X = weight_norm_wrapper(Base): torch's WeightNorm replaces <name> by <name>_g / <name>_v
            base = e["weight_norm_of"]
            inner = allowed_names(base) if base in names else set(torch_params[base])
            return (inner - {e["weight_norm_name"]}) | {e["weight_norm_name"] + "_g", e["weight_norm_name"] + "_v"}
        return set(e["self_attributes"])

    torch.manual_seed(0)
    net = LNN(20, ModelParams.create(cfg_path), device="cpu")
    checked = set()
    for path, mod in net.named_modules():
        cls = type(mod).__name__
        mine = set(mod._modules) | set(mod._parameters) | set(mod._buffers)
        if cls not in names:
            # containers and stock torch layers only
            assert cls in ("ModuleList", "Sequential", "GroupNorm", "Linear", "Dropout", "ReLU", "LeakyReLU", "Tanh", "LogSoftmax", "GELU",
                           "Identity"), f"{path}: class {cls} does not exist in the reference's lattice_modules.py / models.py / utils.py"
            assert mine <= torch_params.get(cls, set()) or cls in ("ModuleList", "Sequential")
            continue
        allowed = allowed_names(cls)
        assert mine <= allowed, f"{path} ({cls}): {sorted(mine - allowed)} are not attributes of the reference's {cls}"
        checked.add(cls)
    # the network is made of the reference's classes, weight-normalised ones included
    assert {"LNN", "PointNetModule", "LinearWN", "ConvLatticeIm2RowWNModule", "CoarsenAct", "GnReluFinefy", "ResnetBlock", "BottleneckBlock",
            "GnReluConv", "GnRelu1x1", "GroupNormLatticeModule", "SliceFastCUDALatticeModule"} <= checked, sorted(checked)
    sd = net.state_dict()
    for path, mod in net.named_modules():  # tensors the reference's lattice operators own directly (created lazily there, eagerly here)
        if type(mod).__name__ in ("ConvLatticeModule", "ConvLatticeIm2RowModule", "CoarsenLatticeModule", "FinefyLatticeModule"):
            assert f"{path}.weight" in sd


def test_every_class_of_the_reference_module_files_exists_under_the_same_name():
    """The fixture's class list (reference lattice_modules.py incl. its weight-normalised aliases, models.py) against the reference's
    import paths: `from latticenet_py.lattice.lattice_modules import *` finds every name (operator modules live in
    lattice_net_amd.lattice_modules, network blocks in lattice_net_amd.lattice_blocks; the alias module joins them as the reference does)."""
    import json
    import os
    import latticenet_py.lattice.lattice_modules as RM
    import latticenet_py.lattice.models as RMD
    import lattice_net_amd.lattice_modules as M
    names = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_attribute_names.json")))
    seen = 0
    for cls, e in names.items():
        if e["file"] == "lattice_modules.py":
            assert hasattr(RM, cls), f"latticenet_py.lattice.lattice_modules.{cls} is missing"
            seen += 1
        elif e["file"] == "models.py":
            assert hasattr(RMD, cls), f"latticenet_py.lattice.models.{cls} is missing"
            seen += 1
    assert seen >= 40
    assert RM.ConvLatticeIm2RowWNModule is M.ConvLatticeIm2RowWNModule and RM.PointNetModule is M.PointNetModule


def _reference_fixture():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "F10_reference_lnn.npz"))


def _oracle_cfg_params():
    import tempfile
    from tests.test_oracle_network import CFG
    with tempfile.NamedTemporaryFile("w", suffix=".cfg") as f:
        f.write(CFG)
        f.flush()
        return ModelParams.create(f.name)


def test_state_dict_keys_equal_the_reference_networks_in_order_with_equal_shapes():
    """SURVEY 8f-3 against the reference's OWN Python: tests/golden/F10_reference_lnn.npz holds the ordered state_dict keys and
    shapes of /root/reference/latticenet_py/lattice/models.py `LNN` after its first forward (lazy parameters created,
    lattice_modules.py:509-516, 554-556, 636-651; ln_eval.py:131-137), recorded by executing that code over CPU stand-ins
    (tests/golden/make_reference_network_fixture.py).  The assembled LNN must expose exactly that list — same names, same ORDER
    (optimizer state of a reference checkpoint is addressed by parameter position), same shapes — before any forward."""
    from lattice_net_amd.models import LNN
    from tests.oracle_lattice import OracleLattice
    fx = _reference_fixture()
    OracleLattice([0.08] * 3, 60000)  # sets the static lattice dimension the modules size their banks with (Lattice.cu:44)
    net = LNN(int(fx["nr_classes"]), _oracle_cfg_params(), device="cpu")
    sd = net.state_dict()
    ref_keys = [str(k) for k in fx["keys"]]
    assert list(sd.keys()) == ref_keys
    assert [",".join(map(str, sd[k].shape)) for k in ref_keys] == [str(s) for s in fx["shapes"]]
    named = dict(net.named_parameters())
    assert [k in named for k in ref_keys] == [bool(b) for b in fx["is_parameter"]]
    # what the reference creates lazily exists here from construction: a reference checkpoint loads before any forward
    lazy = set(ref_keys) - {str(k) for k in fx["keys_at_construction"]}
    assert lazy == {k for k in ref_keys if k.startswith("point_net.layers.") or k.split(".")[1] in ("gamma", "beta", "linear_deltaW", "linear_clasify")}


def test_network_definition_reproduces_the_reference_networks_logits_and_gradients():
    """The LNN definition of this package on the oracle lattice in float64 against the reference's own `LNN` run on the same lattice
    stand-in with the same seeded parameters (F10): logits, loss and every parameter gradient to 1e-9 — a transposed skip concat, a
    norm in the wrong place or a different block order relative to models.py:96-266 / lattice_modules.py shows up here."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_reference_network_fixture import gradient_sample_index, seeded_parameter
    from lattice_net_amd.models import LNN
    from lattice_net_amd.synthetic import box_surface_cloud
    from tests.oracle_lattice import OracleLattice
    fx = _reference_fixture()
    n, c = int(fx["n_points"]), int(fx["nr_classes"])
    lattice = OracleLattice([0.08] * 3, 60000)
    net = LNN(c, _oracle_cfg_params(), device="cpu").double()
    sd = net.state_dict()
    for i, k in enumerate(str(k) for k in fx["keys"]):
        sd[k].copy_(torch.from_numpy(seeded_parameter(i, k, sd[k].shape, int(fx["param_seed"]))))
    pos = torch.from_numpy(box_surface_cloud(n, int(fx["cloud_seed"])))
    target = torch.from_numpy(np.random.default_rng(int(fx["cloud_seed"])).integers(0, c, n))
    logsoftmax, logits = net(lattice, pos, torch.zeros((n, 1), dtype=torch.float64))
    loss = torch.nn.functional.nll_loss(logsoftmax, target)
    loss.backward()
    assert np.abs(logits.detach().numpy() - fx["logits"]).max() <= 1e-9 * np.abs(fx["logits"]).max()
    assert abs(loss.item() - float(fx["loss"])) <= 1e-12
    named = dict(net.named_parameters())
    for i, k in enumerate(str(k) for k in fx["keys"]):
        if k not in named:
            continue
        g = named[k].grad.numpy().reshape(-1)
        if f"grad_full/{i}" in fx:
            ref = fx[f"grad_full/{i}"]
        else:
            ref, g = fx[f"grad_sample/{i}"], g[gradient_sample_index(g.size)]
        assert np.abs(g - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-6), k
        assert abs(np.linalg.norm(named[k].grad.numpy()) - fx["grad_norms"][i]) <= 1e-9 * max(fx["grad_norms"][i], 1e-6), k


KITTI_CFG = """
model: {
    positions_mode: "xyz"
    values_mode: "none"
    pointnet_layers: [16,32]
    pointnet_start_nr_channels: 32
    nr_downsamples: 2
    nr_blocks_down_stage: [1,1,1]
    nr_blocks_bottleneck: 1
    nr_blocks_up_stage: [1,1,1]
    nr_levels_down_with_normal_resnet: 3
    nr_levels_up_with_normal_resnet: 3
    compression_factor: 1.0
    dropout_last_layer: 0.0
}
lattice_gpu: {
    hash_table_capacity: 100000
    nr_sigmas: 1
    sigma_0: "0.9 3"
}
"""  # the model and lattice blocks of config/lnn_train_semantic_kitti.cfg:36-47,62-69


def kitti_fixture_case(device, dtype):
    """(fixture F12, network with the fixture's seeded parameters, positions, target): BASELINE.json configs[2] at its own size."""
    import os
    import sys
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_reference_network_fixture import seeded_parameter
    from lattice_net_amd import Lattice
    from lattice_net_amd.models import LNN
    from lattice_net_amd.synthetic import lidar_cloud
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "F12_reference_lnn_kitti.npz"))
    n, c = int(fx["n_points"]), int(fx["nr_classes"])
    with tempfile.NamedTemporaryFile("w", suffix=".cfg") as f:
        f.write(KITTI_CFG)
        f.flush()
        mp = ModelParams.create(f.name)
        cfg_lattice = Lattice.create(f.name, "lattice")  # (also sets the static lattice dimension the modules size their banks with)
    net = LNN(c, mp, device=device).to(dtype)
    sd = net.state_dict()
    keys = [str(k) for k in fx["keys"]]
    assert list(sd.keys()) == keys and [",".join(map(str, sd[k].shape)) for k in keys] == [str(x) for x in fx["shapes"]]
    for i, k in enumerate(keys):
        sd[k].copy_(torch.from_numpy(seeded_parameter(i, k, sd[k].shape, int(fx["param_seed"]))).to(dtype))
    pos = torch.from_numpy(lidar_cloud(n, int(fx["cloud_seed"])))
    target = torch.from_numpy(np.random.default_rng(int(fx["cloud_seed"])).integers(0, c, n))
    return fx, net, cfg_lattice, pos, target


def compare_with_network_fixture(fx, net, logits, loss, tol, floor_rel):
    """Logits of the fixture's point sample, loss, every parameter gradient (full or the fixed strided sample) and gradient norm."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_reference_network_fixture import gradient_sample_index, logits_sample_index
    lg = logits.detach().cpu().double().numpy()
    if lg.shape[0] != fx["logits"].shape[0]:
        lg = lg[logits_sample_index(lg.shape[0], fx["logits"].shape[0])]
    assert np.abs(lg - fx["logits"]).max() <= tol * np.abs(fx["logits"]).max(), float(np.abs(lg - fx["logits"]).max() / np.abs(fx["logits"]).max())
    assert abs(float(loss) - float(fx["loss"])) <= tol * abs(float(fx["loss"]))
    named = dict(net.named_parameters())
    gmax = float(np.nanmax(fx["grad_norms"]))
    bad = {}
    for i, k in enumerate(str(k) for k in fx["keys"]):
        if k not in named:
            continue
        g = named[k].grad.detach().cpu().double().numpy().reshape(-1)
        norm = float(np.linalg.norm(g))
        if f"grad_full/{i}" in fx:
            ref = fx[f"grad_full/{i}"]
        else:
            ref, g = fx[f"grad_sample/{i}"], g[gradient_sample_index(g.size)]
        # (floor: tensors whose gradient vanishes analytically — biases in front of a normalisation — hold rounding noise only)
        e = float(np.abs(g - ref).max()) / max(float(np.abs(ref).max()), floor_rel * gmax / np.sqrt(max(ref.size, 1)))
        en = abs(norm - float(fx["grad_norms"][i])) / max(float(fx["grad_norms"][i]), floor_rel * gmax)
        if max(e, en) > tol:
            bad[k] = (e, en)
    assert not bad, f"parameter gradients off: {sorted(bad.items(), key=lambda kv: -max(kv[1]))[:8]}"


def test_network_definition_reproduces_the_reference_network_at_semantic_kitti_size():
    """F12 = the reference's own `LNN` (models.py:70-266) with the model block of config/lnn_train_semantic_kitti.cfg:36-47 executed in
    float64 over the oracle lattice on the 120 000-point scan of BASELINE.json configs[2] (46.5 k / 11.4 k / 2.6 k vertices).  This
    package's network on the same oracle lattice, float64, same seeded parameters: logits of the fixture's 4096-point sample, loss
    and every parameter gradient to 1e-9."""
    from tests.oracle_lattice import OracleLattice
    fx, net, _, pos, target = kitti_fixture_case("cpu", torch.float64)
    lattice = OracleLattice([0.9] * 3, 100000)
    logsoftmax, logits = net(lattice, pos, torch.zeros((pos.shape[0], 1), dtype=torch.float64))
    loss = torch.nn.functional.nll_loss(logsoftmax, target)
    loss.backward()
    assert abs(float(logits.detach().abs().sum()) - float(fx["logits_checksum"])) <= 1e-9 * float(fx["logits_checksum"])
    compare_with_network_fixture(fx, net, logits, loss, tol=1e-9, floor_rel=1e-6)
