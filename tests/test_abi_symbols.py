"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every function the
public header declares, and rejects bad arguments without touching a GPU."""
import ctypes as C
import os
import re

import pytest

import lattice_net_amd
from lattice_net_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "latticenet_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ln_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in latticenet_hip.h but not exported by the library"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)


def test_struct_layout_matches_header():
    # int, int, 8 pointers, int, int, int (+4 padding), pointer, int, int, int (+4 padding), pointer -> 120 bytes on LP64
    assert C.sizeof(_lib.LnTable) == 120
    assert [f[0] for f in _lib.LnTable._fields_] == ["capacity", "pos_dim", "slot_keys", "slot_tok", "slot_cnt", "entries", "keys",
                                                     "nr_filled", "status", "host_counters", "host_seq", "key_format", "row_limit",
                                                     "slot_map", "bucket_slots_max", "batch_points", "batch_key_step", "row_regions"]
    assert _lib.LnTable.slot_map.offset == 88 and _lib.LnTable.row_regions.offset == 112
    assert C.sizeof(_lib.LnCsr) == 56  # 4 pointers + seg_region + planes + the dense hint (int, padded)


def test_version_and_kernel_names():
    lib = _lib.load()
    assert b"gfx950" in lib.ln_version()
    names = lib.ln_kernel_names().decode().split(",")
    assert "k_conv_mfma" in names and "k_insert_points" in names and "k_csr_reduce_segments" in names


def test_argument_errors_are_reported_not_fatal():
    lib = _lib.load()
    assert lib.ln_conv_forward(None, None, None, -1, 9, 32, 32, 0, None, None) == -1
    assert b"bad sizes" in lib.ln_last_error_string()
    assert lib.ln_neighbours(None, 10, None, 1, 1, 1, 0, None, None) == -1
    assert b"null table" in lib.ln_last_error_string()
    t = _lib.LnTable(100, 9, 1, 1, 1, 1, 1, 1, 1)  # pos_dim 9 is unsupported
    assert lib.ln_table_clear(C.byref(t), None, 0, None) == -2
    with pytest.raises(_lib.LatticeNetHipError, match="unsupported"):
        _lib.check(-2, "ln_table_clear")


def test_workspace_queries_are_pure_host_functions():
    lib = _lib.load()
    assert lib.ln_build_workspace_bytes(480000, 100000) >= 2 * 480000 * 4
    assert lib.ln_csr_workspace_bytes(480000, 100000) >= 480000 * 4 + 3 * 100000 * 4
    assert lib.ln_csr_max_segments(480000, 100000) == 100000 + 480000 // 16 + 1
    assert lib.ln_conv_grad_filter_workspace_bytes(46538, 9, 32, 32) >= 91 * 9 * 32 * 32 * 4  # one slab per 512 vertices


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/liblatticenet_hip.so")
    with pytest.raises(_lib.LatticeNetHipError, match="no CPU fallback"):
        _lib.load()


def test_reference_module_alias():
    import latticenet
    assert latticenet.Lattice is lattice_net_amd.Lattice and latticenet.HashTable is lattice_net_amd.HashTable


def test_kernel_name_list_matches_the_launch_sites():
    """ln_kernel_names() is what ln_profile_begin accepts: it must list exactly the names the LN_LAUNCH sites use."""
    import glob
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lattice_net_amd", "csrc")
    used = set()
    for path in glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h")):
        with open(path) as f:
            used |= set(re.findall(r'LN_LAUNCH\("([a-z0-9_]+)"', f.read()))
    listed = set(_lib.load().ln_kernel_names().decode().split(","))
    assert listed == used, (sorted(used - listed), sorted(listed - used))


def test_reference_python_package_aliases():
    """Code written against the reference's Python layout imports from latticenet_py.lattice.* / latticenet_py.callbacks.scores;
    every class and Function named there must resolve to this backend."""
    import importlib
    mods = importlib.import_module("latticenet_py.lattice.lattice_modules")
    for name in """DropoutLattice SplatLatticeModule DistributeLatticeModule ExpandLatticeModule ConvLatticeModule ConvLatticeIm2RowModule
                   CoarsenLatticeModule FinefyLatticeModule SliceLatticeModule GatherLatticeModule SliceFastCUDALatticeModule
                   BatchNormLatticeModule GroupNormLatticeModule PointNetModule Conv1x1WN Conv1x1WNAct Conv1x1 GnRelu1x1 GnGelu1x1 Gn
                   GnReluDepthwiseConv ConvAct GnReluConv GnGeluConv BnReluConv CoarsenAct GnCoarsen GnReluCoarsen GnGeluCoarsen FinefyAct
                   GnReluFinefy GnGeluFinefy GnFinefy TwoConv ResnetBlock ResnetBlock2 BottleneckBlock DensenetBlock""".split():
        assert hasattr(mods, name), f"latticenet_py.lattice.lattice_modules.{name} missing"
    funcs = importlib.import_module("latticenet_py.lattice.lattice_funcs")
    for name in """SplatLattice DistributeLattice ExpandLattice Im2RowIndicesLattice Im2RowLattice ConvIm2RowLattice CoarsenLattice
                   FinefyLattice SliceLattice SliceClassifyLattice GatherLattice""".split():
        assert hasattr(funcs, name), f"latticenet_py.lattice.lattice_funcs.{name} missing"
    from latticenet_py.callbacks.scores import Scores  # noqa: F401
    from latticenet_py.lattice.diceloss import GeneralizedSoftDiceLoss  # noqa: F401
    from latticenet_py.lattice.lattice_wrapper import LatticeWrapper  # noqa: F401
    from latticenet_py.lattice.lovasz_loss import LovaszSoftmax  # noqa: F401
    from latticenet_py.lattice.models import LNN, prepare_cloud  # noqa: F401


def test_workspace_size_functions_are_total_over_odd_shapes():
    """Host-only entry points (no GPU call): every *_workspace_bytes function returns a positive size for any plausible shape —
    filter extents that are not 9, widths that are not multiples of anything, empty lattices (a division by a zero workgroup count
    in the block-shape heuristics of the bf16x3 filter gradient once killed the process for filter_extent < 3)."""
    import ctypes as C
    import itertools
    from lattice_net_amd import _lib
    lib = _lib.load()
    for m, e, v, f in itertools.product([0, 1, 100, 4096, 46538, 5000000], [1, 3, 5, 7, 9, 15], [1, 4, 8, 32, 64, 96, 128, 1024], [1, 6, 16, 32, 96, 128]):
        assert lib.ln_conv_grad_filter_workspace_bytes(m, e, v, f) >= 256
        assert lib.ln_conv_forward_workspace_bytes(m, e, v, f) >= 0
    for n, d, v, c in itertools.product([0, 1, 1000, 120000], [1, 2, 3, 6], [1, 8, 64, 96], [1, 20, 50]):
        assert lib.ln_slice_classify_backward_workspace_bytes(n, d, v, c) > 0
