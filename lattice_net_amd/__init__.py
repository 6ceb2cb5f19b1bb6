"""lattice_net_amd — MI355X (gfx950) native permutohedral-lattice backend.

Drop-in for the hot path of AIS-Bonn/lattice_net: the `latticenet.Lattice` operator object and the
`torch.autograd.Function`s built on it.  Compute is hand-written HIP behind a C ABI
(include/latticenet_hip.h, lattice_net_amd/csrc); there is no CPU fallback.
"""
from ._lib import LatticeNetHipError, LIB_PATH, load as load_library  # noqa: F401
from .lattice import HashTable, Lattice, get_row_order, set_hash_capacity_policy, set_row_order  # noqa: F401
from .lattice_wrapper import LatticeWrapper  # noqa: F401
from .lattice_funcs import *  # noqa: F401,F403
from .model_params import ModelParams  # noqa: F401
from .capture import CapturedNetworkStep, CapturedStep  # noqa: F401

__version__ = "0.1.0"
