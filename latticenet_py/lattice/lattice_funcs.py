"""Alias of lattice_net_amd.lattice_funcs (the autograd Functions of the reference's lattice_funcs.py)."""
from lattice_net_amd.lattice_funcs import *  # noqa: F401,F403
from lattice_net_amd.lattice_funcs import LatticeWrapper  # noqa: F401
