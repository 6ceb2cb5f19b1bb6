"""Alias: LatticeWrapper lives next to the autograd Functions in lattice_net_amd.lattice_funcs."""
from lattice_net_amd.lattice_funcs import LatticeWrapper  # noqa: F401
