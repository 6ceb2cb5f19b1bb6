"""Alias: LovaszSoftmax of lattice_net_amd.losses."""
from lattice_net_amd.losses import LovaszSoftmax  # noqa: F401
