"""Alias: the IoU bookkeeping of lattice_net_amd.losses."""
from lattice_net_amd.losses import Scores  # noqa: F401
