"""Alias of lattice_net_amd.models (LNN, prepare_cloud)."""
from lattice_net_amd.models import LNN, prepare_cloud  # noqa: F401
