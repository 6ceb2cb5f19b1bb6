"""Alias: GeneralizedSoftDiceLoss of lattice_net_amd.losses."""
from lattice_net_amd.losses import GeneralizedSoftDiceLoss  # noqa: F401
