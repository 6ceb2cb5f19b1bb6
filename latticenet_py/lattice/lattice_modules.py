"""Alias of the reference's lattice_modules.py: operator modules (lattice_net_amd.lattice_modules) and the network blocks
(lattice_net_amd.lattice_blocks) under one name, as there."""
from lattice_net_amd.lattice_modules import *  # noqa: F401,F403
from lattice_net_amd.lattice_blocks import *  # noqa: F401,F403
