"""Import alias so reference-style code (`from latticenet import Lattice, HashTable`,
src/PyBridge.cxx:27) resolves to the MI355X backend."""
from lattice_net_amd.lattice import HashTable, Lattice  # noqa: F401
from lattice_net_amd.model_params import ModelParams  # noqa: F401  (src/PyBridge.cxx:139-152)
