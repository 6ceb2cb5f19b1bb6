"""Import aliases for code written against the reference's Python package layout (`latticenet_py.lattice.*`,
`latticenet_py.callbacks.scores`): every name resolves to the MI355X backend in `lattice_net_amd`."""
