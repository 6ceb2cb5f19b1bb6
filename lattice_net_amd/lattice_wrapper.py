"""`LatticeWrapper`: lets a `Lattice` travel through `torch.autograd.Function.forward` return values
(reference latticenet_py/lattice/lattice_wrapper.py:12-17)."""
import torch


class LatticeWrapper(torch.Tensor):
    @staticmethod
    def wrap(lattice):
        ls_wrap = LatticeWrapper()
        ls_wrap.lattice = lattice
        return ls_wrap
