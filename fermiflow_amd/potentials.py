"""Potentials (src/potentials.py): harmonic trap and pairwise Coulomb, evaluated by ff_potential."""
import torch

from . import native


class SPPotential(object):
    pass


class HO(SPPotential):
    def V(self, x):
        return native.potential(x.contiguous(), 0.0, True)


class PairPotential(object):
    def rij(self, x):
        n = x.shape[-2]
        row, col = torch.triu_indices(n, n, offset=1)
        return (x[:, :, None] - x[:, None])[:, row, col, :].norm(dim=-1)

    def V(self, x):
        return self.v(self.rij(x)).sum(dim=-1)


class CoulombPairPotential(PairPotential):
    def __init__(self, Z):
        super(CoulombPairPotential, self).__init__()
        self.Z = Z

    def v(self, rij):
        return self.Z / rij

    def V(self, x):
        return native.potential(x.contiguous(), self.Z, False)
