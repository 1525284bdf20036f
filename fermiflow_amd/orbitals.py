"""Single-particle orbitals and many-body state enumeration (host-side set-up).

Mirrors `HO2D` / `Orbitals` of the reference (src/orbitals.py:6-99).  Orbitals cannot cross a C ABI as Python
closures, so an orbital here is a small object carrying its index k into `HO2D().orbitals`
(k <-> (nx, shell - nx), src/orbitals.py:81); the kernels evaluate it from that index.  Calling the object
evaluates the same closed form with torch ops (API compatibility for user code; the hot path never does).
"""
import math

import torch

_HERM = [  # h_n(x) = norm * sum_k c_k x^k  (src/orbitals.py:66-73)
    (1.0, [1.0]),
    (math.sqrt(2.0), [0.0, 1.0]),
    (1 / math.sqrt(2.0), [-1.0, 0.0, 2.0]),
    (1 / math.sqrt(3.0), [0.0, -3.0, 0.0, 2.0]),
    (1 / math.sqrt(6.0), [1.5, 0.0, -6.0, 0.0, 2.0]),
    (1 / math.sqrt(15.0), [0.0, 7.5, 0.0, -10.0, 0.0, 2.0]),
    (1 / math.sqrt(5.0), [-1.25, 0.0, 7.5, 0.0, -5.0, 0.0, 2.0 / 3.0]),
    (1 / math.sqrt(70.0), [0.0, -17.5, 0.0, 35.0, 0.0, -14.0, 0.0, 4.0 / 3.0]),
]


class Orbital:
    """phi_{nx,ny}(r) = pi^-1/2 exp(-|r|^2/2) h_nx(x) h_ny(y)."""

    __slots__ = ("k", "nx", "ny", "E")

    def __init__(self, k, nx, ny):
        self.k, self.nx, self.ny, self.E = k, nx, ny, nx + ny + 1

    @staticmethod
    def _h(n, x):
        norm, c = _HERM[n]
        acc = torch.zeros_like(x)
        for ck in reversed(c):
            acc = acc * x + ck
        return norm * acc

    def __call__(self, x):
        return (1.0 / math.sqrt(math.pi)) * torch.exp(-0.5 * (x ** 2).sum(dim=-1)) \
            * self._h(self.nx, x[..., 0]) * self._h(self.ny, x[..., 1])

    def __repr__(self):
        return f"Orbital(k={self.k}, nx={self.nx}, ny={self.ny})"


def orbital_indices(orbitals):
    """tuple of Orbital objects -> list of integer indices for the kernels."""
    out = []
    for o in orbitals:
        if not isinstance(o, Orbital):
            raise TypeError("fermiflow_amd orbitals must come from fermiflow_amd.orbitals.HO2D().orbitals "
                            "(Python closures cannot be evaluated by the HIP kernels)")
        out.append(o.k)
    return out


class Orbitals(object):
    def fermion_states_random(self, n):
        import random
        orbitals, Es = zip(*random.sample(tuple(zip(self.orbitals, self.Es)), k=n))
        return orbitals, Es

    def subsets(self, k, Pmax, Ps):
        """All index subsets of length k with total price <= Pmax, ordered by total price (ties keep
        lexicographic order) -- same result and order as src/orbitals.py:14-31."""
        n = len(Ps)
        found = []

        def grow(prefix, total, start):
            need = k - len(prefix)
            if need == 0:
                found.append((tuple(prefix), total))
                return
            idx = start
            while idx + need - 1 < n:
                if sum(Ps[idx:idx + need]) <= Pmax - total:   # cheapest completion still affordable
                    prefix.append(idx)
                    grow(prefix, total + Ps[idx], idx + 1)
                    prefix.pop()
                idx += 1
        grow([], 0, 0)
        found.sort(key=lambda t: t[1])
        indices, totals = zip(*found)
        return indices, totals

    def fermion_states(self, nup, ndown, deltaE):
        if ndown != 0:
            raise ValueError("Only the polarized case (i.e., ndown = 0) is allowed "
                             "in the present implementation.")
        E0 = sum(self.Es[:nup])
        indices, Es = self.subsets(nup, E0 + deltaE, self.Es)
        states = tuple((tuple(self.orbitals[i] for i in subset), ()) for subset in indices)
        return states, Es


class HO2D(Orbitals):
    """Orbitals of h = -1/2 laplacian + 1/2 r^2 in 2-D; 36 of them (shells 0..7), E = shell + 1."""

    def __init__(self):
        self.orbitals, self.Es = [], []
        for shell in range(8):
            for nx in range(shell + 1):
                self.orbitals.append(Orbital(len(self.orbitals), nx, shell - nx))
                self.Es.append(shell + 1)
        self.E_indices = lambda n: tuple(range(n * (n + 1) // 2, (n + 1) * (n + 2) // 2))
