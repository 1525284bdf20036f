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
    dim = 2
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


class Orbital3D:
    """phi_{nx,ny,nz}(r) = pi^-3/4 exp(-|r|^2/2) h_nx(x) h_ny(y) h_nz(z)  (HO3D: no upstream counterpart)."""

    __slots__ = ("k", "nx", "ny", "nz", "E")
    dim = 3

    def __init__(self, k, nx, ny, nz):
        self.k, self.nx, self.ny, self.nz, self.E = k, nx, ny, nz, nx + ny + nz + 1.5

    def __call__(self, x):
        return math.pi ** -0.75 * torch.exp(-0.5 * (x ** 2).sum(dim=-1)) \
            * Orbital._h(self.nx, x[..., 0]) * Orbital._h(self.ny, x[..., 1]) * Orbital._h(self.nz, x[..., 2])

    def __repr__(self):
        return f"Orbital3D(k={self.k}, n=({self.nx}, {self.ny}, {self.nz}))"


def orbital_dim(orbitals):
    """2 for HO2D orbitals, 3 for HO3D orbitals (a mixed tuple is an error)."""
    dims = {getattr(o, "dim", 2) for o in orbitals}
    if len(dims) > 1:
        raise TypeError("orbitals of different dimensions in one determinant")
    return dims.pop() if dims else 2


def orbital_indices(orbitals):
    """tuple of Orbital objects -> list of integer indices for the kernels."""
    out = []
    for o in orbitals:
        if not isinstance(o, (Orbital, Orbital3D)):
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
        """Low-lying Slater-determinant states (src/orbitals.py:33-54) as ((orbitals_up, orbitals_down), ...) and their
        energies, enumerated natively (ff_fermion_states, host C++).  ndown = 0 gives the reference's states in the
        reference's order; ndown != 0 -- which the reference rejects (src/orbitals.py:47-49) -- gives all pairs of
        subsets within deltaE of the ground state, by total energy, ties in (up, down)-lexicographic order."""
        import ctypes as C
        import numpy as np
        from . import _lib as L
        if nup < 0 or ndown < 0 or nup + ndown == 0:
            raise ValueError("fermion_states: need nup, ndown >= 0 and at least one particle")
        E = np.ascontiguousarray(self.Es, dtype=np.float64)
        args = (len(E), E.ctypes.data_as(C.c_void_p), int(nup), int(ndown), C.c_double(float(deltaE)))
        ns = L.lib().ff_fermion_states(*args, C.c_int64(0), None, None, None)
        if ns < 0:
            raise ValueError("fermion_states: " + L.lib().ff_last_error().decode())
        up = np.empty((ns, nup), dtype=np.int32); dn = np.empty((ns, ndown), dtype=np.int32); Es = np.empty(ns)
        L.lib().ff_fermion_states(*args, C.c_int64(ns), up.ctypes.data_as(C.c_void_p), dn.ctypes.data_as(C.c_void_p),
                                  Es.ctypes.data_as(C.c_void_p))
        states = tuple((tuple(self.orbitals[i] for i in u), tuple(self.orbitals[i] for i in d)) for u, d in zip(up.tolist(), dn.tolist()))
        ints = all(float(e).is_integer() for e in self.Es)
        return states, tuple(int(e) if ints else float(e) for e in Es)


class HO2D(Orbitals):
    """Orbitals of h = -1/2 laplacian + 1/2 r^2 in 2-D; 36 of them (shells 0..7), E = shell + 1."""

    def __init__(self):
        self.orbitals, self.Es = [], []
        for shell in range(8):
            for nx in range(shell + 1):
                self.orbitals.append(Orbital(len(self.orbitals), nx, shell - nx))
                self.Es.append(shell + 1)
        self.E_indices = lambda n: tuple(range(n * (n + 1) // 2, (n + 1) * (n + 2) // 2))


class HO3D(Orbitals):
    """Orbitals of h = -1/2 laplacian + 1/2 r^2 in 3-D: shells 0..7 (120 orbitals), E = shell + 3/2; list order
    "for shell: for nx in 0..shell: for ny in 0..shell-nx: (nx, ny, shell-nx-ny)".  The reference stops at two dimensions
    (src/orbitals.py:56); this is the set BASELINE.json configs[4] needs (nup = ndown = 10 fills shells 0..2)."""

    def __init__(self):
        self.orbitals, self.Es = [], []
        for shell in range(8):
            for nx in range(shell + 1):
                for ny in range(shell + 1 - nx):
                    self.orbitals.append(Orbital3D(len(self.orbitals), nx, ny, shell - nx - ny))
                    self.Es.append(shell + 1.5)
