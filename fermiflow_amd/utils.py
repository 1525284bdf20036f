"""y_grad_laplacian (src/utils.py:40-65) for the functions the hot path applies it to.

The reference obtains grad and Laplacian of log p by 1 + n*d extra autograd passes through nested adjoint
ODE solves.  Here they come out of one native pass; `f` must therefore be one of the callables that know
how to do that (GSVMC.logp / BetaVMC.logp bound methods, or a FreeFermion.log_prob closure made by
`freefermion_logp`)."""
from . import native
from .orbitals import orbital_indices, orbital_dim


class freefermion_logp:
    """log p_0(x) of a FreeFermion state as a callable understood by y_grad_laplacian."""

    def __init__(self, basedist, orbitals_up, orbitals_down):
        self.nup, self.ndn = len(orbitals_up), len(orbitals_down)
        self.tu = native.orbital_table(orbital_indices(orbitals_up), basedist.device) if self.nup else None
        self.td = native.orbital_table(orbital_indices(orbitals_down), basedist.device) if self.ndn else None
        self._f = native.logprob3d if orbital_dim(tuple(orbitals_up) + tuple(orbitals_down)) == 3 else native.logprob

    def __call__(self, x):
        return self._f(self.tu, self.td, self.nup, self.ndn, x.detach().contiguous())

    def _native_grad_laplacian(self, x):
        return self._f(self.tu, self.td, self.nup, self.ndn, x.detach().contiguous(), derivs=True)


def y_grad_laplacian(f, x):
    target = getattr(f, "__self__", f)
    fn = getattr(target, "_native_grad_laplacian", None)
    if fn is None:
        raise NotImplementedError("y_grad_laplacian: no native gradient/Laplacian for this callable "
                                  "(supported: GSVMC.logp, BetaVMC.logp, utils.freefermion_logp)")
    return fn(x)


def make_adam(params, lr):
    """torch.optim.Adam as the reference's drivers build it (src/FermionHO2D.py:61, lr = 1e-2) -- with PyTorch's fused
    implementation when every parameter lives on the GPU: the same update in 1 launch instead of 7 (each tiny launch is
    ~5 us of a 2.5 ms iteration).  FERMIFLOW_FUSED_ADAM=0 keeps the default implementation."""
    import os
    import torch
    params = list(params)
    fused = os.environ.get("FERMIFLOW_FUSED_ADAM", "1") != "0" and len(params) > 0 and all(p.is_cuda for p in params)
    return torch.optim.Adam(params, lr=lr, fused=True) if fused else torch.optim.Adam(params, lr=lr)
