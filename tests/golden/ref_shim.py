"""Load the *reference* FermiFlow (read-only at /root/reference) inside this container.

Used ONLY by tests/golden/make_golden.py to produce the committed golden vectors.
The reference never travels to the GPU box; nothing in tests/, bench.py or the package
imports this module at run time.

The reference cannot be imported unmodified here:
  1. `torchdiffeq` is not installed -> a stub module is put in sys.modules and the
     reference's own second ODE backend (`implementation="scipy"`,
     src/NeuralODE/nnModule.py:49-61, scipy RK45 = Dormand-Prince 5(4)) is bound instead.
  2. `Backflow._e_e_divergence` (src/equivariant_funs.py:46-47) takes `.norm()` of the full
     n x n difference tensor including the zero diagonal, whose 2nd derivative is NaN on
     torch 2.10.  Selecting the i<j pairs *before* `.norm()` is bit-identical in forward value
     and finite in all derivatives.
  3. the reference prints on every ODE solve; silenced.
"""
import sys, types, functools, io, contextlib

REF = "/root/reference"


def load(rtol=1e-6, atol=1e-8):
    import torch
    torch.set_default_dtype(torch.float64)
    if "torchdiffeq" not in sys.modules:
        stub = types.ModuleType("torchdiffeq")
        def _no(*a, **k):
            raise RuntimeError("torchdiffeq is not available in this container")
        stub.odeint = _no
        sys.modules["torchdiffeq"] = stub
    for p in (REF + "/src", REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    import NeuralODE.nnModule as nnModule
    import flow, equivariant_funs, utils as ref_utils, VMC  # noqa: F401

    # (1) scipy backend, chosen tolerances; propagates into every nested backward solve through
    #     ctx.implementation / ctx.rtol / ctx.atol (nnModule.py:72,93-94)
    orig = nnModule.solve_ivp_nnmodule.__wrapped__ if hasattr(nnModule.solve_ivp_nnmodule, "__wrapped__") \
        else nnModule.solve_ivp_nnmodule
    if not getattr(nnModule, "_ff_orig", None):
        nnModule._ff_orig = orig
    base = nnModule._ff_orig

    def quiet(*a, **k):
        k.setdefault("implementation", "scipy")
        k.setdefault("rtol", rtol)
        k.setdefault("atol", atol)
        with contextlib.redirect_stdout(io.StringIO()):
            return base(*a, **k)
    flow.solve_ivp_nnmodule = quiet
    nnModule.solve_ivp_nnmodule = quiet  # SolveIVP.backward looks the name up in module globals

    # (2) pairs-before-norm divergence
    def _e_e_divergence(self, x):
        _, n, dim = x.shape
        row, col = torch.triu_indices(n, n, offset=1)
        rij = (x[:, :, None] - x[:, None])[:, row, col, :]
        dij = rij.norm(dim=-1, keepdim=True)
        eta, d_eta = self.eta(dij), self.eta.grad(dij)
        return 2 * (d_eta * dij + dim * eta).sum(dim=(-2, -1))
    equivariant_funs.Backflow._e_e_divergence = _e_e_divergence

    # (3) silence y_grad_laplacian prints
    orig_ygl = getattr(ref_utils, "_ff_orig_ygl", None) or ref_utils.y_grad_laplacian
    ref_utils._ff_orig_ygl = orig_ygl

    def ygl(f, x):
        with contextlib.redirect_stdout(io.StringIO()):
            return orig_ygl(f, x)
    ref_utils.y_grad_laplacian = ygl

    import orbitals, slater, base_dist, MLP, potentials
    return types.SimpleNamespace(
        torch=torch, nnModule=nnModule, flow=flow, equivariant_funs=equivariant_funs,
        utils=ref_utils, VMC=VMC, orbitals=orbitals, slater=slater, base_dist=base_dist,
        MLP=MLP, potentials=potentials)
