"""y_grad_laplacian (src/utils.py:40-65) for the functions the hot path applies it to.

The reference obtains grad and Laplacian of log p by 1 + n*d extra autograd passes through nested adjoint
ODE solves.  Here they come out of one native pass; `f` must therefore be one of the callables that know
how to do that (GSVMC.logp / BetaVMC.logp bound methods, or a FreeFermion.log_prob closure made by
`freefermion_logp`)."""
import torch

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


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam for fp64 parameters on the GPU with the whole update in ONE launch (ff_adam_step: the single-tensor formula of
    torch/optim/adam.py operation for operation).  Same hyper-parameters, same per-parameter state -- step (a CPU scalar tensor),
    exp_avg, exp_avg_sq -- so state_dicts move between the two in both directions.  No amsgrad, no maximize, no closure-less tricks:
    anything this class does not do raises, it never falls back to another implementation silently."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("FusedAdam: lr >= 0, eps >= 0, weight_decay >= 0, betas in [0, 1)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @staticmethod
    def _refuse_unsupported(groups, states=()):
        for group in groups:
            for key in ("amsgrad", "maximize", "capturable", "differentiable"):
                if group.get(key):
                    raise RuntimeError(f"FusedAdam does not implement {key}=True (a torch.optim.Adam state with it cannot be continued here)")
        for st in states:
            if "max_exp_avg_sq" in st:
                raise RuntimeError("FusedAdam does not implement amsgrad (the loaded state carries max_exp_avg_sq)")

    def load_state_dict(self, state_dict):
        self._refuse_unsupported(state_dict.get("param_groups", ()), state_dict.get("state", {}).values())
        super().load_state_dict(state_dict)
        for st in self.state.values():      # (torch's fused Adam keeps `step` on the device: reading it there would wait for the GPU every update)
            if torch.is_tensor(st.get("step")) and st["step"].is_cuda:
                st["step"] = st["step"].detach().to("cpu", torch.float32)

    @torch.no_grad()
    def step(self, closure=None):
        from . import native
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._refuse_unsupported(self.param_groups)
        for group in self.param_groups:      # every parameter is validated before any step count moves (an error leaves the state as it was)
            for p in group["params"]:
                g = p.grad
                if g is not None and not (p.is_cuda and p.dtype == torch.float64 and p.is_contiguous() and g.dtype == torch.float64 and g.is_cuda
                                          and not g.is_sparse):
                    raise RuntimeError("FusedAdam serves contiguous fp64 parameters on the GPU with dense fp64 gradients (make_adam picks "
                                       "torch.optim.Adam for everything else)")
        for group in self.param_groups:
            ps, gs, ms, vs, steps = [], [], [], [], set()
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                steps.add(int(st["step"].item()))
                ps.append(p); gs.append(g if g.is_contiguous() else g.contiguous()); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"])
            if not ps:
                continue
            b1, b2 = group["betas"]
            for t in sorted(steps):      # (parameters that joined later have their own count: one launch per distinct count)
                sel = [k for k, p in enumerate(ps) if int(self.state[p]["step"].item()) == t]
                pick = lambda xs: [xs[k] for k in sel]
                native.adam_step(pick(ps), pick(gs), pick(ms), pick(vs), float(group["lr"]), b1, b2, group["eps"], group["weight_decay"], t)
        return loss


def make_adam(params, lr):
    """torch.optim.Adam as the reference's drivers build it (src/FermionHO2D.py:61, lr = 1e-2) -- as FusedAdam (one launch of the
    library's own kernel; PyTorch's fused implementation: two multi-tensor launches of 24 us each for 300 numbers, the default one seven)
    when every parameter is a contiguous fp64 tensor on the GPU, torch.optim.Adam otherwise."""
    params = list(params)
    if len(params) > 0 and all(p.is_cuda and p.dtype == torch.float64 and p.is_contiguous() for p in params):
        return FusedAdam(params, lr=lr)
    return torch.optim.Adam(params, lr=lr)
