"""Permutation-equivariant backflow velocity field (src/equivariant_funs.py):
    v_i = sum_{j != i} eta(|r_i - r_j|) (r_i - r_j) + mu(|r_i|) r_i
and its hand-derived divergence.  Stand-alone calls run ff_backflow_v_div (any n <= 24, d <= 3, any H);
inside flow.CNF the field is fused into the ODE kernels.
"""
import torch

from . import _lib as L
from . import native


class _BackflowV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module):
        xd = x.detach().contiguous()
        ctx.module, ctx.shape = module, x.shape
        ctx.save_for_backward(xd)
        return native.backflow_v_div(module.net(), xd.reshape(-1, *x.shape[-2:]), need_v=True, need_div=False)[0].reshape(x.shape)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        xd, = ctx.saved_tensors
        Aw, _ = native.backflow_vjp(ctx.module.net(), xd.reshape(-1, *ctx.shape[-2:]), w=g.contiguous().reshape(-1, *ctx.shape[-2:]))
        return Aw.reshape(ctx.shape), None


class _BackflowDiv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module):
        xd = x.detach().contiguous()
        ctx.module, ctx.shape = module, x.shape
        ctx.save_for_backward(xd)
        return native.backflow_v_div(module.net(), xd.reshape(-1, *x.shape[-2:]), need_v=False, need_div=True)[1].reshape(x.shape[:-2])

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        xd, = ctx.saved_tensors
        _, gd = native.backflow_vjp(ctx.module.net(), xd.reshape(-1, *ctx.shape[-2:]), need_gdiv=True)
        return gd.reshape(ctx.shape) * g.reshape(*ctx.shape[:-2], 1, 1), None


class Backflow(torch.nn.Module):
    def __init__(self, eta, mu=None):
        """ The argument eta must be an instance of torch.nn.Module. """
        super(Backflow, self).__init__()
        self.eta = eta
        self.mu = mu

    def net(self, radial=None, refresh=False):
        """Device view of the weights (+ the radial table built for them, ~20 us).  The VMC sweeps call this with
        refresh=True once per sweep and hand the result to their three integrations.  Other callers get a cached view
        keyed on the parameters' (data_ptr, _version): in-place edits through `p.data` (EMA code, hand-written
        optimisers) do NOT bump _version -- call invalidate() after such an edit (or pass refresh=True)."""
        key = (radial or L.RADIAL_MODE,) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if refresh or getattr(self, "_net_key", None) != key:
            self._net_key, self._net = key, L.Net(self.eta, self.mu, radial=radial)
        return self._net

    def invalidate(self):
        """Forget the cached radial table (needed only after `p.data` edits, which autograd's version counter misses)."""
        self._net_key = None

    def forward(self, x):
        """v(x), differentiable with respect to x (first order: ff_backflow_vjp supplies (dv/dx)^T w); the parameters are not
        autograd inputs here -- their gradient is the fused adjoint's business (flow.CNF)."""
        return _BackflowV.apply(x, self)

    def divergence(self, x):
        """div v(x) by the hand-derived formula (src/equivariant_funs.py:93-102), differentiable with respect to x."""
        return _BackflowDiv.apply(x, self)

    def _e_e(self, x):
        return native.backflow_v_div(L.Net(self.eta, None, radial="exact"), x.detach().contiguous(), True, False)[0]

    def _e_e_divergence(self, x):
        return native.backflow_v_div(L.Net(self.eta, None, radial="exact"), x.detach().contiguous(), False, True)[1]
