"""solve_ivp_nnmodule with the reference's signature (src/NeuralODE/nnModule.py:161-188).

The reference integrates an arbitrary nn.Module right-hand side with torchdiffeq/scipy and differentiates
through it with nested adjoint solves.  Here the only right-hand sides that exist natively are the two the VMC
hot path uses -- CNF's `V_wrapper` (x' = v(x)) and `F` ((x', logp') = (v, -div v)) over a Backflow field --
which are integrated by the fused Dormand-Prince HIP kernels with an adjoint HIP kernel as backward.
"""
import torch

from .. import _lib as L
from .. import native


class _Generate(torch.autograd.Function):
    """x(t_end) from x(t_start) under x' = v(x); backward = adjoint sweep back to t_start."""

    @staticmethod
    def forward(ctx, v, t_span, rtol, atol, z, *params):
        net = v.net()
        x = native.cnf_generate(net, z.contiguous(), t_span[0], t_span[1], rtol, atol)
        ctx.v, ctx.t_span, ctx.tol, ctx.nparams = v, t_span, (rtol, atol), len(params)
        ctx.save_for_backward(x)
        return x

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_x):
        x, = ctx.saved_tensors
        net = ctx.v.net()
        zeros = torch.zeros(x.shape[0], dtype=x.dtype, device=x.device)
        gz, gp = native.cnf_adjoint(net, x, grad_x.contiguous(), zeros, ctx.t_span[1], ctx.t_span[0], *ctx.tol)
        return (None, None, None, None, gz) + _split(gp, ctx.v, ctx.nparams)


class _DeltaLogp(torch.autograd.Function):
    """(z, delta) at t_end from (x, 0) at t_start under (v, -div v)."""

    @staticmethod
    def forward(ctx, v, t_span, rtol, atol, x, *params):
        net = v.net()
        # native kernel integrates from ode.t1 down to ode.t0
        z, dl = native.cnf_delta_logp(net, x.contiguous(), t_span[1], t_span[0], rtol, atol)
        ctx.v, ctx.t_span, ctx.tol, ctx.nparams = v, t_span, (rtol, atol), len(params)
        ctx.save_for_backward(z)
        return z, dl

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_z, grad_dl):
        z, = ctx.saved_tensors
        net = ctx.v.net()
        gx, gp = native.cnf_adjoint(net, z, grad_z.contiguous(), grad_dl.contiguous(), ctx.t_span[1], ctx.t_span[0],
                                    *ctx.tol)
        return (None, None, None, None, gx) + _split(gp, ctx.v, ctx.nparams)


def _split(gp, v, nparams):
    """flat [eta.w1, eta.b1, eta.w2, mu.w1, mu.b1, mu.w2] -> tuple shaped like Backflow.parameters()."""
    if nparams == 0:
        return ()
    out, off = [], 0
    for p in v.parameters():
        out.append(gp[off:off + p.numel()].reshape(p.shape))
        off += p.numel()
    return tuple(out)


def solve_ivp_nnmodule(f, t_span, x0s, params_require_grad=True,
                       implementation="hip", rtol=1e-6, atol=1e-8):
    if not isinstance(f, torch.nn.Module):
        raise ValueError("f is required to be an instance of torch.nn.Module.")
    from ..equivariant_funs import Backflow
    v = getattr(f, "v", None)
    if not isinstance(v, Backflow):
        raise NotImplementedError("solve_ivp_nnmodule: only the CNF right-hand sides over a Backflow field "
                                  "(flow.CNF.v_wrapper / flow.CNF.f) have native kernels")
    params = tuple(v.parameters()) if params_require_grad else ()
    if isinstance(x0s, torch.Tensor):
        return _Generate.apply(v, tuple(t_span), rtol, atol, x0s, *params)
    x, logp0 = x0s
    z, dl = _DeltaLogp.apply(v, tuple(t_span), rtol, atol, x, *params)
    return z, dl + logp0
