"""Slater log-determinant primitives with the reference's signatures (src/slater.py).

forward / backward run the HIP kernels ff_slater_logabsdet_fwd / _bwd.  The reference's backward is built
with create_graph=True so that autograd can differentiate it again for the Laplacian; here second
derivatives are supplied natively (native.logprob(..., derivs=True), native.eloc) and double-backward
through this Function is not available.
"""
import torch

from . import native
from .orbitals import orbital_indices


def _flatten(x):
    *batch, n, dim = x.shape
    if dim != 2:
        raise ValueError("HO2D orbitals need dim = 2")
    return x.reshape(-1, n, 2), batch


class LogAbsSlaterDet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, orbitals, x):
        xf, batch = _flatten(x)
        table = native.orbital_table(orbital_indices(orbitals), x.device)
        ctx.save_for_backward(x)
        ctx.table = table
        return native.slater_fwd(table, xf).reshape(batch)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_logabsdet):
        x, = ctx.saved_tensors
        xf, _ = _flatten(x)
        gx = native.slater_bwd(ctx.table, xf, grad_logabsdet.reshape(-1).contiguous())
        return None, gx.reshape(x.shape)


def logabsslaterdet(orbitals, x):
    return LogAbsSlaterDet.apply(orbitals, x)


def _walker_states(state_indices_collection, batch, device):
    idx = []
    for k, times in state_indices_collection.items():
        idx.extend([k] * times)
    if len(idx) != batch:
        raise ValueError("state_indices_collection does not add up to the batch size")
    return torch.tensor(idx, dtype=torch.int32, device=device)


class LogAbsSlaterDetMultStates(torch.autograd.Function):
    @staticmethod
    def forward(ctx, states, state_indices_collection, x):
        if x.dim() != 3:
            raise ValueError("LogAbsSlaterDetMultStates: x is required to have only one batch dimension.")
        table = native.orbital_table([orbital_indices(s) for s in states], x.device)
        ws = _walker_states(state_indices_collection, x.shape[0], x.device)
        ctx.save_for_backward(x)
        ctx.table, ctx.ws = table, ws
        return native.slater_fwd(table, x.contiguous(), ws)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_logabsdet):
        x, = ctx.saved_tensors
        gx = native.slater_bwd(ctx.table, x.contiguous(), grad_logabsdet.contiguous(), ctx.ws)
        return None, None, gx


def logabsslaterdetmultstates(states, state_indices_collection, x):
    return LogAbsSlaterDetMultStates.apply(states, state_indices_collection, x)
