"""Scalar MLP eta(r), mu(r): Linear(1,H) -> sigmoid -> Linear(H,1,no bias)  (src/MLP.py).

Parameter names/shapes (fc1.weight, fc1.bias, fc2.weight) are the reference's, so state_dicts and optimisers
are interchangeable.  Inside the CNF the MLP is evaluated by the fused HIP kernels straight from these
parameters; forward()/grad() serve direct calls through the ff_mlp_eval kernel.
"""
import torch

from . import native


class _MLPValue(torch.autograd.Function):
    """MLP.forward, differentiable with respect to its input (first order): backward = grad_out * MLP.grad(x), both from the
    native kernels.  The parameters are not autograd inputs (their gradient comes from the fused adjoint, flow.CNF)."""

    @staticmethod
    def forward(ctx, x, module):
        ctx.module = module
        ctx.save_for_backward(x.detach())
        return module._eval(x, False)[0].reshape(*x.shape[:-1], 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return g * ctx.module._eval(x, True)[1].reshape(x.shape), None


class MLP(torch.nn.Module):
    def __init__(self, D_in, D_hidden):
        super(MLP, self).__init__()
        self.D_in = int(D_in)      # any input dimension up to 64 (src/MLP.py:9-16; the backflow potentials use 1)
        self.fc1 = torch.nn.Linear(D_in, D_hidden)
        self.fc2 = torch.nn.Linear(D_hidden, 1, bias=False)
        self.activation = torch.nn.Sigmoid()

    def init_zeros(self):
        torch.nn.init.zeros_(self.fc1.weight)
        torch.nn.init.zeros_(self.fc1.bias)
        torch.nn.init.zeros_(self.fc2.weight)

    def init_gaussian(self, seed):
        torch.manual_seed(seed)
        std = 1e-3
        torch.nn.init.normal_(self.fc1.weight, std=std)
        torch.nn.init.normal_(self.fc1.bias, std=std)
        torch.nn.init.normal_(self.fc2.weight, std=std)

    def _eval(self, x, need_grad):
        # always the HIP kernels (detached results): derivatives wrt the parameters are produced by the fused adjoint kernel
        # (flow.CNF), never by autograd through here.
        if x.shape[-1] != self.D_in:
            raise RuntimeError(f"MLP: expected inputs of shape (..., {self.D_in}), got {tuple(x.shape)}")
        w1, b1, w2 = self.fc1.weight.detach(), self.fc1.bias.detach(), self.fc2.weight.detach()
        if self.D_in == 1:
            return native.mlp_eval(w1, b1, w2, x.detach().reshape(-1).contiguous(), need_grad=need_grad)
        return native.mlp_eval_nd(w1, b1, w2, x.detach().reshape(-1, self.D_in).contiguous(), need_val=True, need_grad=need_grad)

    def forward(self, x):
        return _MLPValue.apply(x, self)

    def d_sigmoid(self, output):
        return output * (1. - output)

    def grad(self, x):
        _, dval = self._eval(x, True)
        return dval.reshape(x.shape)
