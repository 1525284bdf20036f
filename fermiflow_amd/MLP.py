"""Scalar MLP eta(r), mu(r): Linear(1,H) -> sigmoid -> Linear(H,1,no bias)  (src/MLP.py).

Parameter names/shapes (fc1.weight, fc1.bias, fc2.weight) are the reference's, so state_dicts and optimisers
are interchangeable.  Inside the CNF the MLP is evaluated by the fused HIP kernels straight from these
parameters; forward()/grad() serve direct calls through the ff_mlp_eval kernel.
"""
import torch

from . import native


class MLP(torch.nn.Module):
    def __init__(self, D_in, D_hidden):
        super(MLP, self).__init__()
        if D_in != 1:
            raise ValueError("the backflow potentials are univariate: D_in must be 1")
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
        # always the HIP kernel (detached result): derivatives wrt r come from .grad(); derivatives wrt the
        # parameters are produced by the fused adjoint kernel (flow.CNF), never by autograd through here.
        return native.mlp_eval(self.fc1.weight.detach(), self.fc1.bias.detach(), self.fc2.weight.detach(),
                               x.detach().reshape(-1).contiguous(), need_grad=need_grad)

    def forward(self, x):
        val, _ = self._eval(x, False)
        return val.reshape(x.shape)

    def d_sigmoid(self, output):
        return output * (1. - output)

    def grad(self, x):
        _, dval = self._eval(x, True)
        return dval.reshape(x.shape)
