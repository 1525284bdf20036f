"""Continuous normalizing flow wrapper (src/flow.py:6-55)."""
import torch

from .NeuralODE.nnModule import solve_ivp_nnmodule


class CNF(torch.nn.Module):
    def __init__(self, v, t_span):
        """v: the equivariant velocity field (a Backflow module); t_span: (T0, T)."""
        super(CNF, self).__init__()

        class V_wrapper(torch.nn.Module):
            def __init__(self, v):
                super(V_wrapper, self).__init__()
                self.v = v

            def forward(self, t, x):
                return self.v(x)
        self.v_wrapper = V_wrapper(v)

        class F(torch.nn.Module):
            def __init__(self, v):
                super(F, self).__init__()
                self.v = v

            def forward(self, t, x_and_logp):
                x, _ = x_and_logp
                return self.v(x), -self.v.divergence(x)
        self.f = F(v)

        self.t_span = t_span
        self.t_span_reverse = t_span[1], t_span[0]
        self.rtol, self.atol = 1e-6, 1e-8    # solve_ivp_nnmodule defaults, src/NeuralODE/nnModule.py:162

    def generate(self, z, nframes=None):
        if nframes is not None:
            raise NotImplementedError("generate(nframes=...) (animation frames) is outside the VMC hot path")
        return solve_ivp_nnmodule(self.v_wrapper, self.t_span, z, params_require_grad=False,
                                  rtol=self.rtol, atol=self.atol)

    def delta_logp(self, x, params_require_grad=False):
        batch = x.shape[0]
        z, delta_logp = solve_ivp_nnmodule(self.f, self.t_span_reverse,
                                           (x, torch.zeros(batch, device=x.device, dtype=x.dtype)),
                                           params_require_grad=params_require_grad, rtol=self.rtol, atol=self.atol)
        return z, delta_logp

    def backflow_potential(self):
        return self.v_wrapper.v.eta, self.v_wrapper.v.mu
