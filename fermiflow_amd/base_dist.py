"""Base distribution |psi_0|^2 of free fermions and its Metropolis sampler (src/base_dist.py)."""
import torch

from . import native
from .orbitals import orbital_indices, orbital_dim
from .slater import LogAbsSlaterDet, LogAbsSlaterDetMultStates, _walker_states


class BaseDist(object):
    def log_prob(self, x):
        pass

    def sample(self, sample_shape):
        pass


def _draw_seed():
    """64-bit Philox key drawn from torch's CPU generator, so torch.manual_seed() makes runs repeatable."""
    hi, lo = torch.randint(0, 2 ** 31 - 1, (2,), dtype=torch.int64).tolist()
    return (hi << 31) | lo


class FreeFermion(BaseDist):
    def __init__(self, device=torch.device("cpu")):
        super(FreeFermion, self).__init__()
        self.device = torch.device(device)
        self.walker_offset = 0   # set by the data-parallel driver so that shards draw disjoint Philox streams

    def log_prob(self, orbitals_up, orbitals_down, x):
        nup, ndown = len(orbitals_up), len(orbitals_down)
        if orbital_dim(tuple(orbitals_up) + tuple(orbitals_down)) == 3:      # HO3D: value only (ff_logprob3d)
            tu, td = self._tables(orbitals_up, orbitals_down)
            *batch, n, dim = x.shape
            return native.logprob3d(tu, td, nup, ndown, x.detach().reshape(-1, n, 3).contiguous()).reshape(batch)
        logabspsi = (LogAbsSlaterDet.apply(orbitals_up, x[..., :nup, :]) if nup != 0 else 0) \
            + (LogAbsSlaterDet.apply(orbitals_down, x[..., nup:, :]) if ndown != 0 else 0)
        return 2 * logabspsi

    def _tables(self, orbitals_up, orbitals_down):
        tu = native.orbital_table(orbital_indices(orbitals_up), self.device) if len(orbitals_up) else None
        td = native.orbital_table(orbital_indices(orbitals_down), self.device) if len(orbitals_down) else None
        return tu, td

    def sample(self, orbitals_up, orbitals_down, sample_shape, equilibrim_steps=100, tau=0.1, x_init=None, seed=None):
        """100-step Metropolis chain from N(0,1) walkers, fused in one kernel (src/base_dist.py:58-71).
        x_init (not in the reference): continue the chain of these walkers instead of starting from N(0,1).
        seed (not in the reference): the Philox key; default: drawn from torch's CPU generator."""
        _seed = (lambda: seed) if seed is not None else _draw_seed
        nup, ndown = len(orbitals_up), len(orbitals_down)
        B = 1
        for s in sample_shape:
            B *= int(s)
        tu, td = self._tables(orbitals_up, orbitals_down)
        if orbital_dim(tuple(orbitals_up) + tuple(orbitals_down)) == 3:      # HO3D walkers (B, n, 3)
            if x_init is not None:
                raise NotImplementedError("persistent walkers in three dimensions")
            x, _, _ = native.mcmc_sample3d(tu, td, nup, ndown, B, equilibrim_steps, tau, _seed(), self.device,
                                           walker_offset=self.walker_offset)
            return x.reshape(*sample_shape, nup + ndown, 3)
        if x_init is not None:
            x, _, _ = native.mcmc_continue(tu, td, nup, ndown, x_init.reshape(B, nup + ndown, 2), equilibrim_steps, tau,
                                           _seed(), walker_offset=self.walker_offset)
        else:
            x, _, _ = native.mcmc_sample(tu, td, nup, ndown, B, equilibrim_steps, tau, _seed(), self.device,
                                         walker_offset=self.walker_offset)
        return x.reshape(*sample_shape, nup + ndown, 2)

    def sample_with_noise(self, orbitals_up, orbitals_down, g0, g, u, tau=0.1):
        """Parity mode: consume explicit noise in the reference's draw order (randn(B,n,2); per step
        randn_like(x) then rand_like(p)).  Returns (x, logp, accept[S,B] uint8)."""
        tu, td = self._tables(orbitals_up, orbitals_down)
        if orbital_dim(tuple(orbitals_up) + tuple(orbitals_down)) == 3:
            return native.mcmc_sample_noise3d(tu, td, len(orbitals_up), len(orbitals_down), g0, g, u, tau)
        return native.mcmc_sample_noise(tu, td, len(orbitals_up), len(orbitals_down), g0, g, u, tau)

    # ---- finite temperature: one orbital set per walker ------------------------------------------
    def log_prob_multstates(self, states, state_indices_collection, x, method=2):
        if len(x.shape[:-2]) != 1:
            raise ValueError("FreeFermion.log_prob_multstates: x is required to have "
                             "only one batch dimension.")
        states_up, states_down = tuple(zip(*states))
        nup, ndown = len(states_up[0]), len(states_down[0])
        logabspsi = (LogAbsSlaterDetMultStates.apply(states_up, state_indices_collection, x[..., :nup, :])
                     if nup != 0 else 0) \
            + (LogAbsSlaterDetMultStates.apply(states_down, state_indices_collection, x[..., nup:, :])
               if ndown != 0 else 0)
        return 2 * logabspsi

    def sample_multstates(self, states, state_indices_collection, sample_shape,
                          equilibrim_steps=100, tau=0.1, cpu=False, method=2):
        if len(sample_shape) != 1:
            raise ValueError("FreeFermion.sample_multstates: sample_shape is "
                             "required to have only one batch dimension.")
        states_up, states_down = tuple(zip(*states))
        nup, ndown = len(states_up[0]), len(states_down[0])
        B = int(sample_shape[0])
        tu = native.orbital_table([orbital_indices(s) for s in states_up], self.device) if nup else None
        td = native.orbital_table([orbital_indices(s) for s in states_down], self.device) if ndown else None
        ws = _walker_states(state_indices_collection, B, self.device)
        x, _, _ = native.mcmc_sample(tu, td, nup, ndown, B, equilibrim_steps, tau, _draw_seed(), self.device,
                                     walker_offset=self.walker_offset, walker_state=ws)
        return x
