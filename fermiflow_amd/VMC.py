"""Variational Monte Carlo estimators with the reference's interface (src/VMC.py).

GSVMC.forward(batch) / BetaVMC.forward(batch) run the whole sweep natively:
    MCMC (ff_mcmc_sample) -> CNF.generate (ff_cnf_generate) -> local energy (ff_eloc)
    -> E, E_std -> parameter gradient by the fused adjoint (ff_cnf_adjoint)
and return a scalar whose .backward() deposits that gradient in the parameters' .grad -- so the
reference training loop (`gradE = model(batch); gradE.backward(); optimizer.step()`,
src/FermionHO2D.py:66-72) runs unchanged.  With torch.distributed initialised, `batch` is the GLOBAL
number of walkers; each rank handles its contiguous shard and the estimator sums are all-reduced (dist.py).
"""
import time
from collections import Counter

import torch

from . import dist as D
from . import native
from .orbitals import orbital_indices


class _ScalarWithParamGrads(torch.autograd.Function):
    """value (0-dim) that back-propagates pre-computed gradients into the given parameters."""

    @staticmethod
    def forward(ctx, value, grads, *params):
        ctx.grads = grads
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        return (None, None) + tuple(g * gk for gk in ctx.grads)


def _flow_params(cnf):
    v = cnf.v_wrapper.v
    return v, list(v.parameters())


def _split_like(flat, params):
    out, off = [], 0
    for p in params:
        out.append(flat[off:off + p.numel()].reshape(p.shape))
        off += p.numel()
    return out


class GSVMC(torch.nn.Module):
    def __init__(self, nup, ndown, orbitals, basedist, cnf, pair_potential, sp_potential=None):
        super(GSVMC, self).__init__()
        self.orbitals_up, self.orbitals_down = orbitals.orbitals[:nup], orbitals.orbitals[:ndown]
        self.nup, self.ndown = nup, ndown
        self.basedist = basedist
        self.cnf = cnf
        self.pair_potential = pair_potential
        self.sp_potential = sp_potential
        self.profile = None

    # -- pieces with the reference's names ---------------------------------------------------------
    def sample(self, sample_shape):
        z = self.basedist.sample(self.orbitals_up, self.orbitals_down, sample_shape)
        x = self.cnf.generate(z)
        return z, x

    def logp(self, x, params_require_grad=False):
        z, delta_logp = self.cnf.delta_logp(x, params_require_grad=params_require_grad)
        return self.basedist.log_prob(self.orbitals_up, self.orbitals_down, z) - delta_logp

    def _tables(self, device):
        tu = native.orbital_table(orbital_indices(self.orbitals_up), device) if self.nup else None
        td = native.orbital_table(orbital_indices(self.orbitals_down), device) if self.ndown else None
        return tu, td

    def local_energy(self, x, walker_state=None, want_stats=False):
        """logp, grad logp, laplacian logp, V and E_loc of every walker in one native pass (src/VMC.py:46-55)."""
        tu, td = self._tables(x.device)
        t0, t1 = self.cnf.t_span
        Z = getattr(self.pair_potential, "Z", 0.0)
        return native.eloc(tu, td, self.nup, self.ndown, self.cnf.v_wrapper.v.net(), x.detach(), t0, t1,
                           self.cnf.rtol, self.cnf.atol, Z, self.sp_potential is not None,
                           walker_state=walker_state, want_stats=want_stats)

    def _native_grad_laplacian(self, x):
        r = self.local_energy(x)
        return r["logp"], r["grad"], r["lap"]

    # -- the sweep ------------------------------------------------------------------------------------
    def forward(self, batch):
        rank, ws = D.world()
        offset, nloc = D.shard(batch, rank, ws)
        self.basedist.walker_offset = offset
        prof = self.profile          # None, or a dict that receives per-stage torch.cuda.Event pairs + ODE stats
        ev = {}

        def mark(name):
            if prof is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev[name] = e
        with torch.no_grad():
            mark("t0")
            z = self.basedist.sample(self.orbitals_up, self.orbitals_down, (nloc,))
            mark("mcmc")
            x = self.cnf.generate(z)
            mark("generate")
            p1 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if prof is not None else None
            tu, td = self._tables(x.device)
            t0, t1 = self.cnf.t_span
            r = native.eloc(tu, td, self.nup, self.ndown, self.cnf.v_wrapper.v.net(), x, t0, t1, self.cnf.rtol,
                            self.cnf.atol, getattr(self.pair_potential, "Z", 0.0), self.sp_potential is not None,
                            want_stats=prof is not None, pass1_events=p1)
            mark("eloc")
            Eloc = r["eloc"]
            s0 = native.reduce_moments(Eloc, 0.0)
            self.E, self.E_std, nglob = D.global_mean_std(
                s0[0], nloc, lambda m: native.reduce_moments(Eloc, m)[1])
            w = (Eloc - self.E) / nglob
            v, params = _flow_params(self.cnf)
            mark("estimator")
            _, gp = native.cnf_adjoint(v.net(), r["z"], w[:, None, None] * r["glogp0"], -w, t0, t1,
                                       self.cnf.rtol, self.cnf.atol, need_gx=False)
            buf = torch.cat([(r["logp"] * w).sum().reshape(1), gp])
            D.all_reduce_sum_(buf)
            mark("adjoint")
        if prof is not None:
            prof.setdefault("events", []).append(ev)
            prof.setdefault("pass1", []).append(p1)
            prof.setdefault("eloc_stats", []).append(r["stats"])
        self.Eloc, self.x = Eloc, x
        grads = _split_like(buf[1:], params)
        return _ScalarWithParamGrads.apply(buf[0], grads, *params)


class BetaVMC(torch.nn.Module):
    def __init__(self, beta, nup, ndown, deltaE, boltzmann, orbitals, basedist, cnf, pair_potential, sp_potential=None):
        super(BetaVMC, self).__init__()
        self.beta = beta
        self.nup, self.ndown = nup, ndown
        self.states, self.Es_original = orbitals.fermion_states(nup, ndown, deltaE)
        self.Es_original = torch.tensor(self.Es_original, dtype=torch.float64)
        self.Nstates = len(self.states)
        self.log_state_weights = torch.nn.Parameter(
            -self.beta * (self.Es_original - self.Es_original[0]) if boltzmann else torch.randn(self.Nstates, dtype=torch.float64))
        self.basedist = basedist
        self.cnf = cnf
        self.pair_potential = pair_potential
        self.sp_potential = sp_potential

    def sample(self, sample_shape, nframes=None):
        from torch.distributions.categorical import Categorical
        self.state_dist = Categorical(logits=self.log_state_weights)
        state_indices = self.state_dist.sample(sample_shape)
        self.state_indices_collection = Counter(sorted(state_indices.tolist()))
        z = self.basedist.sample_multstates(self.states, self.state_indices_collection, sample_shape)
        x = self.cnf.generate(z, nframes=nframes)
        return z, x

    def logp(self, x, params_require_grad=False):
        z, delta_logp = self.cnf.delta_logp(x, params_require_grad=params_require_grad)
        return self.basedist.log_prob_multstates(self.states, self.state_indices_collection, z) - delta_logp

    def _state_tables(self, device):
        states_up, states_down = tuple(zip(*self.states))
        tu = native.orbital_table([orbital_indices(s) for s in states_up], device) if self.nup else None
        td = native.orbital_table([orbital_indices(s) for s in states_down], device) if self.ndown else None
        return tu, td

    def local_energy(self, x, walker_state):
        tu, td = self._state_tables(x.device)
        t0, t1 = self.cnf.t_span
        Z = getattr(self.pair_potential, "Z", 0.0)
        return native.eloc(tu, td, self.nup, self.ndown, self.cnf.v_wrapper.v.net(), x.detach(), t0, t1,
                           self.cnf.rtol, self.cnf.atol, Z, self.sp_potential is not None, walker_state=walker_state)

    def _native_grad_laplacian(self, x):
        r = self.local_energy(x, self._walker_state(x.device))
        return r["logp"], r["grad"], r["lap"]

    def _walker_state(self, device):
        return torch.tensor(list(self.state_indices_collection.elements()), dtype=torch.int32, device=device)

    def forward(self, batch):
        """Single-process finite-temperature sweep (src/VMC.py:114-171)."""
        with torch.no_grad():
            _, x = self.sample((batch,))
            device = x.device
            ws = self._walker_state(device)
            r = self.local_energy(x, ws)
            Eloc = r["eloc"]
            self.E, self.E_std = Eloc.mean().item(), Eloc.std().item()
        state_indices = ws.to(torch.int64)
        from torch.distributions.categorical import Categorical
        self.state_dist = Categorical(logits=self.log_state_weights)      # with autograd (sample() ran under no_grad)
        logp_states = self.state_dist.log_prob(state_indices.to(self.log_state_weights.device)).to(device)
        with torch.no_grad():
            Floc = Eloc + logp_states.detach() / self.beta
            self.F, self.F_std = Floc.mean().item(), Floc.std().item()
            self.S = -logp_states.detach().mean().item()
            self.logp_states_all = self.state_dist.log_prob(
                torch.arange(self.Nstates, device=self.log_state_weights.device)).detach()
            self.S_analytical = -(self.logp_states_all * self.logp_states_all.exp()).sum().item()
        gradF_phi = (logp_states * (Floc - self.F)).mean()
        with torch.no_grad():
            # per-state baseline (src/VMC.py:164-169): segmented mean over the state-sorted walkers
            sums = torch.zeros(self.Nstates, dtype=torch.float64, device=device).index_add_(0, state_indices, Eloc)
            cnts = torch.zeros(self.Nstates, dtype=torch.float64, device=device).index_add_(
                0, state_indices, torch.ones_like(Eloc))
            Eloc_x_mean = (sums / cnts.clamp(min=1.0))[state_indices]
            w = (Eloc - Eloc_x_mean) / batch
            v, params = _flow_params(self.cnf)
            t0, t1 = self.cnf.t_span
            _, gp = native.cnf_adjoint(v.net(), r["z"], w[:, None, None] * r["glogp0"], -w, t0, t1,
                                       self.cnf.rtol, self.cnf.atol, need_gx=False)
            val = (r["logp"] * w).sum()
        gradF_theta = _ScalarWithParamGrads.apply(val, _split_like(gp, params), *params)
        return gradF_phi, gradF_theta
