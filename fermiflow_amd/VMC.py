"""Variational Monte Carlo estimators with the reference's interface (src/VMC.py).

GSVMC.forward(batch) / BetaVMC.forward(batch) run the whole sweep natively:
    MCMC (ff_mcmc_sample) -> CNF.generate (ff_cnf_generate) -> local energy (ff_eloc)
    -> E, E_std -> parameter gradient by the fused adjoint (ff_cnf_adjoint)
and return a scalar whose .backward() deposits that gradient in the parameters' .grad -- so the
reference training loop (`gradE = model(batch); gradE.backward(); optimizer.step()`,
src/FermionHO2D.py:66-72) runs unchanged.  With torch.distributed initialised, `batch` is the GLOBAL
number of walkers; each rank handles its contiguous shard and the estimator sums are all-reduced (dist.py).
"""
import os
import time
from collections import Counter

import torch

from . import dist as D
from . import native
from .orbitals import orbital_indices


class _ScalarWithParamGrads(torch.autograd.Function):
    """value (0-dim) that back-propagates pre-computed gradients into the given parameters."""

    @staticmethod
    def forward(ctx, value, flat_grads, *params):
        ctx.flat, ctx.shapes = flat_grads, [p.shape for p in params]
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        scaled = g * ctx.flat        # one launch for all parameters; the per-parameter gradients are views of it
        out, off = [], 0
        for shp in ctx.shapes:
            n = 1
            for d in shp:
                n *= d
            out.append(scaled[off:off + n].reshape(shp))
            off += n
        return (None, None) + tuple(out)


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
        # ODE step-size warm start inside forward(); FERMIFLOW_WARM_START=0 restores the cold start
        self.warm_start = os.environ.get("FERMIFLOW_WARM_START", "1") != "0"
        self._h_flow = None
        self._E_dev = None
        # Persistent walkers (off by default: the reference draws fresh N(0,1) walkers and runs 100 steps in every
        # iteration, src/base_dist.py:62-64): keep the chains and advance them `persistent_steps` steps per sweep.
        self.persistent_walkers = False
        self.persistent_steps = 10
        self._z_prev = None
        # sensitivities' first step / flow's largest step (measured, tools/probes/warm_start.py: the best factor is 0.6 up
        # to 8 particles, 0.4-0.5 at 10, 0.4 at 12; too large a factor costs a rejected step = 7 evaluations)
        n = nup + ndown
        self._h_scale_eloc = 0.6 if n <= 8 else (0.45 if n <= 10 else 0.4)

    # energy estimate of the last forward() (python floats as in the reference, src/VMC.py:57; read lazily from the device)
    @property
    def E(self):
        return self._E_dev.item()

    @property
    def E_std(self):
        n = self._E_n
        return (self._E_ss / (n - 1)).sqrt().item() if n > 1 else float("nan")

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
            if self.persistent_walkers and self._z_prev is not None and self._z_prev.shape[0] == nloc:
                # opt-in: continue the previous sweep's chains for a few steps instead of 100 steps from N(0,1)
                z = self.basedist.sample(self.orbitals_up, self.orbitals_down, (nloc,),
                                         equilibrim_steps=self.persistent_steps, x_init=self._z_prev)
            else:
                z = self.basedist.sample(self.orbitals_up, self.orbitals_down, (nloc,))
            self._z_prev = z if self.persistent_walkers else None
            mark("mcmc")
            # Walker schedule (include/fermiflow.h, ff_ode.walker_cost/_order): the flow pass reports a cost class per
            # walker (how close its trajectory comes to a vanishing radius); the local-energy pass, whose step count
            # depends on exactly that, takes the expensive walkers first.
            t0, t1 = self.cnf.t_span
            net = self.cnf.v_wrapper.v.net()
            steps = torch.empty(nloc, dtype=torch.int32, device=z.device)
            # Step-size warm start (ff_ode.walker_h_*): the three integrations of a sweep follow the same trajectories, so
            # each one opens with the step size the previous one settled on instead of the ~20x too small Hairer start
            # (scaled: the sensitivity system wants 0.4-0.6 of the flow's step depending on the particle number, the
            # adjoint ~1.25 of the sensitivities'); the flow pass
            # itself starts from the step sizes of the previous sweep.  Error control per step is unchanged.
            warm = self.warm_start
            hg = torch.empty(nloc, dtype=torch.float64, device=z.device) if warm else None
            hprev = self._h_flow if (warm and self._h_flow is not None and self._h_flow.shape[0] == nloc
                                     and self._h_flow.device == z.device) else None
            x = native.cnf_generate(net, z, t0, t1, self.cnf.rtol, self.cnf.atol, walker_cost=steps,
                                    walker_h_init=hprev, walker_h_scale=0.75, walker_h_out=hg)
            self._h_flow = hg
            he = torch.empty_like(hg) if warm else None
            order = native.walker_order(steps)
            mark("generate")
            p1 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if prof is not None else None
            tu, td = self._tables(x.device)
            r = native.eloc(tu, td, self.nup, self.ndown, net, x, t0, t1, self.cnf.rtol,
                            self.cnf.atol, getattr(self.pair_potential, "Z", 0.0), self.sp_potential is not None,
                            want_stats=prof is not None, pass1_events=p1, walker_order=order,
                            walker_h_init=hg, walker_h_scale=self._h_scale_eloc, walker_h_out=he)
            mark("eloc")
            Eloc = r["eloc"]
            # E and E_std stay on the device (model.E / model.E_std convert on access): no host round trip inside the
            # sweep, so kernel launches keep running ahead of the GPU
            # One pass, one all-reduce: first and second moment about a shift every rank knows -- the previous sweep's E
            # (0 on the first sweep) -- so the subtraction below cancels nothing that matters (|E - shift| << E_std).
            shift = self._E_dev.reshape(1) if self._E_dev is not None else torch.zeros(1, dtype=Eloc.dtype, device=Eloc.device)
            mom = native.reduce_moments(Eloc, shift_dev=shift)          # [sum(e - c), sum((e - c)^2)]
            D.all_reduce_sum_(mom)
            self._E_dev = shift[0] + mom[0] / batch
            self._E_ss = mom[1] - mom[0] * mom[0] / batch
            self._E_n = batch
            w = (Eloc - self._E_dev) / batch
            v, params = _flow_params(self.cnf)
            mark("estimator")
            _, gp = native.cnf_adjoint(v.net(), r["z"], w[:, None, None] * r["glogp0"], -w, t0, t1,
                                       self.cnf.rtol, self.cnf.atol, need_gx=False,   # (uniform cost: no schedule)
                                       walker_h_init=he, walker_h_scale=1.25)
            buf = torch.cat([(r["logp"] * w).sum().reshape(1), gp])
            D.all_reduce_sum_(buf)
            mark("adjoint")
        if prof is not None:
            prof.setdefault("events", []).append(ev)
            prof.setdefault("pass1", []).append(p1)
            prof.setdefault("eloc_stats", []).append(r["stats"])
        self.Eloc, self.x = Eloc, x
        return _ScalarWithParamGrads.apply(buf[0], buf[1:], *params)


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
        """Draw the many-body state of every walker of the GLOBAL batch (CPU generator, so every rank draws the same
        list), sort by state as the reference does (src/VMC.py:94-96) and keep this rank's contiguous shard."""
        from torch.distributions.categorical import Categorical
        self.state_dist = Categorical(logits=self.log_state_weights)
        cpu_dist = Categorical(logits=self.log_state_weights.detach().cpu())
        all_idx = sorted(cpu_dist.sample(sample_shape).tolist())
        rank, world = D.world()
        off, cnt = D.shard(len(all_idx), rank, world)
        self.state_indices_collection = Counter(all_idx[off:off + cnt])
        self.basedist.walker_offset = off
        self._nglobal = len(all_idx)
        z = self.basedist.sample_multstates(self.states, self.state_indices_collection, (cnt,))
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
        """Finite-temperature sweep (src/VMC.py:114-171).  Under torch.distributed `batch` is the global walker count;
        the per-state sums and both gradients are all-reduced (2*Nstates + Nstates + 3(He+Hm) doubles)."""
        with torch.no_grad():
            _, x = self.sample((batch,))
            device = x.device
            ws = self._walker_state(device)
            nloc, nglob = ws.numel(), self._nglobal
            r = self.local_energy(x, ws)
            Eloc = r["eloc"]
            self.E, self.E_std, _ = D.global_mean_std(Eloc.sum(), nloc, lambda m: ((Eloc - m) ** 2).sum())
            state_indices = ws.to(torch.int64)
            logits = self.log_state_weights.detach().to(device)
            logp_all = torch.log_softmax(logits, dim=0)
            logp_states = logp_all[state_indices]
            Floc = Eloc + logp_states / self.beta
            self.F, self.F_std, _ = D.global_mean_std(Floc.sum(), nloc, lambda m: ((Floc - m) ** 2).sum())
            # per-state sums: entropy estimate, baseline of the theta-gradient, gradient wrt the state logits
            cF = (Floc - self.F) / nglob
            stat = torch.zeros(4, self.Nstates, dtype=torch.float64, device=device)
            stat[0].index_add_(0, state_indices, Eloc)
            stat[1].index_add_(0, state_indices, torch.ones_like(Eloc))
            stat[2].index_add_(0, state_indices, cF)
            stat[3, 0] = (logp_states * cF).sum()           # value of gradF_phi (local part)
            D.all_reduce_sum_(stat)
            sums, cnts, cF_state = stat[0], stat[1], stat[2]
            self.S = -(cnts * logp_all).sum().item() / nglob
            self.logp_states_all = logp_all.to(self.log_state_weights.device)
            self.S_analytical = -(logp_all * logp_all.exp()).sum().item()
            # d/dlogits sum_b cF_b log_softmax(logits)[s_b] = cF_state - softmax * sum(cF_state)
            g_phi = (cF_state - logp_all.exp() * cF_state.sum()).to(self.log_state_weights.device)
            Eloc_x_mean = (sums / cnts.clamp(min=1.0))[state_indices]       # per-state baseline (src/VMC.py:164-169)
            w = (Eloc - Eloc_x_mean) / nglob
            v, params = _flow_params(self.cnf)
            t0, t1 = self.cnf.t_span
            _, gp = native.cnf_adjoint(v.net(), r["z"], w[:, None, None] * r["glogp0"], -w, t0, t1,
                                       self.cnf.rtol, self.cnf.atol, need_gx=False)
            buf = torch.cat([(r["logp"] * w).sum().reshape(1), gp])
            D.all_reduce_sum_(buf)
        self.Eloc, self.x = Eloc, x
        gradF_phi = _ScalarWithParamGrads.apply(stat[3, 0].to(self.log_state_weights.device), g_phi.reshape(-1), self.log_state_weights)
        gradF_theta = _ScalarWithParamGrads.apply(buf[0], buf[1:], *params)
        return gradF_phi, gradF_theta
