"""Variational Monte Carlo estimators with the reference's interface (src/VMC.py).

GSVMC.forward(batch) / BetaVMC.forward(batch) run the whole sweep natively:
    MCMC (ff_mcmc_sample) -> CNF.generate (ff_cnf_generate) -> local energy (ff_eloc)
    -> E, E_std -> parameter gradient by the fused adjoint (ff_cnf_adjoint)
and return a scalar whose .backward() deposits that gradient in the parameters' .grad -- so the
reference training loop (`gradE = model(batch); gradE.backward(); optimizer.step()`,
src/FermionHO2D.py:66-72) runs unchanged.  With torch.distributed initialised, `batch` is the GLOBAL
number of walkers; each rank handles its contiguous shard and the estimator sums are all-reduced (dist.py).

`forward_from(z, ...)` is the same sweep on GIVEN base-distribution walkers (every line after the sampling is
shared with forward()); the parity tests feed it the reference's walkers.

Nothing inside a sweep waits for the host: E, E_std (F, F_std, S ...) stay on the device and are converted when
the attributes are read.
"""
import os
from collections import Counter

import torch

from . import dist as D
from . import native
from .orbitals import orbital_indices, orbital_dim


class _ScalarWithParamGrads(torch.autograd.Function):
    """value (0-dim) that back-propagates pre-computed gradients into the given parameters."""

    @staticmethod
    def forward(ctx, value, flat_grads, *params):
        ctx.flat, ctx.shapes = flat_grads, [p.shape for p in params]
        return value.view_as(value)      # (an alias, no copy launch; nothing ever writes to it)

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


class _SweepScalar(torch.Tensor):
    """What a sweep returns: the 0-dim surrogate value, attached to the autograd graph (through _ScalarWithParamGrads) like any
    tensor -- and with a short cut for the one thing the reference's training loop does with it, `gradE.backward()`
    (src/FermionHO2D.py:71): called without arguments on the sweep's own result it hands the pre-computed gradient views to the
    parameters' .grad directly (accumulating if they are set), instead of running the graph -- ones_like, a multiplication by 1 and
    their launches, 15 us of a 1.6 ms iteration.  Anything else -- a gradient argument, inputs=, create_graph, or backward() on a tensor
    computed FROM this one -- takes the ordinary autograd path with the same result."""

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        fast = getattr(self, "_ff_fast", None)
        if fast is None or gradient is not None or inputs is not None or create_graph:
            return super().backward(gradient, retain_graph, create_graph, inputs=inputs)
        params, flat = fast
        # The short cut never runs the autograd graph, so tensor hooks on the parameters (register_hook,
        # register_post_accumulate_grad_hook: FSDP reducers, hook-based clipping) would not fire: with any such hook present take the
        # ordinary path.  torch's DistributedDataParallel is NOT detectable this way -- its reducer hangs on the C++ AccumulateGrad node,
        # which neither attribute shows -- and it is not the supported data-parallel mode: the sweeps all-reduce the estimator
        # and the parameter gradient themselves (fermiflow_amd/dist.py), so every rank's .grad is already the global one.  A model that is
        # wrapped in DDP anyway sets `model.fast_backward = False` (the sweep then returns a scalar without the short cut).
        # Note that on the short cut the parameters' .grad are VIEWS of one flat buffer.
        if any(getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None) for p in params):
            return super().backward(gradient, retain_graph, create_graph, inputs=inputs)
        off = 0
        for p in params:
            n = p.numel()
            view = flat[off:off + n].view(p.shape)
            off += n
            if p.requires_grad:
                if p.grad is None:
                    p.grad = view
                else:
                    p.grad.add_(view)


def _sweep_scalar(value, flat, params, fast=True):
    out = _ScalarWithParamGrads.apply(value, flat, *params).as_subclass(_SweepScalar)
    if fast:
        out._ff_fast = (list(params), flat)
    return out


def _flow_params(cnf):
    v = cnf.v_wrapper.v
    return v, list(v.parameters())


def potential_plan(pair_potential, sp_potential):
    """How `pair_potential.V(x) + sp_potential.V(x)` (src/VMC.py:51-53) is evaluated: (Z, use_ho, extra).

    The stock potentials -- CoulombPairPotential(Z) and HO(), exact types -- are fused into the local-energy finish
    (ff_eloc: Z, use_ho).  Anything else (a PairPotential subclass with its own v(), another trap) keeps the reference's
    generic semantics: the native pass is given Z = 0 / no trap for that part and `extra` lists the objects whose V(x) is
    called on the device walkers and added to E_loc.  An object without a callable V raises TypeError -- nothing is ever
    silently treated as Z = 0 or as a harmonic trap."""
    from .potentials import CoulombPairPotential, HO
    Z, use_ho, extra = 0.0, False, []
    if type(pair_potential) is CoulombPairPotential:
        Z = float(pair_potential.Z)
    elif pair_potential is not None:
        if not callable(getattr(pair_potential, "V", None)):
            raise TypeError(f"pair_potential {pair_potential!r} has no V(x) (src/potentials.py:23-47)")
        extra.append(pair_potential)
    else:
        raise TypeError("pair_potential is required (the reference calls pair_potential.V(x) unconditionally, src/VMC.py:51)")
    if type(sp_potential) is HO:
        use_ho = True
    elif sp_potential:                       # the reference's own test: `if self.sp_potential:`
        if not callable(getattr(sp_potential, "V", None)):
            raise TypeError(f"sp_potential {sp_potential!r} has no V(x) (src/potentials.py:5-14)")
        extra.append(sp_potential)
    return Z, use_ho, extra


def _add_generic_potentials(r, x, extra):
    """E_loc and V of a native pass plus the user-defined potentials' V(x) (torch code of the caller, on the device)."""
    if extra:
        v = sum(p.V(x) for p in extra).to(r["eloc"].dtype)
        r["V"] = r["V"] + v
        r["eloc"] = r["eloc"] + v
    return r


# The adjoint opens with this factor x the largest step the walker's flow pass accepted (_adjoint_open; probed in round 5: 1.25 equal,
# 1.4 worse over a 3000-iteration run)
_ADJOINT_SCALE = 1.1


class _Sweep:
    """What GSVMC and BetaVMC share: flow + local energy of given base walkers with the walker schedule and the
    step-size warm start, the lazily converted device scalars, and the checkpointable sweep state."""

    def _init_sweep(self, n, dim=2):
        # None, or a dict that receives per-stage torch.cuda.Event pairs + ODE stats.  {"stages": False}: only the event pair around the
        # local-energy pass and its statistics -- every stage marker is a hipEventRecord, ~8 us of pipeline bubble on this GPU
        self.profile = None
        self.fast_backward = True    # _SweepScalar: `gradE.backward()` hands the gradient views to .grad without running the graph
        # ODE step-size warm start inside the sweep; FERMIFLOW_WARM_START=0 restores the cold start
        self.warm_start = os.environ.get("FERMIFLOW_WARM_START", "1") != "0"
        self._h_flow = None          # previous sweep's accepted flow steps: per walker (persistent walkers) or their mean
        self._dev = {}               # device scalars of the last sweep (E, E_ss, ...), converted on attribute access
        self._synced = False
        # sensitivities' first step / flow's largest step (measured, tools/probes/warm_start.py: the best factor is 0.6 up
        # to 8 particles, 0.4-0.5 at 10, 0.4 at 12; too large a factor costs a rejected step = 7 evaluations)
        self._h_scale_eloc = 0.6 if n <= 8 else (0.45 if n <= 10 else 0.4)
        # Tolerance of the sensitivity components (ff_ode.walker_class / sens_tol).  Rounds 2-4 integrated J, the Laplacian of
        # z and the Delta derivatives of the walkers of flow cost class <= 8 at 10 x rtol/atol.  Round 5 measured that policy on TRAINED
        # flows (tools/probes/policy_sweep.py, tests/golden/trained_weights.npz; 5-8 x 65 536 walkers against a 1e-11 solve, bar 1e-5):
        #                        synthetic weights   +300 it. lr 1e-4   init_zeros +300 it. lr 1e-2   +1000 it.
        #   10 x, class <= 8          4.3e-7             1.0e-5               4.0e-6                  1.3e-5
        #    5 x, class <= 6          8.1e-7             1.0e-5               4.4e-7                  2.7e-5
        #    1 x (one tolerance)      8.1e-7             1.7e-6               4.4e-7                  6.7e-6
        # -- it passed on the weights it was tuned on only: as the flow strengthens the sensitivities grow and any factor f is f times the
        # error of the plain solve in the worst walkers.  So the sweeps now keep ONE tolerance for every component (sens_tol = 1: never
        # looser than the reference's own control, src/NeuralODE/nnModule.py:161-162); what the cost classes still decide is the step a
        # walker opens with (_h_tab below) and the routing of the heavy walkers.  model.sens_tol = 10, model.sens_tol_class = 8 is the
        # old policy (FERMIFLOW_SENS_TOL sets the factor for every shape; 1 = the reference's control everywhere).
        # Beyond 12 coordinates the plain solve is 5-20 x more accurate in E_loc (larger |E_loc|, more terms to average over: max error
        # 1e-7 .. 3.7e-7 at 6 + 6 particles and 9e-8 .. 1.9e-7 at configs[4] on synthetic, trained and driver-trained flows), and one
        # tolerance costs 30-40 % more evaluations there (a third step for every walker).  Those systems keep a factor for the walkers of
        # class <= 8 -- 10 up to 24 coordinates (measured max 7.6e-7 / 4.3e-7 / 4.3e-7 at 6 + 6 particles on synthetic / trained /
        # driver-trained flows), 5 beyond (configs[4]: 1.2e-6 / 2.6e-7; 10 x: 2.4e-6 / 7.7e-7) -- tools/probes/policy_sweep.py.
        M = n * dim
        self.sens_tol = float(os.environ.get("FERMIFLOW_SENS_TOL", "1" if M <= 12 else ("10" if M <= 24 else "5")))
        self.sens_tol_class = 6 if M <= 12 else 8
        self._h_scale_loose = 0.9
        # First step of the local-energy pass = (largest step the flow pass accepted) x a factor BY COST CLASS that follows the passes
        # (ff_walker_schedule: more than _h_shrink_at of a class rejected their first step -> x 0.93; fewer than 5 % AND 70 % of its walkers with
        # three or more planned steps accepted a step of the plan one shorter -> x 1.02; the step is then rounded down to interval / k:
        # equal steps).  A fixed factor is
        # right for one set of weights only: 0.9 is accepted by 99 % of the walkers on the synthetic weights and rejected by 80 % after
        # 300 training iterations -- a whole wasted step each (29 evaluations per walker where 23 do).  Device-resident, no host round
        # trip; FERMIFLOW_ADAPTIVE_H=0 keeps the fixed factors (0.9 for class <= 6, _h_scale_eloc beyond).
        self.adaptive_h = os.environ.get("FERMIFLOW_ADAPTIVE_H", "1") != "0"
        # The rejection rate from which a class's factor shrinks, by the number G of walkers the local-energy kernel advances in lockstep
        # (a rejected first step costs the whole wave two more attempts, and at a rate p that is 1 - (1 - p)^G of the waves):
        # 1 - 0.65^(1/G) within [0.06, 0.25] -- 0.10 at the matrix-core kernel's four (measured on settled tables, tools/probes/
        # table_settle.py: trained flow 24.4 -> 23.9 evaluations per walker against 0.20, driver-1000 32.5 -> 31.2), 0.25 for the
        # one-walker-per-workgroup kernels (configs[4]: 0.10 costs 3.4 % more evaluations than 0.20 there).  Kernels by shape: csrc/
        # ff_cnf_fwd.hip dispatch_fwd (column sweep up to 3 particles: 64 / (n d) walkers per wave; matrix-core kernel 4-6: four; row
        # layout 7-8: two; one walker per wave beyond, and from 5 particles in d = 3).
        if dim == 2:
            G = min(16, 64 // (2 * n)) if n <= 3 else (4 if n <= 6 else (2 if n <= 8 else 1))
        else:
            G = max(1, 64 // (3 * n)) if n <= 4 else 1
        self._h_shrink_at = min(0.25, max(0.06, 1.0 - 0.65 ** (1.0 / G)))
        self._h_tab = None           # [2, 32] device table (double-buffered), row _h_tab_cur is current
        self._h_tab_cur = 0
        self._h_prev = None          # (cost, hs, he) of the previous local-energy pass
        self._h_counts = None        # data-parallel runs: its first-step statistics, summed over the ranks, instead
        self._h_counts_local = None
        # routing of the local-energy pass (ff_ode.heavy_class / heavy_tol / sum_weight; 0 = the library's defaults 12 (16 at 12 coordinates), 0.3, 4;
        # heavy_class < 0: no routing).  Reference semantics -- one tolerance, one kernel for every walker -- are
        # sens_tol = 1, heavy_class = -1 (FERMIFLOW_SENS_TOL=1 FERMIFLOW_HEAVY_CLASS=-1).
        self.heavy_class = int(os.environ.get("FERMIFLOW_HEAVY_CLASS", "0"))
        # ff_ode.compact_finish (None: native.eloc decides by the size of the full workspace; FERMIFLOW_COMPACT=1/0 forces it)
        self.compact_finish = {"1": True, "0": False}.get(os.environ.get("FERMIFLOW_COMPACT", ""), None)
        self.heavy_tol = 0.0         # 0: the library's defaults (ff_ode.heavy_tol 0.3, ff_ode.sum_weight 4); attributes, for probes
        self.sum_weight = 0.0

    def _mark(self, ev, name):
        if self.profile is not None and self.profile.get("stages", True):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev[name] = e

    def _first_sweep_sync(self):
        if not self._synced:
            D.sync_parameters(self)
            self._synced = True

    def _flow_and_local_energy(self, net, z, tu, td, nup, ndown, walker_state, ev, per_walker_h=False):
        """x = CNF.generate(z) and the local energy of x.
        Walker schedule (include/fermiflow.h, ff_ode.walker_cost/_order): the flow pass reports a cost class per walker
        (how close its trajectory comes to a vanishing radius); the local-energy pass, whose step count depends on exactly
        that, takes the expensive walkers first.
        Step-size warm start (ff_ode.walker_h_*): the three integrations of a sweep follow the same trajectories, so each
        one opens with the step size the previous one settled on instead of the ~20x too small Hairer start (scaled: the
        sensitivity system wants 0.4-0.6 of the flow's step depending on the particle number, the adjoint ~1.25 of the
        sensitivities').  The flow pass itself opens with 0.75 x the MEAN step the previous sweep's flow accepted (one
        number for all walkers: fresh walkers are unrelated to last sweep's) -- or, with persistent walkers, each
        chain's own.  Error control per step is unchanged."""
        prof = self.profile
        nloc = z.shape[0]
        t0, t1 = self.cnf.t_span
        cost = torch.empty(nloc, dtype=torch.int32, device=z.device)
        warm = self.warm_start
        hg = torch.empty(nloc, dtype=torch.float64, device=z.device) if warm else None
        hprev, uniform = None, False
        if warm and self._h_flow is not None and self._h_flow.device == z.device:
            if per_walker_h and self._h_flow.shape[0] == nloc:
                hprev = self._h_flow
            elif self._h_flow.numel() == 1:
                hprev, uniform = self._h_flow, True
        x = native.cnf_generate(net, z, t0, t1, self.cnf.rtol, self.cnf.atol, walker_cost=cost,
                                walker_h_init=hprev, walker_h_scale=0.75, walker_h_out=hg, walker_h_uniform=uniform)
        self.walker_cost = cost          # the flow pass's cost class per walker (diagnostics and tests; ff_ode.walker_cost)
        hs = None
        if warm and not per_walker_h and self.adaptive_h:
            if self._h_tab is None or self._h_tab.device != z.device:
                tab = torch.full((2, 32), float(self._h_scale_eloc), dtype=torch.float64, device=z.device)
                tab[:, :9] = self._h_scale_loose          # classes <= 8 open with 0.9 x the flow's step; the table takes it from there
                # (a table rebuilt for another device starts cold: the statistics of the old one do not update it)
                self._h_tab, self._h_tab_cur, self._h_prev, self._h_counts = tab, 0, None, None
            if self._h_counts is not None and self._h_counts.device != z.device:
                # a checkpoint loaded with map_location='cpu' (or rank 0's on another rank): the counts are a 64-word host-side sum, move them
                self._h_counts = self._h_counts.to(z.device)
            cur = self._h_tab_cur
            if D._active():
                # data-parallel: the table follows the statistics of the GLOBAL batch (counted per shard behind the pass, summed in the
                # estimator's all-reduce: _reduce_with_counts) -- every rank holds the same factors whatever the sharding
                order, self._h_flow, hs = native.walker_schedule(cost, hg, self._h_tab[cur], self._h_tab[1 - cur], None, interval=t1 - t0,
                                                                 counts=self._h_counts, shrink_at=self._h_shrink_at)
                self._h_counts = None
            else:
                prev = self._h_prev if (self._h_prev is not None and self._h_prev[0].shape[0] == nloc and self._h_prev[0].device == z.device) else None
                order, self._h_flow, hs = native.walker_schedule(cost, hg, self._h_tab[cur], self._h_tab[1 - cur], prev, interval=t1 - t0,
                                                                 shrink_at=self._h_shrink_at)
            self._h_tab_cur = 1 - cur
        elif warm and not per_walker_h:
            order, self._h_flow = native.walker_order(cost, hval=hg)      # the schedule and the mean accepted step from the same launches
        else:
            order = native.walker_order(cost)
            if warm:
                self._h_flow = hg
        he = torch.empty_like(hg) if warm else None
        self._mark(ev, "generate")
        p1 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if prof is not None else None
        Z, use_ho, extra = potential_plan(self.pair_potential, self.sp_potential)
        r = native.eloc(tu, td, nup, ndown, net, x, t0, t1, self.cnf.rtol, self.cnf.atol,
                        Z, use_ho, walker_state=walker_state,
                        want_stats=prof is not None, pass1_events=p1, walker_order=order,
                        walker_h_init=hs if hs is not None else hg, walker_h_scale=1.0 if hs is not None else self._h_scale_eloc, walker_h_out=he,
                        walker_class=cost if (self.sens_tol > 1.0 or self.heavy_class >= 0) else None, sens_tol=self.sens_tol,
                        sens_tol_class=self.sens_tol_class, walker_h_scale_loose=1.0 if hs is not None else self._h_scale_loose,
                        heavy_class=self.heavy_class, heavy_tol=self.heavy_tol, sum_weight=self.sum_weight,
                        compact=self.compact_finish)
        self._h_counts_local = None
        if hs is not None:
            if D._active():
                self._h_counts_local = native.scale_counts(cost, hs, he, interval=t1 - t0)
            else:
                self._h_prev = (cost, hs, he)
        _add_generic_potentials(r, x, extra)
        self._mark(ev, "eloc")
        if prof is not None:
            prof.setdefault("pass1", []).append(p1)
            prof.setdefault("eloc_stats", []).append(r["stats"])
        self._hg_last = hg      # the largest step every walker's flow pass accepted (the adjoint opens with it: _adjoint_open)
        return x, r, he

    def _adjoint_open(self, he):
        """Warm start of the adjoint pass (ff_ode.walker_h_init / _scale / _equal).  Rounds 2-4 opened it with 1.25 x the largest step
        the walker's LOCAL-ENERGY pass accepted -- but the sensitivities are the stiffest of the three systems, and once they need three
        steps of 1/3 the adjoint was opened with 0.42: never rejected, and a third step for every walker whose second step did not grow
        by 1.4 x.  The adjoint follows the flow's own trajectory with a linear costate: it opens with 1.1 x the largest step the
        walker's FLOW pass accepted, rounded down to equal steps of the interval (config 2, 200 iterations into the benchmark's
        training: 15.9 -> 14.2 evaluations per walker, no rejections; 12 particles: 13.7 -> 13.1, rejected first steps 4 % -> 0.06 %;
        configs[4]: 20.7 -> 18.9, 19 % -> 3 %; 1.3 x: 36 % rejected there -- tools/probes/adjoint_open.py)."""
        hg = getattr(self, "_hg_last", None)
        if not self.warm_start or he is None:
            return {}
        if hg is None:
            return dict(walker_h_init=he, walker_h_scale=1.25)
        return dict(walker_h_init=hg, walker_h_scale=_ADJOINT_SCALE, walker_h_equal=True)

    def _reduce_with_counts(self, buf):
        """all-reduce of the estimator's sums with the first-step statistics of this pass riding along (one collective, not two)"""
        cl = getattr(self, "_h_counts_local", None)
        if cl is None or not D._active():
            return D.all_reduce_sum_(buf)
        n = buf.numel()
        both = torch.cat([buf.reshape(-1), cl])
        D.all_reduce_sum_(both)
        self._h_counts, self._h_counts_local = both[n:], None
        buf.copy_(both[:n].view_as(buf))
        return buf

    def _scalar(self, key):
        return self._dev[key].item()

    def _std(self, key):
        n = self._n_global
        return (self._dev[key + "_ss"] / (n - 1)).sqrt().item() if n > 1 else float("nan")

    # sweep state that a checkpoint must carry for the next iteration to repeat exactly (torch.nn.Module hooks: the
    # values travel inside state_dict() under "_extra_state")
    def get_extra_state(self):
        st = {"h_flow": self._h_flow, "dev": dict(self._dev), "n_global": getattr(self, "_n_global", 0)}
        if self._h_tab is not None:      # the learned first-step factors and the pass they will next be updated from
            st["h_tab"] = self._h_tab[self._h_tab_cur].clone()
            if self._h_prev is not None:
                st["h_prev"] = tuple(self._h_prev)
            if self._h_counts is not None:
                st["h_counts"] = self._h_counts.clone()
        if getattr(self, "_z_prev", None) is not None:
            st["z_prev"] = self._z_prev
        q = getattr(self, "_z_queue", None)
        if q:      # prefetched walkers of the next iterations (GSVMC), oldest first
            for ent in q:
                if ent[1] is not None:
                    ent[1].synchronize()
            # they are THIS rank's shard: stamped with (rank, world, walker offset) so that no other rank mistakes them for its own
            rank, ws = D.world()
            st["z_queue"] = [{"z": ent[0], "seed": (int(ent[4]) if ent[4] is not None else None)} for ent in q]      # the Philox keys they were drawn with
            st["z_next_shard"] = (int(rank), int(ws), int(getattr(self.basedist, "walker_offset", 0)))
            if q[-1][3] is not None:
                st["z_next_rng"] = q[-1][3]            # torch's CPU generator right after the LAST key was drawn (_prefetched_ok)
        return st

    def set_extra_state(self, st):
        st = st or {}      # (an absent _extra_state -- a plain parameter state_dict -- is a cold sweep state)
        self._resume_seeds = []           # (a later load without prefetched walkers must not inherit an earlier load's keys)
        self._resume_rng = None
        self._h_flow = st.get("h_flow")
        self._h_tab, self._h_tab_cur, self._h_prev, self._h_counts = None, 0, None, None
        if st.get("h_tab") is not None:
            self._h_tab = torch.stack([st["h_tab"], st["h_tab"]]).contiguous()
            self._h_prev = tuple(st["h_prev"]) if st.get("h_prev") is not None else None
            hc = st.get("h_counts")
            self._h_counts = hc if (hc is not None and hc.numel() == native.SCALE_COUNTS) else None      # (64 before ABI 107: one update skipped)
        self._dev = dict(st.get("dev", {}))
        self._n_global = st.get("n_global", 0)
        if "z_prev" in st:
            self._z_prev = st["z_prev"]
        if hasattr(self, "_z_queue"):
            self._z_queue = []
            ents = st.get("z_queue")
            if ents is None and "z_next" in st:      # (checkpoints of rounds 2-4: one prefetched batch)
                ents = [{"z": st["z_next"], "seed": st.get("z_next_seed")}]
            if ents:
                # The drivers write ONE checkpoint (rank 0's) and every rank loads it: only the rank the prefetched walkers belong
                # to takes them; the others sample afresh from the restored seeds -- which is what they would have drawn anyway
                # (the Philox keys come from the restored CPU generator, the counters from the global walker index).
                rank, ws = D.world()
                shard = tuple(st.get("z_next_shard", (0, 1, 0)))
                snap = st.get("z_next_rng")
                snap = snap.cpu() if snap is not None else None
                if shard[0] == rank and shard[1] == ws:
                    self._z_queue = [(e["z"], None, int(e["z"].shape[0]), snap, e.get("seed")) for e in ents]
                else:
                    # this rank re-draws ITS shard with the same keys (forward()), under the owner's guard (_prefetched_ok): only if
                    # torch's CPU generator is where it was when the last key was drawn -- checkpoint.load puts it there; a
                    # torch.manual_seed() between the load and the sweep means fresh walkers on EVERY rank
                    seeds = [e.get("seed") for e in ents]
                    if all(sd is not None for sd in seeds):
                        self._resume_seeds, self._resume_rng = seeds, snap

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # a state_dict without "_extra_state" (weights trained with the reference, an earlier checkpoint, another model's
        # parameters) loads as a cold sweep state instead of failing strict loading
        key = prefix + "_extra_state"
        if key not in state_dict:
            state_dict[key] = {}
            try:
                return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
            finally:
                del state_dict[key]
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class GSVMC(_Sweep, torch.nn.Module):
    def __init__(self, nup, ndown, orbitals, basedist, cnf, pair_potential, sp_potential=None):
        super(GSVMC, self).__init__()
        self.orbitals_up, self.orbitals_down = orbitals.orbitals[:nup], orbitals.orbitals[:ndown]
        self.nup, self.ndown = nup, ndown
        self.basedist = basedist
        self.cnf = cnf
        self.pair_potential = pair_potential
        self.sp_potential = sp_potential
        self._init_sweep(nup + ndown, orbital_dim(tuple(self.orbitals_up) + tuple(self.orbitals_down)))
        # Persistent walkers (off by default: the reference draws fresh N(0,1) walkers and runs 100 steps in every
        # iteration, src/base_dist.py:62-64): keep the chains and advance them `persistent_steps` steps per sweep.
        self.persistent_walkers = False
        self.persistent_steps = 10
        self._z_prev = None
        # Walker prefetch (FERMIFLOW_PREFETCH=0 or model.prefetch_walkers = False: off).  The base distribution |psi_0|^2 has no
        # trainable parameter, so the NEXT iteration's Metropolis kernel does not have to wait for this iteration's update: it is
        # started on a side stream a moment after the adjoint kernel and runs beside it on the same SIMDs (292 + 164 registers;
        # DESIGN.md 6).  Same seeds in the same order, hence the same walkers as without it; the prefetched walkers travel in
        # checkpoints.
        # Measured (round 4, with the determinant-ratio samplers): it pays at 6 particles (1.67 -> 1.61 ms per iteration: the adjoint
        # and the sampler fit one SIMD together) and costs at 12 and beyond (6.65 against 6.45 ms at 12 particles, 145 against 139 ms
        # at configs[4]: both kernels fill the SIMDs on their own and only slow each other down) -- default on up to 8 particles.
        _pf = int(os.environ.get("FERMIFLOW_PREFETCH", "2" if nup + ndown <= 8 else "0"))
        self.prefetch_walkers = _pf != 0
        # Round 5: TWO iterations ahead, released behind the adjoint kernel (ff_ode.after_main_event).  The two-wave adjoint fills the
        # register files, so a sampler released beside it ran in its tail anyway -- 0.05 ms of interference at its start and 0.05 ms of
        # the next iteration waiting for the walkers.  Released when the adjoint kernel ends, the sampler runs under the small
        # latency-bound kernels behind it (table reduction, contraction, Adam, next radial table: ~150 us of mostly idle SIMDs), and
        # with a batch in hand nobody ever waits for it.  FERMIFLOW_PREFETCH=<n> keeps n batches ahead (0: off).
        self.prefetch_depth = max(1, _pf)
        self._z_queue = []           # entries (walkers, event on the side stream, nloc, CPU generator state after their seed was drawn, seed)
        self._resume_seeds, self._resume_rng = [], None
        self._side = None

    # energy estimate of the last forward() (python floats as in the reference, src/VMC.py:57; read lazily from the device)
    @property
    def E(self):
        return self._scalar("E")

    @property
    def E_std(self):
        return self._std("E")

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
        Z, use_ho, extra = potential_plan(self.pair_potential, self.sp_potential)
        r = native.eloc(tu, td, self.nup, self.ndown, self.cnf.v_wrapper.v.net(), x.detach(), t0, t1,
                        self.cnf.rtol, self.cnf.atol, Z, use_ho, walker_state=walker_state, want_stats=want_stats)
        return _add_generic_potentials(r, x.detach(), extra)

    def _native_grad_laplacian(self, x):
        r = self.local_energy(x)
        return r["logp"], r["grad"], r["lap"]

    # -- the sweep ------------------------------------------------------------------------------------
    def forward(self, batch):
        rank, ws = D.world()
        offset, nloc = D.shard(batch, rank, ws)
        self.basedist.walker_offset = offset
        ev = {}
        self._mark(ev, "t0")
        with torch.no_grad():
            if self.persistent_walkers and self._z_prev is not None and self._z_prev.shape[0] == nloc:
                # opt-in: continue the previous sweep's chains for a few steps instead of 100 steps from N(0,1)
                z = self.basedist.sample(self.orbitals_up, self.orbitals_down, (nloc,),
                                         equilibrim_steps=self.persistent_steps, x_init=self._z_prev)
            elif not self._z_queue and self._resume_ok():
                # resumed on a rank other than the one whose prefetched walkers the checkpoint holds: the same Philox keys and this
                # rank's walker offset give exactly the walkers the uninterrupted run prefetched here (the CPU generator is not
                # touched) -- all of them now, so that from here on this rank is where the owner is
                snap = self._resume_rng
                self._z_queue = [(self.basedist.sample(self.orbitals_up, self.orbitals_down, (nloc,), seed=sd), None, nloc, snap, sd)
                                 for sd in self._resume_seeds]
                z = self._z_queue.pop(0)[0]
            elif self._prefetched_ok(nloc):
                z, done = self._z_queue[0][0], self._z_queue[0][1]          # sampled behind an earlier iteration's adjoint
                self._z_queue.pop(0)
                if done is not None:
                    torch.cuda.current_stream().wait_event(done)
                    z.record_stream(torch.cuda.current_stream())
            else:
                self._z_queue = []
                z = self.basedist.sample(self.orbitals_up, self.orbitals_down, (nloc,))
            self._resume_seeds, self._resume_rng = [], None
            self._z_prev = z if self.persistent_walkers else None
        self._mark(ev, "mcmc")
        return self._sweep(z, batch, ev, prefetch=nloc if (self.prefetch_walkers and not self.persistent_walkers and z.is_cuda) else 0)

    def _prefetched_ok(self, nloc):
        """The prefetched walkers stand for "sample now" only if nothing touched torch's CPU generator since their seed was drawn
        from it: after a torch.manual_seed() (or any draw) in between they are dropped and fresh ones are sampled, so re-seeding
        between iterations means what it means without the prefetch."""
        q = self._z_queue
        if not q or self.persistent_walkers or any(ent[2] != nloc for ent in q):
            return False
        snap = q[-1][3]      # (the generator right after the LAST key was drawn: the earlier entries' draws are its own history)
        return snap is None or torch.equal(snap, torch.get_rng_state())

    # (rounds 2-4 kept ONE prefetched batch under these names; tests and diagnostics still read them)
    @property
    def _z_next(self):
        return self._z_queue[0] if self._z_queue else None

    @property
    def _resume_seed(self):
        return self._resume_seeds[0] if self._resume_seeds else None

    def _resume_ok(self):
        """A rank that resumes from another rank's checkpoint re-draws its shard with the checkpointed Philox keys -- under the same
        condition as _prefetched_ok: nobody touched torch's CPU generator since the last key was drawn."""
        if not self._resume_seeds or self.persistent_walkers:
            return False
        snap = self._resume_rng
        return snap is None or torch.equal(snap, torch.get_rng_state())

    def _prefetch(self, nloc, go):
        """Walkers of the iterations ahead on the side stream, until prefetch_depth batches are in hand: released when `go` fires -- the
        event the adjoint call records behind its main kernel (ff_ode.after_main_event)."""
        if self._side is None:
            self._side = torch.cuda.Stream()
        from .base_dist import _draw_seed
        with torch.cuda.stream(self._side):
            self._side.wait_event(go)
            while len(self._z_queue) < self.prefetch_depth:
                seed = _draw_seed()
                z = self.basedist.sample(self.orbitals_up, self.orbitals_down, (nloc,), seed=seed)
                done = torch.cuda.Event()
                done.record()
                self._z_queue.append((z, done, nloc, None, seed))
        snap = torch.get_rng_state()
        self._z_queue = [(e[0], e[1], e[2], snap, e[4]) for e in self._z_queue]

    def forward_from(self, z, batch=None):
        """forward() on GIVEN base walkers z (nloc, n, 2) -- this rank's shard of a global batch of `batch` walkers
        (default: z is the whole batch).  Everything after the Metropolis sampling is the code forward() runs."""
        ev = {}
        self._mark(ev, "t0")
        self._mark(ev, "mcmc")
        return self._sweep(z.detach().contiguous(), int(batch if batch is not None else z.shape[0]), ev)

    def _sweep(self, z, batch, ev, prefetch=0):
        self._first_sweep_sync()
        prof = self.profile
        with torch.no_grad():
            t0, t1 = self.cnf.t_span
            v, params = _flow_params(self.cnf)
            net = v.net(refresh=True)
            tu, td = self._tables(z.device)
            x, r, he = self._flow_and_local_energy(net, z, tu, td, self.nup, self.ndown, None, ev,
                                                   per_walker_h=self.persistent_walkers)
            Eloc = r["eloc"]
            # E, E_std and the surrogate's value (src/VMC.py:56-59) from ONE pass over (E_loc, logp) and one all-reduce of four
            # sums, taken about the previous sweep's mean (no cancellation); the adjoint forms its seeds (E_loc - E) / batch
            # from E on the device (ff_cnf_adjoint_energy) -- nothing here waits for the host
            prev = self._dev.get("E")
            shift = prev.reshape(1) if prev is not None else torch.zeros(1, dtype=Eloc.dtype, device=Eloc.device)
            if not D._active():
                _, est = native.energy_estimate(Eloc, r["logp"], shift, batch)       # one launch: sums and finish (nothing to all-reduce)
            else:
                sums, _ = native.energy_estimate(Eloc, r["logp"], shift, 0)
                self._reduce_with_counts(sums)
                est = native.energy_finish(sums, shift, batch)    # [E, sum (e - E)^2, mean(logp (e - E))]
            self._dev["E"], self._dev["E_ss"], self._n_global = est[0], est[1], batch
            self._mark(ev, "estimator")
            go, after = None, None
            if prefetch:      # the sampler's side stream is released by the library right behind the adjoint's main kernel (ff_ode.after_main_event)
                go = torch.cuda.Event()
                if self._side is None:
                    self._side = torch.cuda.Stream()
                go.record(self._side)      # (creates the handle; the library re-records it on the main stream)
                after = go
            adj = native.cnf_adjoint(net, r["z"], r["glogp0"], None, t0, t1, self.cnf.rtol, self.cnf.atol,
                                     need_gx=False, energy=(Eloc, est, 1.0 / batch),   # (uniform cost: no schedule)
                                     after_main_event=after, want_stats=prof is not None, **self._adjoint_open(he))
            gp = adj[1]
            if prof is not None:
                prof.setdefault("adjoint_stats", []).append(adj[2])
            if prefetch:
                self._prefetch(prefetch, go)
            D.all_reduce_sum_(gp)
            self._mark(ev, "adjoint")
        if prof is not None:
            prof.setdefault("events", []).append(ev)
        self.Eloc, self.x = Eloc, x
        return _sweep_scalar(est[2], gp, params, self.fast_backward)


def draw_states_reference(logits, batch):
    """The reference's state draw (src/VMC.py:90-96): `Categorical(logits=log_state_weights).sample((batch,))` on torch's default (CPU)
    generator in the reference's fp64, then sorted -- the list `Counter(sorted(state_indices.tolist()))` counts.  Returns int64 on the CPU."""
    from torch.distributions.categorical import Categorical
    idx = Categorical(logits=logits.detach().to("cpu", torch.float64)).sample((int(batch),))
    return torch.sort(idx).values


class BetaVMC(_Sweep, torch.nn.Module):
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
        self._init_sweep(nup + ndown, orbital_dim(tuple(self.states[0][0]) + tuple(self.states[0][1])))
        self._ws = None

    E = property(lambda self: self._scalar("E"))
    E_std = property(lambda self: self._std("E"))
    F = property(lambda self: self._scalar("F"))
    F_std = property(lambda self: self._std("F"))
    S = property(lambda self: self._scalar("S"))
    S_analytical = property(lambda self: self._scalar("S_analytical"))

    # the reference's Counter {state index: number of walkers} of this rank's shard (src/VMC.py:96), built on demand
    @property
    def state_indices_collection(self):
        if self._coll is None:
            self._coll = Counter(self._ws.tolist())
        return self._coll

    @state_indices_collection.setter
    def state_indices_collection(self, coll):
        self._coll = Counter(dict(coll))

    _coll = None

    # How the many-body states of a batch are drawn: "order_statistics" (default; below) or "reference" -- the reference's own draw,
    # Categorical(logits).sample((batch,)) on torch's CPU generator, sorted (src/VMC.py:90-96): after the same torch.manual_seed the state
    # list IS the reference's, so a seeded BetaVMC.forward can be compared with it end to end (VERDICT r05 missing #3).  A host round trip
    # per sweep: opt-in.
    state_draw = "order_statistics"

    def _draw_states(self, batch):
        """Many-body state of every walker of the GLOBAL batch, sorted as the reference does (src/VMC.py:94-96), on the
        device; rank 0 draws and broadcasts so that every rank cuts its shard from the same list."""
        logits = self.log_state_weights.detach()
        if self.state_draw == "reference":
            idx = draw_states_reference(logits, batch).to(logits.device)
            D.broadcast_(idx)
            return idx
        if self.state_draw != "order_statistics":
            raise ValueError(f"BetaVMC.state_draw must be 'order_statistics' or 'reference', not {self.state_draw!r}")
        # sorted(Categorical(logits).sample((batch,))) without the sort: the order statistics of `batch` uniforms are the
        # normalised partial sums of batch + 1 exponentials, and the inverse CDF is monotone -- so pushing the (already sorted)
        # uniforms through it gives the sorted state list directly (one scan instead of multinomial + radix/merge sort)
        g = torch.empty(int(batch) + 1, dtype=torch.float64, device=logits.device).exponential_()
        c = torch.cumsum(g, 0)
        u = c[:-1] / c[-1]
        cdf = torch.cumsum(torch.softmax(logits.double(), dim=0), 0)
        idx = torch.bucketize(u, cdf[:-1], right=True)
        D.broadcast_(idx)
        return idx

    def _set_shard(self, idx_global):
        rank, world = D.world()
        off, cnt = D.shard(idx_global.numel(), rank, world)
        self._ws = idx_global[off:off + cnt].to(device=self.basedist.device, dtype=torch.int32).contiguous()
        self._coll = None
        self._nglobal = idx_global.numel()
        self.basedist.walker_offset = off
        return cnt

    def sample(self, sample_shape, nframes=None):
        from torch.distributions.categorical import Categorical
        self.state_dist = Categorical(logits=self.log_state_weights)
        batch = 1
        for s in sample_shape:
            batch *= int(s)
        cnt = self._set_shard(self._draw_states(batch))
        z = self._sample_base(cnt)
        x = self.cnf.generate(z, nframes=nframes)
        return z, x

    def _sample_base(self, cnt):
        from .base_dist import _draw_seed
        tu, td = self._state_tables(self.basedist.device)
        z, _, _ = native.mcmc_sample(tu, td, self.nup, self.ndown, cnt, 100, 0.1, _draw_seed(), self.basedist.device,
                                     walker_offset=self.basedist.walker_offset, walker_state=self._ws)
        return z

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
        Z, use_ho, extra = potential_plan(self.pair_potential, self.sp_potential)
        r = native.eloc(tu, td, self.nup, self.ndown, self.cnf.v_wrapper.v.net(), x.detach(), t0, t1,
                        self.cnf.rtol, self.cnf.atol, Z, use_ho, walker_state=walker_state)
        return _add_generic_potentials(r, x.detach(), extra)

    def _native_grad_laplacian(self, x):
        r = self.local_energy(x, self._walker_state(x.device))
        return r["logp"], r["grad"], r["lap"]

    def _walker_state(self, device):
        return self._ws.to(device)

    def forward(self, batch):
        """Finite-temperature sweep (src/VMC.py:114-171).  Under torch.distributed `batch` is the global walker count;
        the per-state sums and both gradients are all-reduced (2*Nstates + 4 doubles, then 1 + 3(He+Hm))."""
        self._first_sweep_sync()
        ev = {}
        self._mark(ev, "t0")
        with torch.no_grad():
            cnt = self._set_shard(self._draw_states(batch))
            z = self._sample_base(cnt)
        self._mark(ev, "mcmc")
        return self._sweep(z, ev)

    def forward_from(self, z, state_indices, batch=None):
        """forward() on GIVEN base walkers z (nloc, n, 2) in the given many-body states (int tensor / sequence, sorted as
        the reference keeps them; this rank's shard of a global batch of `batch` walkers)."""
        self._first_sweep_sync()
        ws = torch.as_tensor(state_indices).to(device=self.basedist.device, dtype=torch.int32).contiguous()
        self._ws, self._coll = ws, None
        self._nglobal = int(batch if batch is not None else ws.numel())
        ev = {}
        self._mark(ev, "t0")
        self._mark(ev, "mcmc")
        return self._sweep(z.detach().contiguous(), ev)

    def _sweep(self, z, ev):
        prof = self.profile
        with torch.no_grad():
            device = z.device
            ws, nglob, Ns = self._ws, self._nglobal, self.Nstates
            t0, t1 = self.cnf.t_span
            v, params = _flow_params(self.cnf)
            net = v.net(refresh=True)
            tu, td = self._state_tables(device)
            x, r, he = self._flow_and_local_energy(net, z, tu, td, self.nup, self.ndown, ws, ev)
            Eloc = r["eloc"]
            # Estimator (src/VMC.py:146-171) in two launches around ONE all-reduce: moments of E_loc about the previous sweep's
            # mean + partial per-state sums of (E_loc, 1, logp, logp E_loc); ff_beta_finish turns the totals into E, F, S, both
            # surrogates' values, the logits gradient (closed form) and the per-state baseline, which the adjoint reads by
            # state index when it forms its seeds (ff_cnf_adjoint_energy) -- no per-walker torch arithmetic, no host round trip
            prev = self._dev.get("E")
            shE = prev.reshape(1) if prev is not None else torch.zeros(1, dtype=Eloc.dtype, device=device)
            buf1 = native.beta_buffer(Ns, device)
            native.reduce_moments(Eloc, shift_dev=shE, out=buf1[:2])
            native.beta_state_partials(Eloc, r["logp"], ws, Ns, buf1)
            self._reduce_with_counts(buf1)
            logits = self.log_state_weights.detach().to(device)
            est, g_phi, mean_e, logp_all = native.beta_finish(buf1, shE, logits, self.beta, nglob)
            for k, key in enumerate(("E", "E_ss", "F", "F_ss", "S", "S_analytical")):
                self._dev[key] = est[k]
            self._n_global = nglob
            self.logp_states_all = logp_all.to(self.log_state_weights.device)
            self._mark(ev, "estimator")
            _, gp = native.cnf_adjoint(net, r["z"], r["glogp0"], None, t0, t1, self.cnf.rtol, self.cnf.atol, need_gx=False,
                                       energy=(Eloc, mean_e, 1.0 / nglob, ws), **self._adjoint_open(he))
            D.all_reduce_sum_(gp)
            self._mark(ev, "adjoint")
        if prof is not None:
            prof.setdefault("events", []).append(ev)
        self.Eloc, self.x = Eloc, x
        pdev = self.log_state_weights.device
        gradF_phi = _sweep_scalar(est[6].to(pdev), g_phi.to(pdev), [self.log_state_weights], self.fast_backward)
        gradF_theta = _sweep_scalar(est[7], gp, params, self.fast_backward)
        return gradF_phi, gradF_theta
