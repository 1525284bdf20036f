"""Tensor-level wrappers over the C ABI (include/fermiflow.h).  Every function takes/returns fp64 CUDA
tensors, enqueues on torch's current stream and never synchronises."""
import ctypes as C
import os

import torch

from . import _lib as L


_TABLES = {}   # (orbital indices, device) -> device table; a host->device copy is a synchronisation point, so tables
               # are uploaded once (they are a few integers that never change for a model)


def orbital_table(idx, device):
    """int32 device table [n_states, n_spin] from a (list of) orbital index list(s); cached per device."""
    t = torch.as_tensor(idx, dtype=torch.int32)
    if t.dim() == 1:
        t = t[None]
    key = (tuple(t.shape), tuple(t.reshape(-1).tolist()), str(torch.device(device)))
    tab = _TABLES.get(key)
    if tab is None:
        if len(_TABLES) > 256:
            _TABLES.clear()
        tab = _TABLES[key] = t.contiguous().to(device)
    return tab


def set_kernel_family(family):
    """ff_set_kernel_family: 0 = by particle number (default), 1 = one walker per workgroup for every particle number.
    Returns the previous setting."""
    return int(L.lib().ff_set_kernel_family(int(family)))


def set_sens_precision(bits):
    """ff_set_sens_precision: 64 (default) or 32 = single-precision sensitivity matrices in the matrix-core local-energy kernel
    (11 particles and more).  Returns the previous setting."""
    return int(L.lib().ff_set_sens_precision(int(bits)))


def _state(ws):
    return None if ws is None else L.dev(ws, torch.int32, "walker_state")


def slater_fwd(table, x, walker_state=None):
    x = L.dev(x, name="x")
    B, n = x.shape[0], x.shape[1]
    out = torch.empty(B, dtype=torch.float64, device=x.device)
    L.check(L.lib().ff_slater_logabsdet_fwd(L.stream(), L.i64(B), n, L.ptr(table), L.ptr(_state(walker_state)), L.ptr(x),
                                            L.ptr(out)), "ff_slater_logabsdet_fwd")
    return out


def slater_bwd(table, x, grad_out, walker_state=None):
    x = L.dev(x, name="x"); go = L.dev(grad_out, name="grad_out")
    B, n = x.shape[0], x.shape[1]
    gx = torch.empty_like(x)
    L.check(L.lib().ff_slater_logabsdet_bwd(L.stream(), L.i64(B), n, L.ptr(table), L.ptr(_state(walker_state)), L.ptr(x),
                                            L.ptr(go), L.ptr(gx)), "ff_slater_logabsdet_bwd")
    return gx


def logprob(tab_up, tab_dn, nup, ndn, x, walker_state=None, derivs=False):
    x = L.dev(x, name="x")
    B = x.shape[0]
    logp = torch.empty(B, dtype=torch.float64, device=x.device)
    grad = torch.empty_like(x) if derivs else None
    lap = torch.empty(B, dtype=torch.float64, device=x.device) if derivs else None
    L.check(L.lib().ff_logprob(L.stream(), L.i64(B), nup, ndn, L.ptr(tab_up), L.ptr(tab_dn), L.ptr(_state(walker_state)),
                               L.ptr(x), L.ptr(logp), L.ptr(grad), L.ptr(lap)), "ff_logprob")
    return (logp, grad, lap) if derivs else logp


def mcmc_sample(tab_up, tab_dn, nup, ndn, B, steps, tau, seed, device, walker_offset=0, walker_state=None):
    n = nup + ndn
    x = torch.empty(B, n, 2, dtype=torch.float64, device=device)
    logp = torch.empty(B, dtype=torch.float64, device=device)
    cnt = torch.empty(B, dtype=torch.int32, device=device)
    L.check(L.lib().ff_mcmc_sample(L.stream(), L.i64(B), nup, ndn, L.ptr(tab_up), L.ptr(tab_dn), L.ptr(_state(walker_state)),
                                   int(steps), L.f64(tau), C.c_uint64(int(seed) & (2**64 - 1)), L.i64(walker_offset),
                                   L.ptr(x), L.ptr(logp), L.ptr(cnt)), "ff_mcmc_sample")
    return x, logp, cnt


def mcmc_continue(tab_up, tab_dn, nup, ndn, x_init, steps, tau, seed, walker_offset=0, walker_state=None):
    """ff_mcmc_continue: `steps` more Metropolis steps from the walkers x_init (B,n,2)."""
    x0 = L.dev(x_init, name="x_init")
    B = x0.shape[0]
    x = torch.empty_like(x0)
    logp = torch.empty(B, dtype=torch.float64, device=x0.device)
    cnt = torch.empty(B, dtype=torch.int32, device=x0.device)
    L.check(L.lib().ff_mcmc_continue(L.stream(), L.i64(B), nup, ndn, L.ptr(tab_up), L.ptr(tab_dn), L.ptr(_state(walker_state)),
                                     int(steps), L.f64(tau), C.c_uint64(int(seed) & (2**64 - 1)), L.i64(walker_offset),
                                     L.ptr(x0), L.ptr(x), L.ptr(logp), L.ptr(cnt)), "ff_mcmc_continue")
    return x, logp, cnt


def mcmc_sample_noise(tab_up, tab_dn, nup, ndn, g0, g, u, tau=0.1, walker_state=None):
    g0, g, u = L.dev(g0, name="g0"), L.dev(g, name="g"), L.dev(u, name="u")
    B, steps = g0.shape[0], g.shape[0]
    x = torch.empty_like(g0)
    logp = torch.empty(B, dtype=torch.float64, device=g0.device)
    acc = torch.empty(steps, B, dtype=torch.uint8, device=g0.device)
    L.check(L.lib().ff_mcmc_sample_noise(L.stream(), L.i64(B), nup, ndn, L.ptr(tab_up), L.ptr(tab_dn),
                                         L.ptr(_state(walker_state)), int(steps), L.f64(tau), L.ptr(g0), L.ptr(g), L.ptr(u),
                                         L.ptr(x), L.ptr(logp), L.ptr(acc)), "ff_mcmc_sample_noise")
    return x, logp, acc


def rng_fill(B, n, steps, seed, device, walker_offset=0, dim=2):
    """the noise ff_mcmc_sample (dim = 2) / ff_mcmc_sample3d (dim = 3) consume for these walkers, materialised"""
    g0 = torch.empty(B, n, dim, dtype=torch.float64, device=device)
    g = torch.empty(steps, B, n, dim, dtype=torch.float64, device=device)
    u = torch.empty(steps, B, dtype=torch.float64, device=device)
    fn, name = (L.lib().ff_rng_fill, "ff_rng_fill") if dim == 2 else (L.lib().ff_rng_fill3d, "ff_rng_fill3d")
    L.check(fn(L.stream(), L.i64(B), n, int(steps), C.c_uint64(int(seed) & (2**64 - 1)), L.i64(walker_offset),
               L.ptr(g0), L.ptr(g), L.ptr(u)), name)
    return g0, g, u


def mlp_eval(w1, b1, w2, r, need_grad=True):
    r = L.dev(r, name="r")
    w1, b1, w2 = L.dev(w1.reshape(-1)), L.dev(b1), L.dev(w2.reshape(-1))
    val = torch.empty_like(r)
    dval = torch.empty_like(r) if need_grad else None
    L.check(L.lib().ff_mlp_eval(L.stream(), L.i64(r.numel()), b1.numel(), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(r),
                                L.ptr(val), L.ptr(dval)), "ff_mlp_eval")
    return val, dval


def mlp_eval_nd(w1, b1, w2, x, need_val=True, need_grad=False):
    """ff_mlp_eval_nd: x (N, D_in) -> val (N), grad (N, D_in)."""
    x = L.dev(x, name="x")
    N, Din = x.shape
    w1, b1, w2 = L.dev(w1.reshape(-1, Din)), L.dev(b1), L.dev(w2.reshape(-1))
    val = torch.empty(N, dtype=torch.float64, device=x.device) if need_val else None
    grad = torch.empty_like(x) if need_grad else None
    L.check(L.lib().ff_mlp_eval_nd(L.stream(), L.i64(N), int(Din), b1.numel(), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(x), L.ptr(val), L.ptr(grad)),
            "ff_mlp_eval_nd")
    return val, grad


def backflow_vjp(net, x, w=None, need_gdiv=False):
    """ff_backflow_vjp: ((dv/dx)^T w or None, grad_x div v or None)."""
    x = L.dev(x, name="x")
    B, n, d = x.shape
    w = L.dev(w, name="w") if w is not None else None
    Aw = torch.empty_like(x) if w is not None else None
    gd = torch.empty_like(x) if need_gdiv else None
    L.check(L.lib().ff_backflow_vjp(L.stream(), L.i64(B), n, d, net.ref(), L.ptr(x), L.ptr(w), L.ptr(Aw), L.ptr(gd)), "ff_backflow_vjp")
    return Aw, gd


def backflow_v_div(net, x, need_v=True, need_div=True):
    x = L.dev(x, name="x")
    B, n, d = x.shape
    v = torch.empty_like(x) if need_v else None
    div = torch.empty(B, dtype=torch.float64, device=x.device) if need_div else None
    L.check(L.lib().ff_backflow_v_div(L.stream(), L.i64(B), n, d, net.ref(), L.ptr(x), L.ptr(v), L.ptr(div)),
            "ff_backflow_v_div")
    return v, div


def potential(x, Z, use_ho):
    x = L.dev(x, name="x")
    B, n, d = x.shape
    V = torch.empty(B, dtype=torch.float64, device=x.device)
    L.check(L.lib().ff_potential(L.stream(), L.i64(B), n, d, L.f64(Z), int(bool(use_ho)), L.ptr(x), L.ptr(V)), "ff_potential")
    return V


def _stats(device, want):
    # [0..3] stats, [8..] diagnostic stamps (FF_STAMPS builds; FF_STATS_WORDS makes room for their per-workgroup trace)
    return torch.zeros(int(os.environ.get("FF_STATS_WORDS", "32")), dtype=torch.int32, device=device) if want else None


def cnf_generate(net, z, t0, t1, rtol, atol, want_stats=False, walker_cost=None, walker_order=None, **warm):
    z = L.dev(z, name="z")
    B, n, d = z.shape
    x = torch.empty_like(z)
    st = _stats(z.device, want_stats)
    o = L.ode(t0, t1, rtol, atol, walker_cost=walker_cost, walker_order=walker_order, **warm)
    L.check(L.lib().ff_cnf_generate(L.stream(), L.i64(B), n, d, net.ref(), C.byref(o), L.ptr(z), L.ptr(x), L.ptr(st)),
            "ff_cnf_generate")
    return (x, st) if want_stats else x


def cnf_delta_logp(net, x, t0, t1, rtol, atol, want_stats=False, walker_cost=None, walker_order=None, **warm):
    x = L.dev(x, name="x")
    B, n, d = x.shape
    z = torch.empty_like(x)
    dl = torch.empty(B, dtype=torch.float64, device=x.device)
    st = _stats(x.device, want_stats)
    o = L.ode(t0, t1, rtol, atol, walker_cost=walker_cost, walker_order=walker_order, **warm)
    L.check(L.lib().ff_cnf_delta_logp(L.stream(), L.i64(B), n, d, net.ref(), C.byref(o), L.ptr(x), L.ptr(z), L.ptr(dl),
                                      L.ptr(st)), "ff_cnf_delta_logp")
    return (z, dl, st) if want_stats else (z, dl)


def cnf_adjoint(net, y_start, a_z, a_d, t_from, t_to, rtol, atol, need_gx=True, want_stats=False, walker_cost=None,
                walker_order=None, energy=None, **warm):
    """Adjoint sweep from t_from (where y_start, a_z, a_d are given) to t_to.
    energy = (eloc (B,), e_mean device tensor, scale[, mean_index int32 (B,)]): ff_cnf_adjoint_energy -- a_z is glogp0
    (B,n,d) and the kernel forms the seeds w_b * glogp0[b], -w_b with w_b = (eloc[b] - e_mean[mean_index[b] or 0]) * scale
    itself (a_d is ignored)."""
    y = L.dev(y_start, name="y_start"); a_z = L.dev(a_z, name="a_z")
    if energy is None:
        a_d = L.dev(a_d, name="a_d")
    B, n, d = y.shape
    gx = torch.empty_like(y) if need_gx else None
    gp = torch.empty(net.nparams, dtype=torch.float64, device=y.device)
    nbytes = L.lib().ff_cnf_adjoint_workspace_bytes(L.i64(B), n, d, net.He, net.Hm)
    ws = torch.empty(max(1, nbytes // 8), dtype=torch.float64, device=y.device)
    st = _stats(y.device, want_stats)
    o = L.ode(t_from, t_to, rtol, atol, walker_cost=walker_cost, walker_order=walker_order, **warm)
    if energy is not None:
        e = L.dev(energy[0], name="eloc")
        mi = L.dev(energy[3], torch.int32, "mean_index") if len(energy) > 3 and energy[3] is not None else None
        em = L.dev(energy[1].reshape(-1) if mi is not None else energy[1].reshape(-1)[:1], name="e_mean")
        L.check(L.lib().ff_cnf_adjoint_energy(L.stream(), L.i64(B), n, d, net.ref(), C.byref(o), L.ptr(y), L.ptr(a_z), L.ptr(e), L.ptr(em),
                                              L.ptr(mi), L.f64(energy[2]), L.ptr(gx), L.ptr(gp), L.ptr(ws), L.ptr(st)), "ff_cnf_adjoint_energy")
    else:
        L.check(L.lib().ff_cnf_adjoint(L.stream(), L.i64(B), n, d, net.ref(), C.byref(o), L.ptr(y), L.ptr(a_z), L.ptr(a_d),
                                       L.ptr(gx), L.ptr(gp), L.ptr(ws), L.ptr(st)), "ff_cnf_adjoint")
    return (gx, gp, st) if want_stats else (gx, gp)


COMPACT_WORKSPACE_BYTES = 32 << 30     # eloc(compact=None): beyond this many bytes of full workspace, ask for the compact one


def eloc(tab_up, tab_dn, nup, ndn, net, x, t0, t1, rtol, atol, Z, use_ho, walker_state=None, want_stats=False,
         pass1_events=None, walker_cost=None, walker_order=None, two_pass=False, compact=None, **warm):
    """ff_eloc.  pass1_events: optional (start, end) torch.cuda.Event pair recorded around the sensitivity pass
    (bench.py times the dominant kernel with it).  two_pass: ff_eloc_sensitivities + ff_eloc_finish as separate calls with the
    sensitivities in the workspace between them (the path every kernel without a fused finish takes anyway).
    compact: ff_ode::compact_finish -- the one-walker-per-workgroup kernels finish their walkers themselves and the workspace
    beyond 24 coordinates is z(t0) | Delta only; slower by 6-8 % (include/fermiflow.h), so the default (None) asks for it only when
    the full workspace would pass COMPACT_WORKSPACE_BYTES."""
    x = L.dev(x, name="x")
    B, n, d = x.shape[0], nup + ndn, x.shape[2]
    f = dict(dtype=torch.float64, device=x.device)
    M = n * d
    full_bytes = L.lib().ff_eloc_workspace_bytes(L.i64(B), n, d)
    if compact is None:
        compact = M > 24 and full_bytes > COMPACT_WORKSPACE_BYTES
    if two_pass and compact:
        raise ValueError("two_pass keeps the sensitivities in the workspace: not with compact=True")
    compact_layout = bool(compact) and M > 24       # ff_eloc_nd beyond 24 coordinates: z(t0) | Delta only (include/fermiflow.h)
    nbytes = L.lib().ff_eloc_nd_workspace_bytes(L.i64(B), n, d, int(bool(compact))) if not two_pass else full_bytes
    ws = torch.empty(max(1, nbytes // 8), **f)
    # z(t0) and Delta stay where the sensitivity pass leaves them: views of the workspace (its documented head, include/fermiflow.h)
    # instead of two device-to-device copies per sweep
    dl0 = B * M if compact_layout else B * (M * M + 4 * M)
    out = dict(logp=torch.empty(B, **f), grad=torch.empty_like(x), lap=torch.empty(B, **f), V=torch.empty(B, **f),
               eloc=torch.empty(B, **f), z=ws[:B * M].view(B, n, d), dlogp=ws[dl0:dl0 + B],
               glogp0=torch.empty_like(x))
    st = _stats(x.device, want_stats)
    o = L.ode(t0, t1, rtol, atol, walker_cost=walker_cost, walker_order=walker_order, compact_finish=bool(compact), **warm)
    if pass1_events is not None:
        pass1_events[0].record()
    if not two_pass:
        # one call: the library fuses the finish into the sensitivity kernel where that kernel implements it (then the second
        # event of pass1_events closes the whole pass)
        L.check(L.lib().ff_eloc_nd(L.stream(), L.i64(B), nup, ndn, int(d), L.ptr(tab_up), L.ptr(tab_dn), L.ptr(_state(walker_state)), net.ref(),
                                   C.byref(o), L.f64(Z), int(bool(use_ho)), L.ptr(x), L.ptr(out["logp"]), L.ptr(out["grad"]),
                                   L.ptr(out["lap"]), L.ptr(out["V"]), L.ptr(out["eloc"]), None, None, L.ptr(out["glogp0"]), L.ptr(ws),
                                   L.ptr(st)), "ff_eloc_nd")
        if pass1_events is not None:
            pass1_events[1].record()
    else:
        L.check(L.lib().ff_eloc_sensitivities(L.stream(), L.i64(B), n, d, net.ref(), C.byref(o), L.ptr(x), L.ptr(ws), L.ptr(st)),
                "ff_eloc_sensitivities")
        if pass1_events is not None:
            pass1_events[1].record()
        finish = L.lib().ff_eloc_finish3d if d == 3 else L.lib().ff_eloc_finish      # d = 3: HO3D orbital tables (csrc/ff_ho3d.hip)
        L.check(finish(L.stream(), L.i64(B), nup, ndn, L.ptr(tab_up), L.ptr(tab_dn), L.ptr(_state(walker_state)),
                       L.f64(Z), int(bool(use_ho)), L.ptr(x), L.ptr(ws), L.ptr(out["logp"]), L.ptr(out["grad"]),
                       L.ptr(out["lap"]), L.ptr(out["V"]), L.ptr(out["eloc"]), None,
                       None, L.ptr(out["glogp0"])), "ff_eloc_finish")
    if want_stats:
        out["stats"] = st
    return out


def walker_order(cost, hval=None):
    """ff_walker_order: int32 permutation, most expensive walkers first (cost: int32 per-walker step counts).
    hval (float64, B): ff_walker_order_mean -- returns (order, 1-element tensor mean(hval)) from the same two launches."""
    cost = cost.contiguous()
    if cost.dtype != torch.int32 or not cost.is_cuda:
        raise ValueError("cost must be an int32 device tensor")
    order = torch.empty_like(cost)
    ws = torch.empty(max(1, (L.lib().ff_walker_order_workspace_bytes(L.i64(cost.numel())) + 7) // 8), dtype=torch.float64, device=cost.device)
    if hval is None:
        L.check(L.lib().ff_walker_order(L.stream(), L.i64(cost.numel()), L.ptr(cost), L.ptr(order), L.ptr(ws)), "ff_walker_order")
        return order
    hval = L.dev(hval, name="hval")
    hmean = torch.empty(1, dtype=torch.float64, device=cost.device)
    L.check(L.lib().ff_walker_order_mean(L.stream(), L.i64(cost.numel()), L.ptr(cost), L.ptr(order), L.ptr(ws), L.ptr(hval), L.ptr(hmean)),
            "ff_walker_order_mean")
    return order, hmean


def walker_schedule(cost, hval, scale_in, scale_out, prev=None, interval=0.0, counts=None, shrink_at=0.0):
    """ff_walker_schedule: (order, mean(hval) as a 1-element tensor, hs) with hs[b] = hval[b] * scale_in[cost[b]] -- the step every walker's
    local-energy pass opens with; scale_out receives the table updated from the previous pass prev = (cost, hs, he) (None: copied)."""
    cost = cost.contiguous()
    if cost.dtype != torch.int32 or not cost.is_cuda:
        raise ValueError("cost must be an int32 device tensor")
    hval = L.dev(hval, name="hval")
    order = torch.empty_like(cost)
    hs = torch.empty_like(hval)
    hmean = torch.empty(1, dtype=torch.float64, device=cost.device)
    ws = torch.empty(max(1, (L.lib().ff_walker_order_workspace_bytes(L.i64(cost.numel())) + 7) // 8), dtype=torch.float64, device=cost.device)
    pc, ph, pe = prev if prev is not None else (None, None, None)
    if prev is not None and not (pc.numel() == ph.numel() == pe.numel() == cost.numel()):
        raise ValueError("walker_schedule: the previous pass must have this call's batch size")
    if counts is not None and not (counts.numel() == SCALE_COUNTS and counts.dtype == torch.float64 and counts.is_contiguous()
                                   and counts.is_cuda and counts.device == cost.device):
        raise ValueError(f"walker_schedule: counts must be the {SCALE_COUNTS} doubles of scale_counts, contiguous, on the walkers' device")
    L.check(L.lib().ff_walker_schedule(L.stream(), L.i64(cost.numel()), L.ptr(cost), L.ptr(order), L.ptr(ws), L.ptr(hval), L.ptr(hmean),
                                       L.ptr(scale_in), L.ptr(scale_out), L.ptr(pc), L.ptr(ph), L.ptr(pe), L.ptr(counts), L.f64(abs(float(interval))), L.ptr(hs),
                                       L.f64(shrink_at)),
            "ff_walker_schedule")
    return order, hmean, hs


SCALE_COUNTS = 128      # doubles of ff_scale_counts / ff_walker_schedule's prev_counts (include/fermiflow.h)


def scale_counts(cost, hs, he, interval=0.0):
    """ff_scale_counts: 128 doubles [walkers by cost class | of them, first step rejected | planned for >= 3 equal steps of `interval` | of
    those, accepted a step of the plan one shorter] of a local-energy pass (this rank's shard)"""
    counts = torch.zeros(SCALE_COUNTS, dtype=torch.float64, device=cost.device)
    L.check(L.lib().ff_scale_counts(L.stream(), L.i64(cost.numel()), L.ptr(cost), L.ptr(hs), L.ptr(he), L.f64(abs(float(interval))), L.ptr(counts)),
            "ff_scale_counts")
    return counts


def reduce_moments(e, shift=0.0, shift_dev=None, shift_dev_scale=1.0, out=None):
    """tensor [sum(e - shift), sum((e - shift)^2)] on the device; shift_dev: optional 1-element device tensor, then
    shift = shift_dev[0] * shift_dev_scale (no host round trip for the mean); out: where to write the two sums."""
    e = L.dev(e, name="e")
    out = torch.empty(2, dtype=torch.float64, device=e.device) if out is None else out
    if shift_dev is not None:
        shift_dev = L.dev(shift_dev.reshape(1), name="shift_dev")
    L.check(L.lib().ff_reduce_moments(L.stream(), L.i64(e.numel()), L.ptr(e), L.f64(shift), L.ptr(shift_dev),
                                      L.f64(shift_dev_scale), L.ptr(out)), "ff_reduce_moments")
    return out


def adam_step(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step):
    """ff_adam_step: one Adam update of the given fp64 device tensors (lists of equal length) in one launch."""
    n = len(params)
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    sizes = (C.c_int64 * n)(*[p.numel() for p in params])
    L.check(L.lib().ff_adam_step(L.stream(), n, sizes, arr(params), arr(grads), arr(exp_avg), arr(exp_avg_sq), L.f64(lr), L.f64(beta1), L.f64(beta2),
                                 L.f64(eps), L.f64(weight_decay), L.i64(step)), "ff_adam_step")


def stream_delay(microseconds):
    """ff_stream_delay on the current stream."""
    L.check(L.lib().ff_stream_delay(L.stream(), L.f64(microseconds)), "ff_stream_delay")


def reduce_energy(e, logp, shift_dev):
    """ff_reduce_energy: tensor [sum(e - c), sum((e - c)^2), sum(logp), sum(logp (e - c))], c = shift_dev[0] (device)."""
    e = L.dev(e, name="e"); logp = L.dev(logp, name="logp"); shift_dev = L.dev(shift_dev.reshape(1), name="shift_dev")
    out = torch.empty(4, dtype=torch.float64, device=e.device)
    L.check(L.lib().ff_reduce_energy(L.stream(), L.i64(e.numel()), L.ptr(e), L.ptr(logp), L.ptr(shift_dev), L.ptr(out)), "ff_reduce_energy")
    return out


_EST_WS = {}      # (device, stream, B) -> zero-initialised workspace of ff_energy_estimate (its counter is left at zero by every call)


def energy_estimate(e, logp, shift_dev, n_global):
    """ff_energy_estimate: (sums4, est3) in one launch; n_global = 0: sums4 only (est3 is None) -- all-reduce it, then energy_finish."""
    e = L.dev(e, name="e"); logp = L.dev(logp, name="logp"); shift_dev = L.dev(shift_dev.reshape(1), name="shift_dev")
    B = e.numel()
    # the workspace (last-workgroup counter + per-segment partials) is the caller's per call STREAM (include/fermiflow.h): two sweeps on
    # different streams of one device must not share it
    key = (str(e.device), int(torch.cuda.current_stream(e.device).cuda_stream), B)
    ws = _EST_WS.get(key)
    if ws is None:
        if len(_EST_WS) > 64:
            _EST_WS.clear()
        ws = _EST_WS[key] = torch.zeros(L.lib().ff_energy_estimate_workspace_bytes(L.i64(B)) // 8, dtype=torch.float64, device=e.device)
    sums = torch.empty(4, dtype=torch.float64, device=e.device)
    est = torch.empty(3, dtype=torch.float64, device=e.device) if n_global else None
    try:
        L.check(L.lib().ff_energy_estimate(L.stream(), L.i64(B), L.ptr(e), L.ptr(logp), L.ptr(shift_dev), L.i64(n_global), L.ptr(sums), L.ptr(est),
                                           L.ptr(ws)), "ff_energy_estimate")
    except Exception:
        _EST_WS.pop(key, None)      # a launch that did not run to its end may have left the counter non-zero: never reuse this workspace
        raise
    return sums, est


def energy_finish(sums4, shift_dev, n_global):
    """ff_energy_finish: tensor [E, sum((e - E)^2), mean(logp (e - E))] from the (all-reduced) sums of reduce_energy."""
    sums4 = L.dev(sums4, name="sums4"); shift_dev = L.dev(shift_dev.reshape(1), name="shift_dev")
    out = torch.empty(3, dtype=torch.float64, device=sums4.device)
    L.check(L.lib().ff_energy_finish(L.stream(), L.ptr(sums4), L.ptr(shift_dev), L.i64(n_global), L.ptr(out)), "ff_energy_finish")
    return out


def beta_buffer(nstates, device):
    """buffer of ff_beta_state_partials / ff_beta_finish: [0:2] moments of E_loc, [2:] partial per-state sums."""
    return torch.empty(L.lib().ff_beta_buffer_doubles(int(nstates)), dtype=torch.float64, device=device)


def beta_state_partials(e, logp, walker_state, nstates, buf):
    e = L.dev(e, name="e"); logp = L.dev(logp, name="logp"); ws = L.dev(walker_state, torch.int32, "walker_state")
    L.check(L.lib().ff_beta_state_partials(L.stream(), L.i64(e.numel()), int(nstates), L.ptr(ws), L.ptr(e), L.ptr(logp), L.ptr(buf)),
            "ff_beta_state_partials")


def beta_finish(buf, shift_dev, logits, beta, n_global):
    """ff_beta_finish: (est8, gphi, mean_e, logp_all) from the all-reduced buffer."""
    logits = L.dev(logits, name="logits"); shift_dev = L.dev(shift_dev.reshape(1), name="shift_dev")
    ns = logits.numel()
    f = dict(dtype=torch.float64, device=buf.device)
    est, gphi, mean_e, lpa = torch.empty(8, **f), torch.empty(ns, **f), torch.empty(ns, **f), torch.empty(ns, **f)
    L.check(L.lib().ff_beta_finish(L.stream(), L.ptr(buf), L.ptr(shift_dev), L.ptr(logits), ns, L.f64(beta), L.i64(n_global),
                                   L.ptr(est), L.ptr(gphi), L.ptr(mean_e), L.ptr(lpa)), "ff_beta_finish")
    return est, gphi, mean_e, lpa


# ---- d = 3 (csrc/ff_ho3d.hip) ---------------------------------------------------------------------------
def logprob3d(tab_up, tab_dn, nup, ndn, x, walker_state=None, derivs=False):
    x = L.dev(x, name="x")
    B = x.shape[0]
    logp = torch.empty(B, dtype=torch.float64, device=x.device)
    grad = torch.empty_like(x) if derivs else None
    lap = torch.empty(B, dtype=torch.float64, device=x.device) if derivs else None
    L.check(L.lib().ff_logprob3d(L.stream(), L.i64(B), nup, ndn, L.ptr(tab_up), L.ptr(tab_dn), L.ptr(_state(walker_state)),
                                 L.ptr(x), L.ptr(logp), L.ptr(grad), L.ptr(lap)), "ff_logprob3d")
    return (logp, grad, lap) if derivs else logp


def mcmc_sample_noise3d(tab_up, tab_dn, nup, ndn, g0, g, u, tau=0.1, walker_state=None):
    g0, g, u = L.dev(g0, name="g0"), L.dev(g, name="g"), L.dev(u, name="u")
    B, steps = g0.shape[0], g.shape[0]
    x = torch.empty_like(g0)
    logp = torch.empty(B, dtype=torch.float64, device=g0.device)
    acc = torch.empty(steps, B, dtype=torch.uint8, device=g0.device)
    L.check(L.lib().ff_mcmc_sample_noise3d(L.stream(), L.i64(B), nup, ndn, L.ptr(tab_up), L.ptr(tab_dn), L.ptr(_state(walker_state)),
                                           int(steps), L.f64(tau), L.ptr(g0), L.ptr(g), L.ptr(u), L.ptr(x), L.ptr(logp), L.ptr(acc)),
            "ff_mcmc_sample_noise3d")
    return x, logp, acc


def mcmc_sample3d(tab_up, tab_dn, nup, ndn, B, steps, tau, seed, device, walker_offset=0, walker_state=None):
    n = nup + ndn
    x = torch.empty(B, n, 3, dtype=torch.float64, device=device)
    logp = torch.empty(B, dtype=torch.float64, device=device)
    cnt = torch.empty(B, dtype=torch.int32, device=device)
    L.check(L.lib().ff_mcmc_sample3d(L.stream(), L.i64(B), nup, ndn, L.ptr(tab_up), L.ptr(tab_dn), L.ptr(_state(walker_state)),
                                     int(steps), L.f64(tau), C.c_uint64(int(seed) & (2**64 - 1)), L.i64(walker_offset),
                                     L.ptr(x), L.ptr(logp), L.ptr(cnt)), "ff_mcmc_sample3d")
    return x, logp, cnt


def backflow_v_div_f32(net, x):
    """Backflow v and div v with the arithmetic in fp32 (fp64 tensors at the boundary)."""
    x = L.dev(x, name="x")
    B, n, d = x.shape
    v = torch.empty_like(x)
    div = torch.empty(B, dtype=torch.float64, device=x.device)
    L.check(L.lib().ff_backflow_v_div_f32(L.stream(), L.i64(B), n, d, net.ref(), L.ptr(x), L.ptr(v), L.ptr(div)), "ff_backflow_v_div_f32")
    return v, div
