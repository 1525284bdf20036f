"""TEST-ONLY: numpy/ctypes access to tests/hostsim/libff_hostsim.so -- the kernel sources of
fermiflow_amd/csrc compiled for the host (see hip_shim.h).  "Device" pointers are numpy buffers.
Used by tests/test_hostsim.py to exercise the kernels' logic in the GPU-less build container.
The product package never imports this."""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class FFNet(C.Structure):
    _fields_ = [("He", C.c_int32), ("ew1", C.c_void_p), ("eb1", C.c_void_p), ("ew2", C.c_void_p),
                ("Hm", C.c_int32), ("mw1", C.c_void_p), ("mb1", C.c_void_p), ("mw2", C.c_void_p),
                ("radial_table", C.c_void_p)]


class FFOde(C.Structure):
    _fields_ = [("t0", C.c_double), ("t1", C.c_double), ("rtol", C.c_double), ("atol", C.c_double),
                ("max_steps", C.c_int32), ("walker_cost", C.c_void_p), ("walker_order", C.c_void_p),
                ("walker_h_init", C.c_void_p), ("walker_h_scale", C.c_double), ("walker_h_out", C.c_void_p),
                ("walker_class", C.c_void_p), ("sens_tol", C.c_double), ("walker_h_scale_loose", C.c_double), ("sens_tol_class", C.c_int32),
                ("walker_h_uniform", C.c_int32), ("heavy_class", C.c_int32), ("heavy_tol", C.c_double), ("sum_weight", C.c_double),
                ("compact_finish", C.c_int32), ("after_main_event", C.c_void_p), ("walker_h_equal", C.c_int32)]


def build():
    subprocess.check_call(["make", "-s", "-j4", "-C", HERE])


def lib():
    global _LIB
    if _LIB is None:
        # Which local-energy kernel "auto" picks is a measured-on-MI355X choice; under the simulator the matrix-core kernel is the
        # slowest by far (every ds_bpermute / DPP exchange is a barrier of 64 host threads), so the simulator's default stays the
        # column sweep (tests/hostsim/Makefile: -DFF_MFMA_FROM=99) and the tests that are ABOUT the matrix-core kernel select it
        # (FF_ELOC_KERNEL=mfma in a subprocess).
        alt = os.environ.get("FF_HOSTSIM_LIB")     # e.g. a build of the same sources under the address / UB sanitizers
        if not alt:
            build()
        _LIB = C.CDLL(alt or os.path.join(HERE, "libff_hostsim.so"))
        _LIB.ff_last_error.restype = C.c_char_p
        _LIB.ff_eloc_workspace_bytes.restype = C.c_size_t
        _LIB.ff_cnf_adjoint_workspace_bytes.restype = C.c_size_t
    return _LIB


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _ck(st):
    if st != 0:
        raise RuntimeError(f"ff status {st}: {lib().ff_last_error().decode()}")


class Net:
    def __init__(self, eta, mu=None, table=False):
        self.e = [_d(np.reshape(a, -1)) for a in eta]
        self.m = [_d(np.reshape(a, -1)) for a in mu] if mu is not None else None
        self.c = FFNet(len(self.e[0]), _p(self.e[0]), _p(self.e[1]), _p(self.e[2]),
                       len(self.m[0]) if self.m else 0,
                       _p(self.m[0]) if self.m else None, _p(self.m[1]) if self.m else None,
                       _p(self.m[2]) if self.m else None, None)
        if table:
            lib().ff_radial_table_bytes.restype = C.c_size_t
            self.tab = np.zeros(lib().ff_radial_table_bytes() // 8)
            _ck(lib().ff_radial_table_build(None, C.byref(self.c), _p(self.tab)))
            self.c.radial_table = self.tab.ctypes.data

    @property
    def nparams(self):
        return 3 * self.c.He + 3 * self.c.Hm


def _tabs(nup, ndn, tab_up, tab_dn):
    tu = _i(np.arange(nup) if tab_up is None else tab_up) if nup else None
    td = _i(np.arange(ndn) if tab_dn is None else tab_dn) if ndn else None
    return tu, td


def logprob(x, nup, ndn, tab_up=None, tab_dn=None, wstate=None):
    x = _d(x); B = x.shape[0]
    tu, td = _tabs(nup, ndn, tab_up, tab_dn); ws = _i(wstate) if wstate is not None else None
    lp = np.empty(B); g = np.empty_like(x); l = np.empty(B)
    _ck(lib().ff_logprob(None, C.c_int64(B), nup, ndn, _p(tu), _p(td), _p(ws), _p(x), _p(lp), _p(g), _p(l)))
    return lp, g, l


def slater(x, orb, wstate=None, gout=None):
    x = _d(x); B, n, _ = x.shape
    t = _i(orb); ws = _i(wstate) if wstate is not None else None
    lad = np.empty(B)
    _ck(lib().ff_slater_logabsdet_fwd(None, C.c_int64(B), n, _p(t), _p(ws), _p(x), _p(lad)))
    gx = None
    if gout is not None:
        gx = np.empty_like(x); go = _d(gout)
        _ck(lib().ff_slater_logabsdet_bwd(None, C.c_int64(B), n, _p(t), _p(ws), _p(x), _p(go), _p(gx)))
    return lad, gx


def mcmc_noise(g0, g, u, nup, ndn, tau=0.1, tab_up=None, tab_dn=None, wstate=None):
    g0, g, u = _d(g0), _d(g), _d(u)
    B = g0.shape[0]; steps = g.shape[0]
    tu, td = _tabs(nup, ndn, tab_up, tab_dn); ws = _i(wstate) if wstate is not None else None
    x = np.empty_like(g0); lp = np.empty(B); acc = np.empty((steps, B), dtype=np.uint8)
    _ck(lib().ff_mcmc_sample_noise(None, C.c_int64(B), nup, ndn, _p(tu), _p(td), _p(ws), steps, C.c_double(tau),
                                   _p(g0), _p(g), _p(u), _p(x), _p(lp), _p(acc)))
    return x, lp, acc


def mcmc(B, nup, ndn, steps, seed, offset=0, tau=0.1):
    n = nup + ndn
    tu, td = _tabs(nup, ndn, None, None)
    x = np.empty((B, n, 2)); lp = np.empty(B); cnt = np.empty(B, dtype=np.int32)
    _ck(lib().ff_mcmc_sample(None, C.c_int64(B), nup, ndn, _p(tu), _p(td), None, steps, C.c_double(tau),
                             C.c_uint64(seed), C.c_int64(offset), _p(x), _p(lp), _p(cnt)))
    return x, lp, cnt


def mcmc_continue(x_init, nup, ndn, steps, seed, offset=0, tau=0.1):
    x0 = _d(x_init); B = x0.shape[0]
    tu, td = _tabs(nup, ndn, None, None)
    x = np.empty_like(x0); lp = np.empty(B); cnt = np.empty(B, dtype=np.int32)
    _ck(lib().ff_mcmc_continue(None, C.c_int64(B), nup, ndn, _p(tu), _p(td), None, steps, C.c_double(tau),
                               C.c_uint64(seed), C.c_int64(offset), _p(x0), _p(x), _p(lp), _p(cnt)))
    return x, lp, cnt


def rng_fill(B, n, steps, seed, offset=0, dim=2):
    g0 = np.empty((B, n, dim)); g = np.empty((steps, B, n, dim)); u = np.empty((steps, B))
    fn = lib().ff_rng_fill if dim == 2 else lib().ff_rng_fill3d
    _ck(fn(None, C.c_int64(B), n, steps, C.c_uint64(seed), C.c_int64(offset), _p(g0), _p(g), _p(u)))
    return g0, g, u


def backflow(x, net):
    x = _d(x); B, n, d = x.shape
    v = np.empty_like(x); div = np.empty(B)
    _ck(lib().ff_backflow_v_div(None, C.c_int64(B), n, d, C.byref(net.c), _p(x), _p(v), _p(div)))
    return v, div


def potential(x, Z, use_ho=True):
    x = _d(x); B, n, d = x.shape
    V = np.empty(B)
    _ck(lib().ff_potential(None, C.c_int64(B), n, d, C.c_double(Z), int(use_ho), _p(x), _p(V)))
    return V


def mlp(r, w1, b1, w2):
    r = _d(r).reshape(-1); w1, b1, w2 = _d(w1).reshape(-1), _d(b1), _d(w2).reshape(-1)
    v = np.empty_like(r); dv = np.empty_like(r)
    _ck(lib().ff_mlp_eval(None, C.c_int64(len(r)), len(b1), _p(w1), _p(b1), _p(w2), _p(r), _p(v), _p(dv)))
    return v, dv


_WARM = {}     # set by warm(h_init=..., h_scale=..., h_out=...) for the next calls (keeps the wrappers' signatures short)


def warm(h_init=None, h_scale=1.0, h_out=None, uniform=False, max_steps=0, wclass=None, sens_tol=1.0, sens_class=0, h_scale_loose=0.0,
         heavy_class=0, heavy_tol=0.0, sum_weight=0.0, h_equal=False):
    _WARM.clear()
    _WARM.update(h_init=h_init, h_scale=h_scale, h_out=h_out, uniform=uniform, max_steps=max_steps, wclass=wclass, sens_tol=sens_tol, sens_class=sens_class,
                 h_scale_loose=h_scale_loose, heavy_class=heavy_class, heavy_tol=heavy_tol, sum_weight=sum_weight, h_equal=h_equal)


def _ode(t0, t1, rtol, atol, steps=None, order=None, compact=False):
    q = lambda a: a.ctypes.data if a is not None else None
    qi = q
    return FFOde(t0, t1, rtol, atol, int(_WARM.get("max_steps", 0)), q(steps), q(order), q(_WARM.get("h_init")), float(_WARM.get("h_scale", 1.0)),
                 q(_WARM.get("h_out")), qi(_WARM.get("wclass")), float(_WARM.get("sens_tol", 1.0)), float(_WARM.get("h_scale_loose", 0.0)),
                 int(_WARM.get("sens_class", 0)), int(bool(_WARM.get("uniform", False))), int(_WARM.get("heavy_class", 0)),
                 float(_WARM.get("heavy_tol", 0.0)), float(_WARM.get("sum_weight", 0.0)), int(bool(compact)), None, int(bool(_WARM.get("h_equal", False))))


def walker_order(cost, hval=None):
    cost = _i(cost); order = np.empty_like(cost)
    lib().ff_walker_order_workspace_bytes.restype = C.c_size_t
    ws = np.zeros((lib().ff_walker_order_workspace_bytes(C.c_int64(len(cost))) + 7) // 8)
    if hval is None:
        _ck(lib().ff_walker_order(None, C.c_int64(len(cost)), _p(cost), _p(order), _p(ws)))
        return order
    hval = _d(hval); hm = np.empty(1)
    _ck(lib().ff_walker_order_mean(None, C.c_int64(len(cost)), _p(cost), _p(order), _p(ws), _p(hval), _p(hm)))
    return order, hm[0]


def walker_schedule(cost, hval, scale_in, prev=None, interval=0.0, counts=None, shrink_at=0.0):
    """ff_walker_schedule -> (order, mean(hval), hs, updated scale table)"""
    cost = _i(cost); hval = _d(hval); scale_in = _d(scale_in)
    order = np.empty_like(cost); hm = np.empty(1); hs = np.empty_like(hval); scale_out = np.full(32, np.nan)
    lib().ff_walker_order_workspace_bytes.restype = C.c_size_t
    ws = np.zeros((lib().ff_walker_order_workspace_bytes(C.c_int64(len(cost))) + 7) // 8)
    pc, ph, pe = (None, None, None) if prev is None else (_i(prev[0]), _d(prev[1]), _d(prev[2]))
    _ck(lib().ff_walker_schedule(None, C.c_int64(len(cost)), _p(cost), _p(order), _p(ws), _p(hval), _p(hm), _p(scale_in), _p(scale_out),
                                 _p(pc), _p(ph), _p(pe), _p(_d(counts)) if counts is not None else None, C.c_double(interval), _p(hs), C.c_double(shrink_at)))
    return order, hm[0], hs, scale_out


def adam_step(params, grads, m, v, lr, beta1, beta2, eps, wd, step):
    """ff_adam_step on lists of float64 arrays (updated in place)"""
    n = len(params)
    arr = lambda xs: (C.c_void_p * n)(*[x.ctypes.data for x in xs])
    sizes = (C.c_int64 * n)(*[x.size for x in params])
    _ck(lib().ff_adam_step(None, n, sizes, arr(params), arr(grads), arr(m), arr(v), C.c_double(lr), C.c_double(beta1), C.c_double(beta2),
                           C.c_double(eps), C.c_double(wd), C.c_int64(step)))


def scale_counts(cost, hs, he, interval=0.0):
    cost = _i(cost); counts = np.zeros(128)
    _ck(lib().ff_scale_counts(None, C.c_int64(len(cost)), _p(cost), _p(_d(hs)), _p(_d(he)), C.c_double(interval), _p(counts)))
    return counts


def cnf_generate(z, net, t0=0.0, t1=1.0, rtol=1e-6, atol=1e-8, steps=None, order=None):
    z = _d(z); B, n, d = z.shape
    x = np.empty_like(z); stats = np.zeros(4, dtype=np.int32); ode = _ode(t0, t1, rtol, atol, steps, order)
    _ck(lib().ff_cnf_generate(None, C.c_int64(B), n, d, C.byref(net.c), C.byref(ode), _p(z), _p(x), _p(stats)))
    return x, stats


def cnf_delta_logp(x, net, t0=0.0, t1=1.0, rtol=1e-6, atol=1e-8, steps=None, order=None):
    x = _d(x); B, n, d = x.shape
    z = np.empty_like(x); dl = np.empty(B); stats = np.zeros(4, dtype=np.int32); ode = _ode(t0, t1, rtol, atol, steps, order)
    _ck(lib().ff_cnf_delta_logp(None, C.c_int64(B), n, d, C.byref(net.c), C.byref(ode), _p(x), _p(z), _p(dl), _p(stats)))
    return z, dl, stats


def cnf_adjoint(z0, a_z, a_d, net, t0=0.0, t1=1.0, rtol=1e-6, atol=1e-8, steps=None, order=None):
    z0, a_z, a_d = _d(z0), _d(a_z), _d(a_d); B, n, d = z0.shape
    gx = np.empty_like(z0); gp = np.empty(net.nparams); stats = np.zeros(4, dtype=np.int32); ode = _ode(t0, t1, rtol, atol, steps, order)
    ws = np.zeros(max(1, lib().ff_cnf_adjoint_workspace_bytes(C.c_int64(B), n, d, net.c.He, net.c.Hm) // 8))
    _ck(lib().ff_cnf_adjoint(None, C.c_int64(B), n, d, C.byref(net.c), C.byref(ode), _p(z0), _p(a_z), _p(a_d),
                             _p(gx), _p(gp), _p(ws), _p(stats)))
    return gx, gp, stats


def eloc(x, nup, ndn, net, Z, use_ho=True, t0=0.0, t1=1.0, rtol=1e-6, atol=1e-8, tab_up=None, tab_dn=None, wstate=None,
         steps=None, order=None):
    x = _d(x); B = x.shape[0]; n = nup + ndn
    tu, td = _tabs(nup, ndn, tab_up, tab_dn); ws = _i(wstate) if wstate is not None else None
    o = dict(logp=np.empty(B), grad=np.empty_like(x), lap=np.empty(B), V=np.empty(B), eloc=np.empty(B),
             z=np.empty_like(x), dlogp=np.empty(B), glogp0=np.empty_like(x))
    stats = np.zeros(4, dtype=np.int32); ode = _ode(t0, t1, rtol, atol, steps, order)
    wk = np.zeros(lib().ff_eloc_workspace_bytes(C.c_int64(B), n, 2) // 8)
    _ck(lib().ff_eloc(None, C.c_int64(B), nup, ndn, _p(tu), _p(td), _p(ws), C.byref(net.c), C.byref(ode), C.c_double(Z),
                      int(use_ho), _p(x), _p(o["logp"]), _p(o["grad"]), _p(o["lap"]), _p(o["V"]), _p(o["eloc"]),
                      _p(o["z"]), _p(o["dlogp"]), _p(o["glogp0"]), _p(wk), _p(stats)))
    o["stats"] = stats
    return o


def eloc_nd(x, nup, ndn, net, Z, use_ho=True, t0=0.0, t1=1.0, rtol=1e-6, atol=1e-8, tab_up=None, tab_dn=None, wstate=None, compact=True):
    """ff_eloc_nd (d from x) with ff_ode.compact_finish; z / dlogp are read from the head of the workspace."""
    x = _d(x); B, n, d = x.shape
    tu, td = _tabs(nup, ndn, tab_up, tab_dn); ws = _i(wstate) if wstate is not None else None
    o = dict(logp=np.empty(B), grad=np.empty_like(x), lap=np.empty(B), V=np.empty(B), eloc=np.empty(B), glogp0=np.empty_like(x))
    stats = np.zeros(4, dtype=np.int32); ode = _ode(t0, t1, rtol, atol, compact=compact)
    lib().ff_eloc_nd_workspace_bytes.restype = C.c_size_t
    nb = lib().ff_eloc_nd_workspace_bytes(C.c_int64(B), n, d, int(bool(compact)))
    wk = np.zeros(nb // 8)
    _ck(lib().ff_eloc_nd(None, C.c_int64(B), nup, ndn, d, _p(tu), _p(td), _p(ws), C.byref(net.c), C.byref(ode), C.c_double(Z),
                         int(use_ho), _p(x), _p(o["logp"]), _p(o["grad"]), _p(o["lap"]), _p(o["V"]), _p(o["eloc"]),
                         None, None, _p(o["glogp0"]), _p(wk), _p(stats)))
    M = n * d
    o["z"] = wk[:B * M].reshape(B, n, d).copy()
    dl0 = B * M if (compact and M > 24) else B * (M * M + 4 * M)
    o["dlogp"] = wk[dl0:dl0 + B].copy(); o["stats"] = stats; o["workspace_bytes"] = nb
    return o


def reduce_energy(e, logp, shift):
    e = _d(e); logp = _d(logp); sh = np.array([shift], dtype=np.float64); out = np.empty(4)
    _ck(lib().ff_reduce_energy(None, C.c_int64(len(e)), _p(e), _p(logp), _p(sh), _p(out)))
    return out


def energy_estimate(e, logp, shift, n_global, ws=None):
    """ff_energy_estimate: (sums4, est3 or None, workspace) -- pass the returned workspace to the next call (zeroed once)."""
    e = _d(e); logp = _d(logp); sh = np.array([shift], dtype=np.float64)
    lib().ff_energy_estimate_workspace_bytes.restype = C.c_size_t
    if ws is None:
        ws = np.zeros(lib().ff_energy_estimate_workspace_bytes(C.c_int64(len(e))) // 8)
    sums = np.empty(4); est = np.empty(3) if n_global else None
    _ck(lib().ff_energy_estimate(None, C.c_int64(len(e)), _p(e), _p(logp), _p(sh), C.c_int64(n_global), _p(sums), _p(est), _p(ws)))
    return sums, est, ws


def energy_finish(sums4, shift, n):
    sums4 = _d(sums4); sh = np.array([shift], dtype=np.float64); out = np.empty(3)
    _ck(lib().ff_energy_finish(None, _p(sums4), _p(sh), C.c_int64(n), _p(out)))
    return out


def cnf_adjoint_energy(z0, glogp0, eloc, e_mean, scale, net, t0=0.0, t1=1.0, rtol=1e-6, atol=1e-8, mean_index=None):
    z0 = _d(z0); glogp0 = _d(glogp0); eloc = _d(eloc); B, n, d = z0.shape
    em = np.atleast_1d(np.asarray(e_mean, dtype=np.float64)).copy()
    mi = _i(mean_index) if mean_index is not None else None
    gx = np.empty_like(z0); gp = np.empty(3 * net.c.He + 3 * net.c.Hm); stats = np.zeros(4, dtype=np.int32); ode = _ode(t0, t1, rtol, atol)
    lib().ff_cnf_adjoint_workspace_bytes.restype = C.c_size_t
    ws = np.zeros(max(1, lib().ff_cnf_adjoint_workspace_bytes(C.c_int64(B), n, d, net.c.He, net.c.Hm) // 8))
    _ck(lib().ff_cnf_adjoint_energy(None, C.c_int64(B), n, d, C.byref(net.c), C.byref(ode), _p(z0), _p(glogp0), _p(eloc), _p(em),
                                    _p(mi), C.c_double(scale), _p(gx), _p(gp), _p(ws), _p(stats)))
    return gx, gp, stats


def beta_estimator(e, logp, ws, logits, beta, shift, n_global=None):
    """ff_reduce_moments + ff_beta_state_partials + ff_beta_finish on one rank: (est8, gphi, mean_e, logp_all)."""
    e = _d(e); logp = _d(logp); ws = _i(ws); logits = _d(logits); ns = len(logits)
    lib().ff_beta_buffer_doubles.restype = C.c_size_t
    buf = np.zeros(lib().ff_beta_buffer_doubles(ns))
    sh = np.array([shift], dtype=np.float64)
    mom = np.empty(2)
    _ck(lib().ff_reduce_moments(None, C.c_int64(len(e)), _p(e), C.c_double(0.0), _p(sh), C.c_double(1.0), _p(mom)))
    _ck(lib().ff_beta_state_partials(None, C.c_int64(len(e)), ns, _p(ws), _p(e), _p(logp), _p(buf)))
    buf[:2] = mom
    est, gphi, mean_e, lpa = np.empty(8), np.empty(ns), np.empty(ns), np.empty(ns)
    _ck(lib().ff_beta_finish(None, _p(buf), _p(sh), _p(logits), ns, C.c_double(beta), C.c_int64(n_global or len(e)), _p(est), _p(gphi),
                             _p(mean_e), _p(lpa)))
    return est, gphi, mean_e, lpa


def moments(e, shift=0.0):
    e = _d(e); out = np.empty(2)
    _ck(lib().ff_reduce_moments(None, C.c_int64(len(e)), _p(e), C.c_double(shift), None, C.c_double(1.0), _p(out)))
    return out


def state_sums(e, ws, nstates):
    e = _d(e); ws = _i(ws); sums = np.empty(nstates); cnt = np.empty(nstates)
    _ck(lib().ff_state_sums(None, C.c_int64(len(e)), int(nstates), _p(ws), _p(e), _p(sums), _p(cnt)))
    return sums, cnt


# ---- d = 3 groundwork (csrc/ff_ho3d.hip)
def logprob3d(x, nup, ndn, tab_up=None, tab_dn=None, wstate=None):
    x = _d(x); B = x.shape[0]
    tu, td = _tabs(nup, ndn, tab_up, tab_dn); ws = _i(wstate) if wstate is not None else None
    lp = np.empty(B); g = np.empty_like(x); lap = np.empty(B)
    _ck(lib().ff_logprob3d(None, C.c_int64(B), nup, ndn, _p(tu), _p(td), _p(ws), _p(x), _p(lp), _p(g), _p(lap)))
    return lp, g, lap


def mcmc_noise3d(g0, g, u, nup, ndn, tau=0.1, tab_up=None, tab_dn=None, wstate=None):
    g0, g, u = _d(g0), _d(g), _d(u); B, steps = g0.shape[0], g.shape[0]
    tu, td = _tabs(nup, ndn, tab_up, tab_dn); ws = _i(wstate) if wstate is not None else None
    x = np.empty_like(g0); lp = np.empty(B); acc = np.empty((steps, B), dtype=np.uint8)
    _ck(lib().ff_mcmc_sample_noise3d(None, C.c_int64(B), nup, ndn, _p(tu), _p(td), _p(ws), steps, C.c_double(tau), _p(g0), _p(g), _p(u),
                                     _p(x), _p(lp), acc.ctypes.data_as(C.c_void_p)))
    return x, lp, acc


def mcmc3d(B, nup, ndn, steps, seed, offset=0, tau=0.1, tab_up=None, tab_dn=None):
    tu, td = _tabs(nup, ndn, tab_up, tab_dn)
    x = np.empty((B, nup + ndn, 3)); lp = np.empty(B); cnt = np.empty(B, dtype=np.int32)
    _ck(lib().ff_mcmc_sample3d(None, C.c_int64(B), nup, ndn, _p(tu), _p(td), None, steps, C.c_double(tau), C.c_uint64(seed),
                               C.c_int64(offset), _p(x), _p(lp), _p(cnt)))
    return x, lp, cnt


def backflow_f32(x, net):
    x = _d(x); B, n, d = x.shape
    v = np.empty_like(x); div = np.empty(B)
    _ck(lib().ff_backflow_v_div_f32(None, C.c_int64(B), n, d, C.byref(net.c), _p(x), _p(v), _p(div)))
    return v, div


def eloc3d(x, nup, ndn, net, Z, use_ho=True, t0=0.0, t1=1.0, rtol=1e-6, atol=1e-8, tab_up=None, tab_dn=None, wstate=None):
    """ff_eloc_sensitivities (d = 3) + ff_eloc_finish3d."""
    x = _d(x); B = x.shape[0]; n = nup + ndn
    tu, td = _tabs(nup, ndn, tab_up, tab_dn); ws = _i(wstate) if wstate is not None else None
    o = dict(logp=np.empty(B), grad=np.empty_like(x), lap=np.empty(B), V=np.empty(B), eloc=np.empty(B),
             z=np.empty_like(x), dlogp=np.empty(B), glogp0=np.empty_like(x))
    stats = np.zeros(4, dtype=np.int32); ode = _ode(t0, t1, rtol, atol)
    wk = np.zeros(lib().ff_eloc_workspace_bytes(C.c_int64(B), n, 3) // 8)
    _ck(lib().ff_eloc_sensitivities(None, C.c_int64(B), n, 3, C.byref(net.c), C.byref(ode), _p(x), _p(wk), _p(stats)))
    _ck(lib().ff_eloc_finish3d(None, C.c_int64(B), nup, ndn, _p(tu), _p(td), _p(ws), C.c_double(Z), int(use_ho), _p(x), _p(wk),
                               _p(o["logp"]), _p(o["grad"]), _p(o["lap"]), _p(o["V"]), _p(o["eloc"]), _p(o["z"]), _p(o["dlogp"]),
                               _p(o["glogp0"])))
    o["stats"] = stats
    return o
