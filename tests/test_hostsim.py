"""The kernel SOURCES of fermiflow_amd/csrc compiled for the host (tests/hostsim: one OS thread per lane, real
barriers) checked against the golden vectors and the oracle.  This exercises the kernels' logic -- lane
mapping, LDS hand-offs, step control, reductions -- in the GPU-less build container; the `-m gpu` tests in
test_gpu_parity.py are the parity tests proper (hipcc build, through the C ABI, on the MI355X)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from tests.common import mcmc_noise_from_seed, net_arrays, cnf_param_grads, gsvmc_param_grads
from tests.hostsim import simlib as S


def test_slater_kernels(golden):
    G = golden["g2_slater"]
    for n in (3, 6, 10):
        lad, gx = S.slater(G[f"n{n}_x"], G[f"n{n}_orb"], gout=np.ones(24))
        np.testing.assert_allclose(lad, G[f"n{n}_logabsdet"], atol=1e-12)
        np.testing.assert_allclose(gx, G[f"n{n}_grad"], rtol=1e-9, atol=1e-9)
    lp, g, lap = S.logprob(G["lp_x"], 3, 6, tab_up=G["lp_up"], tab_dn=G["lp_dn"])
    np.testing.assert_allclose(lp, G["lp_logp"], atol=1e-11)
    np.testing.assert_allclose(g, G["lp_grad"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(lap, G["lp_lap"], rtol=1e-9, atol=1e-6)
    ws = np.repeat(G["ms_keys"], G["ms_counts"])
    lp, g, lap = S.logprob(G["ms_x"], 3, 0, tab_up=G["ms_states"], wstate=ws)
    np.testing.assert_allclose(lp, G["ms_logp"], atol=1e-12)
    np.testing.assert_allclose(lap, G["ms_lap"], rtol=1e-9, atol=1e-8)


@pytest.mark.parametrize("name", ["u3d3", "u6d6", "u10d0", "u6d0", "u1d0"])
def test_mcmc_kernel_bit_exact(golden, name):
    G = golden["g1_mcmc"]
    nup, ndn, g0, g, u, accept = mcmc_noise_from_seed(G, name)
    x, logp, acc = S.mcmc_noise(g0, g, u, nup, ndn)
    assert (acc == accept).all() and (x == G[name + "_x"]).all()


def test_philox_mcmc_equals_noise_path():
    g0, g, u = S.rng_fill(40, 6, 20, 1234, offset=5)
    x1, _, acc1 = S.mcmc_noise(g0, g, u, 3, 3)
    x2, _, cnt = S.mcmc(40, 3, 3, 20, 1234, offset=5)
    assert (x1 == x2).all() and (acc1.sum(0) == cnt).all()
    # sharding invariance: walkers 5..44 drawn as two shards
    xa, _, _ = S.mcmc(15, 3, 3, 20, 1234, offset=5)
    xb, _, _ = S.mcmc(25, 3, 3, 20, 1234, offset=20)
    assert (np.concatenate([xa, xb]) == x2).all()
    # the stream itself (Box-Muller on 32-bit words in single precision, explicit sign bits; csrc/ff_rng.h): moments and a
    # Kolmogorov-Smirnov test of 123 000 normals, exact symmetry of the sign pattern aside (the GPU test repeats this on the
    # hardware transcendentals)
    stats = pytest.importorskip("scipy.stats")          # (the only use of scipy in the host suite)
    h0, h, hu = S.rng_fill(512, 6, 20, 77, offset=0)
    z = h.ravel()
    assert abs(z.mean()) < 0.012 and abs(z.std() - 1) < 0.01 and abs(hu.mean() - 0.5) < 0.012
    assert stats.kstest(z, "norm").pvalue > 1e-3 and stats.kstest(hu.ravel(), "uniform").pvalue > 1e-3
    assert np.isfinite(z).all() and np.abs(z).max() < 6.8


@pytest.mark.parametrize("nup", [2, 3, 4, 5, 6])
def test_particle_split_metropolis_kernel(nup):
    """ff_mcmc_pair_kernel (one spin species, two lanes per walker): the noise-fed chain equals the oracle's bit for bit --
    walkers, log-probabilities and accept masks, also with a different orbital set per walker -- and the Philox path equals
    the noise path on the materialised stream (ragged batch: the last workgroup has idle lanes)."""
    rng = np.random.default_rng(nup)
    B, steps = 37, 12
    g0 = rng.normal(size=(B, nup, 2)); g = rng.normal(size=(steps, B, nup, 2)); u = rng.random((steps, B))
    x, lp, acc = S.mcmc_noise(g0, g, u, nup, 0)
    xo, lpo, acco = O.mcmc_noise(g0, g, u, nup, 0)
    assert (x == xo).all() and (acc == acco).all() and np.allclose(lp, lpo, rtol=1e-13, atol=1e-13)
    tab = np.stack([np.sort(rng.choice(15, size=nup, replace=False)) for _ in range(4)]).astype(np.int32)      # four orbital sets
    ws = np.sort(rng.integers(0, 4, size=B)).astype(np.int32)
    x, lp, acc = S.mcmc_noise(g0, g, u, nup, 0, tab_up=tab, wstate=ws)
    xo, lpo, acco = O.mcmc_noise(g0, g, u, nup, 0, tab_up=tab, wstate=ws)
    assert (x == xo).all() and (acc == acco).all() and np.allclose(lp, lpo, rtol=1e-13, atol=1e-13)
    h0, h, hu = S.rng_fill(B, nup, steps, 99, offset=3)
    x1, _, acc1 = S.mcmc_noise(h0, h, hu, nup, 0)
    x2, _, cnt = S.mcmc(B, nup, 0, steps, 99, offset=3)
    assert (x1 == x2).all() and (acc1.sum(0) == cnt).all()


def test_backflow_kernel(golden):
    G = golden["g3_backflow"]
    for k in (1, 2, 3):
        n, d, He, Hm = G[f"c{k}_cfg"]
        eta, mu = net_arrays(G, f"c{k}_", Hm > 0)
        v, div = S.backflow(G[f"c{k}_x"], S.Net(eta, mu))
        np.testing.assert_allclose(v, G[f"c{k}_v"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(div, G[f"c{k}_div"], rtol=1e-12, atol=1e-12)


def test_cnf_kernels(golden):
    G = golden["g4_cnf"]
    net = S.Net(*net_arrays(G, ""))
    tag, rt, at = "tol10", 1e-10, 1e-12
    x, st = S.cnf_generate(G[tag + "_z"], net, rtol=rt, atol=at)
    np.testing.assert_allclose(x, G[tag + "_x"], atol=2e-10)
    z, dl, st = S.cnf_delta_logp(G[tag + "_x"], net, rtol=rt, atol=at)
    np.testing.assert_allclose(z, G[tag + "_zback"], atol=2e-10)
    np.testing.assert_allclose(dl, G[tag + "_dlogp"], atol=2e-10)
    gx, gp, st = S.cnf_adjoint(G[tag + "_zback"], G[tag + "_cz"], G[tag + "_cd"], net, rtol=rt, atol=at)
    np.testing.assert_allclose(gx, G[tag + "_gx"], atol=1e-9)
    ref = cnf_param_grads(G, tag)
    np.testing.assert_allclose(gp, ref, atol=1e-9 * np.abs(ref).max())
    assert st[3] == 0


@pytest.mark.parametrize("name", ["z2_zero", "z2_nt", "u6_nt", "z2_nomu"])
def test_eloc_kernel(golden, name):
    G = golden["g5_gsvmc"]
    nup, ndn, B, seed = (int(v) for v in G[name + "_cfg"])
    use_mu = bool(G[name + "_use_mu"])
    net = S.Net(*net_arrays(G, name + "_", use_mu))
    x = G[name + "_x"][:12]
    r = S.eloc(x, nup, ndn, net, float(G[name + "_Z"]))           # reference default tolerances
    np.testing.assert_allclose(r["eloc"], G[name + "_Eloc"][:12], rtol=1e-6)   # bar: 1e-5
    np.testing.assert_allclose(r["grad"], G[name + "_grad"][:12], atol=1e-5)
    onet = O.Net(*net_arrays(G, name + "_", use_mu))
    ref = O.eloc(x, nup, ndn, onet, float(G[name + "_Z"]), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(r["eloc"], ref["eloc"], rtol=1e-6)
    assert r["stats"][3] == 0


def test_radial_table_matches_direct_evaluation(golden):
    """ff_radial.h: the tabulated eta/mu heads reproduce the direct sigmoid evaluation to ~1e-13 in every
    fused kernel that uses them (generate, delta_logp, local energy)."""
    G = golden["g5_gsvmc"]
    name = "z2_nt"
    eta, mu = net_arrays(G, name + "_")
    exact, tab = S.Net(eta, mu), S.Net(eta, mu, table=True)
    assert tab.tab[3] == 0.0 and tab.tab[0] == 64.0          # grid 1/64 for these weights (max |w1| ~ 0.8)
    x = G[name + "_x"][:10]
    xe, _ = S.cnf_generate(x, exact, rtol=1e-9, atol=1e-11); xt, _ = S.cnf_generate(x, tab, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(xt, xe, atol=1e-12)
    ze, de, _ = S.cnf_delta_logp(x, exact, rtol=1e-9, atol=1e-11); zt, dt, _ = S.cnf_delta_logp(x, tab, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(zt, ze, atol=1e-12); np.testing.assert_allclose(dt, de, atol=1e-12)
    re = S.eloc(x, 3, 3, exact, 2.0, rtol=1e-9, atol=1e-11); rt = S.eloc(x, 3, 3, tab, 2.0, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(rt["eloc"], re["eloc"], rtol=1e-11)
    np.testing.assert_allclose(rt["lap"], re["lap"], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(rt["eloc"], G[name + "_Eloc"][:10], rtol=1e-8)
    # stiff weights switch to a finer grid; absurdly stiff ones disable the table (flag) and results stay exact
    stiff = (eta[0] * 10.0, eta[1], eta[2])
    t2 = S.Net(stiff, mu, table=True)
    assert t2.tab[3] == 0.0 and t2.tab[0] >= 128.0
    xe2, _ = S.cnf_generate(x, S.Net(stiff, mu), rtol=1e-9, atol=1e-11); xt2, _ = S.cnf_generate(x, t2, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(xt2, xe2, atol=1e-11)
    wild = (eta[0] * 1e4, eta[1], eta[2])
    t3 = S.Net(wild, mu, table=True)
    assert t3.tab[3] == 1.0


def test_tabulated_adjoint_matches_direct(golden):
    """ff_ode_adjtab_kernel + deposit/contract kernels vs the direct adjoint kernel and the reference gradients."""
    G = golden["g4_cnf"]
    eta, mu = net_arrays(G, "")
    exact, tab = S.Net(eta, mu), S.Net(eta, mu, table=True)
    assert tab.tab[4] == 0.0
    tag, rt, at = "tol10", 1e-10, 1e-12
    gx_e, gp_e, st_e = S.cnf_adjoint(G[tag + "_zback"], G[tag + "_cz"], G[tag + "_cd"], exact, rtol=rt, atol=at)
    gx_t, gp_t, st_t = S.cnf_adjoint(G[tag + "_zback"], G[tag + "_cz"], G[tag + "_cd"], tab, rtol=rt, atol=at)
    np.testing.assert_allclose(gx_t, gx_e, atol=1e-11)
    np.testing.assert_allclose(gp_t, gp_e, atol=1e-10 * np.abs(gp_e).max())
    ref = cnf_param_grads(G, tag)
    np.testing.assert_allclose(gp_t, ref, atol=1e-9 * np.abs(ref).max())
    np.testing.assert_allclose(gx_t, G[tag + "_gx"], atol=1e-9)
    # stiff weights: the deposit grid is refused on the device and the direct kernel serves the call
    stiff = (eta[0] * 20.0, eta[1], eta[2])
    t2 = S.Net(stiff, mu, table=True)
    assert t2.tab[4] == 1.0 and t2.tab[3] == 0.0
    _, gp_e2, _ = S.cnf_adjoint(G[tag + "_zback"][:5], G[tag + "_cz"][:5], G[tag + "_cd"][:5], S.Net(stiff, mu), rtol=1e-8, atol=1e-10)
    _, gp_t2, _ = S.cnf_adjoint(G[tag + "_zback"][:5], G[tag + "_cz"][:5], G[tag + "_cd"][:5], t2, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(gp_t2, gp_e2, atol=1e-9 * np.abs(gp_e2).max())


def test_deposit_rows_are_as_long_as_the_weights_need(golden):
    """ff_radial.h header slot 5: the table kernel picks 6, 8, 10 or 12 coefficients per deposit row from max|w1| (remainder of the
    expansion held at 4.5e-12); the tabulated adjoint's parameter gradient equals the direct kernel's for every choice."""
    G = golden["g4_cnf"]
    eta, mu = net_arrays(G, "")
    wmax = max(np.abs(eta[0]).max(), np.abs(mu[0]).max())
    tag, rt, at = "tol10", 1e-10, 1e-12
    seen = set()
    for target in (0.3, 1.0, 2.5, 5.0):      # max|w1| after scaling: h_d = 1/16 -> 1.5 w h_d = 0.028 ... 0.47
        sc = target / wmax
        e2, m2 = (eta[0] * sc, eta[1], eta[2]), (mu[0] * sc, mu[1], mu[2])
        tab = S.Net(e2, m2, table=True)
        assert tab.tab[3] == 0.0 and tab.tab[4] == 0.0
        nrow = int(tab.tab[5])
        xd = 1.5 * target / 16.0
        import math
        want = next((n for n in (6, 8, 10) if xd ** n / math.factorial(n) <= 4.5e-12), 12)
        assert nrow == want, (target, nrow, want)
        seen.add(nrow)
        _, gp_e, _ = S.cnf_adjoint(G[tag + "_zback"][:6], G[tag + "_cz"][:6], G[tag + "_cd"][:6], S.Net(e2, m2), rtol=rt, atol=at)
        _, gp_t, _ = S.cnf_adjoint(G[tag + "_zback"][:6], G[tag + "_cz"][:6], G[tag + "_cd"][:6], tab, rtol=rt, atol=at)
        np.testing.assert_allclose(gp_t, gp_e, atol=1e-10 * np.abs(gp_e).max())
    assert seen == {6, 8, 10, 12}


def test_radii_beyond_the_table_fall_back_to_direct_evaluation(golden):
    """walkers spread over +-60 (pair distances far beyond FF_TAB_RMAX = 32): the forward kernels evaluate those radii
    directly, the tabulated adjoint raises its off-table flag and the direct adjoint kernel redoes the call."""
    G = golden["g4_cnf"]
    eta, mu = net_arrays(G, "")
    exact, tab = S.Net(eta, mu), S.Net(eta, mu, table=True)
    rng = np.random.RandomState(5)
    z = rng.randn(7, 6, 2) * 25.0
    cz, cd = rng.randn(7, 6, 2), rng.randn(7)
    assert np.sqrt(((z[:, :, None] - z[:, None]) ** 2).sum(-1)).max() > 40
    xe, _ = S.cnf_generate(z, exact, rtol=1e-9, atol=1e-11); xt, _ = S.cnf_generate(z, tab, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(xt, xe, rtol=1e-12, atol=1e-11)
    gx_e, gp_e, _ = S.cnf_adjoint(z, cz, cd, exact, rtol=1e-9, atol=1e-11)
    gx_t, gp_t, _ = S.cnf_adjoint(z, cz, cd, tab, rtol=1e-9, atol=1e-11)
    np.testing.assert_array_equal(gp_t, gp_e)        # same kernel served both calls
    np.testing.assert_array_equal(gx_t, gx_e)


@pytest.mark.parametrize("nup,ndn,B", [(4, 4, 5), (6, 6, 1)])
def test_split_column_eloc_kernel(nup, ndn, B):
    """The default local-energy kernels of 8 and 12 particles against the oracle, with direct and tabulated radial functions:
    ff_eloc_split_kernel at n = 8 (two lanes per direction, half columns exchanged through LDS; B ragged: G = 2 walkers per
    wave), the one-walker-per-workgroup matrix-core kernel at n = 12 (the dispatcher's choice from 11 particles on)."""
    rng = np.random.default_rng(5 + nup)
    He, Hm = 16, 12
    eta = [rng.normal(size=He) * 0.5, rng.normal(size=He) * 0.3, rng.normal(size=He) * 0.05]
    mu = [rng.normal(size=Hm) * 0.5, rng.normal(size=Hm) * 0.3, rng.normal(size=Hm) * 0.05]
    x = rng.normal(size=(B, nup + ndn, 2)) * 1.2
    ref = O.eloc(x, nup, ndn, O.Net(eta, mu), 2.0, rtol=1e-11, atol=1e-13)
    for table in (False, True):
        r = S.eloc(x, nup, ndn, S.Net(eta, mu, table=table), 2.0, rtol=1e-9, atol=1e-11)
        assert r["stats"][3] == 0
        np.testing.assert_allclose(r["eloc"], ref["eloc"], rtol=1e-8)
        np.testing.assert_allclose(r["grad"], ref["grad"], atol=1e-8)
        np.testing.assert_allclose(r["lap"], ref["lap"], rtol=1e-7)


def test_walker_schedule_changes_nothing_but_the_order_of_work(golden):
    """ff_ode.walker_cost / walker_order + ff_walker_order: a permutation by descending cost; every walker's results
    are bit-identical whatever the processing order (each walker adapts its own steps, so its arithmetic does not
    depend on its wave-mates), only the parameter gradient (a sum over walkers) moves by rounding."""
    G = golden["g5_gsvmc"]
    name = "z2_nt"
    eta, mu = net_arrays(G, name + "_")
    x = G[name + "_x"][:13]
    B = len(x)
    cost = np.array([3, 7, 7, 0, 40, 5, 31, 5, 2, 99, 7, 1, 5], dtype=np.int32)
    order = S.walker_order(cost)
    assert sorted(order.tolist()) == list(range(B))
    c = np.clip(cost[order], 0, 31)
    assert np.all(c[:-1] >= c[1:])
    big = np.random.default_rng(3).integers(-2, 40, size=7001).astype(np.int32)      # several segments, ragged tail
    ob = S.walker_order(big)
    cb = np.clip(big[ob], 0, 31)
    assert sorted(ob.tolist()) == list(range(len(big))) and np.all(cb[:-1] >= cb[1:])
    assert np.array_equal(ob, S.walker_order(big))                                   # deterministic
    for table in (False, True):
        net = S.Net(eta, mu, table=table)
        st0, st1 = np.full(B, -1, np.int32), np.full(B, -1, np.int32)
        x0, _ = S.cnf_generate(x, net, steps=st0)
        x1, _ = S.cnf_generate(x, net, steps=st1, order=order)
        assert np.array_equal(x0, x1) and np.array_equal(st0, st1) and st0.min() >= 1
        r0 = S.eloc(x, 3, 3, net, 2.0, steps=st0)
        r1 = S.eloc(x, 3, 3, net, 2.0, steps=st1, order=S.walker_order(st0))
        for k in ("eloc", "grad", "lap", "z", "dlogp", "glogp0"):
            assert np.array_equal(r0[k], r1[k]), k
        assert np.array_equal(st0, st1)
        w = (r0["eloc"] - r0["eloc"].mean()) / B
        gx0, gp0, _ = S.cnf_adjoint(r0["z"], w[:, None, None] * r0["glogp0"], -w, net, steps=st0)
        gx1, gp1, _ = S.cnf_adjoint(r0["z"], w[:, None, None] * r0["glogp0"], -w, net, steps=st1, order=S.walker_order(st0))
        assert np.array_equal(gx0, gx1) and np.array_equal(st0, st1)
        np.testing.assert_allclose(gp1, gp0, rtol=1e-10, atol=1e-14)


def test_step_size_warm_start_semantics(golden):
    """ff_ode.walker_h_init / walker_h_scale / walker_h_out in the kernels' host build: non-positive entries start cold
    (bit-identical), positive ones skip the probe stage and open with that step; h_out is the largest accepted step."""
    G = golden["g5_gsvmc"]
    eta, mu = net_arrays(G, "z2_nt_")
    net = S.Net(eta, mu, table=True)
    x = G["z2_nt_x"][:7]
    B = len(x)
    try:
        hg = np.zeros(B)
        S.warm(h_out=hg)
        z0, st0 = S.cnf_generate(x, net)
        assert np.all(hg > 0) and np.all(hg <= 1.0)
        S.warm(h_init=np.zeros(B))
        z1, st1 = S.cnf_generate(x, net)
        assert np.array_equal(z0, z1) and st0[0] == st1[0]
        S.warm(h_init=hg, h_scale=0.75)
        z2, st2 = S.cnf_generate(x, net)
        assert st2[0] < st0[0] and np.abs(z2 - z0).max() < 1e-6
        he = np.zeros(B)
        S.warm()
        r0 = S.eloc(x, 3, 3, net, 2.0)
        S.warm(h_init=hg, h_scale=0.6, h_out=he)
        r1 = S.eloc(x, 3, 3, net, 2.0)
        assert r1["stats"][0] < r0["stats"][0] and np.all(he > 0)
        np.testing.assert_allclose(r1["eloc"], r0["eloc"], rtol=1e-6)
        w = (r0["eloc"] - r0["eloc"].mean()) / B
        S.warm()
        _, gp0, s0 = S.cnf_adjoint(r0["z"], w[:, None, None] * r0["glogp0"], -w, net)
        S.warm(h_init=he, h_scale=1.25)
        _, gp1, s1 = S.cnf_adjoint(r0["z"], w[:, None, None] * r0["glogp0"], -w, net)
        assert s1[0] < s0[0]
        np.testing.assert_allclose(gp1, gp0, rtol=1e-5, atol=1e-9 * np.abs(gp0).max())
    finally:
        S.warm()


@pytest.mark.parametrize("nup,ndn", [(3, 3), (3, 0)])
def test_mcmc_continue_is_the_same_chain_from_given_walkers(nup, ndn):
    """ff_mcmc_continue (persistent walkers): from x_init, with the Philox stream of (seed, offset) from step 1 on --
    i.e. exactly ff_mcmc_sample_noise fed x_init and the materialised stream; 0 steps return x_init and its log-prob."""
    rng = np.random.default_rng(11)
    n, B, steps = nup + ndn, 9, 7
    x0 = rng.normal(size=(B, n, 2))
    _, g, u = S.rng_fill(B, n, steps, seed=42, offset=3)
    xr, lr, acc = S.mcmc_noise(x0, g, u, nup, ndn)
    xc, lc, cnt = S.mcmc_continue(x0, nup, ndn, steps, seed=42, offset=3)
    assert np.array_equal(xc, xr) and np.array_equal(lc, lr) and np.array_equal(cnt, acc.sum(0))
    xz, lz, _ = S.mcmc_continue(x0, nup, ndn, 0, seed=1)
    assert np.array_equal(xz, x0)
    np.testing.assert_allclose(lz, S.logprob(x0, nup, ndn)[0], rtol=1e-13)


def test_failed_integration_poisons_its_outputs(golden):
    """a walker that hits max_steps (or a NaN error norm) must not hand back its partial state -- every
    ODE entry point writes NaN for it, independent of the stats word."""
    G = golden["g5_gsvmc"]
    eta, mu = net_arrays(G, "z2_nt_")
    x = G["z2_nt_x"][:6]
    for table in (False, True):
        net = S.Net(eta, mu, table=table)
        try:
            S.warm(max_steps=1)          # cold start: the first, tiny step is accepted, the walker is not done -> failure
            y, st = S.cnf_generate(x, net)
            assert st[3] == 1 and np.isnan(y).all()
            z, dl, st = S.cnf_delta_logp(x, net)
            assert st[3] == 1 and np.isnan(z).all() and np.isnan(dl).all()
            r = S.eloc(x, 3, 3, net, 2.0)
            assert r["stats"][3] == 1 and np.isnan(r["eloc"]).all()
            gx, gp, st = S.cnf_adjoint(x, np.ones_like(x), np.ones(len(x)), net)
            assert st[3] == 1 and np.isnan(gx).all() and np.isnan(gp).any()
        finally:
            S.warm()
        y, st = S.cnf_generate(x, net)
        assert st[3] == 0 and np.isfinite(y).all()


def test_uniform_warm_start_entry(golden):
    """ff_ode.walker_h_uniform: ONE first step size for every walker == the same value repeated per walker."""
    G = golden["g5_gsvmc"]
    eta, mu = net_arrays(G, "z2_nt_")
    net = S.Net(eta, mu, table=True)
    z = G["z2_nt_z"][:7]
    try:
        S.warm(h_init=np.full(7, 0.4), h_scale=0.75)
        xa, sa = S.cnf_generate(z, net)
        S.warm(h_init=np.array([0.4]), h_scale=0.75, uniform=True)
        xb, sb = S.cnf_generate(z, net)
    finally:
        S.warm()
    assert (xa == xb).all() and (sa == sb).all()


@pytest.mark.parametrize("nup,ndn,He,Hm,B", [(4, 3, 16, 12, 5), (1, 0, 8, 8, 19), (5, 4, 10, 6, 1), (6, 5, 8, 8, 1), (3, 3, 100, 70, 3)])
def test_any_particle_number_and_hidden_width(nup, ndn, He, Hm, B):
    """VERDICT r01 missing #3: the reference is shape-generic (src/equivariant_funs.py:17-102, --Deta/--Dmu in
    src/FermionHO2D.py:24-27): odd particle numbers (row-layout local-energy kernel), n = 1, hidden widths beyond 64
    (one direct-adjoint launch per chunk of units) -- flow, log-density, local energy and adjoint against the oracle,
    tabulated and direct radial functions."""
    n = nup + ndn
    rng = np.random.default_rng(100 * nup + ndn + He)
    sc = 4.0 / np.sqrt(He)
    eta = [rng.normal(size=He) * 0.5, rng.normal(size=He) * 0.3, rng.normal(size=He) * 0.05 * sc]
    mu = [rng.normal(size=Hm) * 0.5, rng.normal(size=Hm) * 0.3, rng.normal(size=Hm) * 0.05 * sc]
    z = rng.normal(size=(B, n, 2)) * 1.2
    onet = O.Net(eta, mu)
    xo, _ = O.cnf_generate(z, onet, rtol=1e-11, atol=1e-13)
    ref = O.eloc(xo, nup, ndn, onet, 2.0, rtol=1e-11, atol=1e-13)
    zo, dlo, _ = O.cnf_delta_logp(xo, onet, rtol=1e-11, atol=1e-13)
    az, ad = rng.normal(size=z.shape), rng.normal(size=B)
    gxo, gpo, _ = O.cnf_adjoint(zo, dlo, az, ad, onet, rtol=1e-11, atol=1e-13)
    for table in (False, True):
        net = S.Net(eta, mu, table=table)
        x, st = S.cnf_generate(z, net, rtol=1e-9, atol=1e-11)
        assert st[3] == 0
        np.testing.assert_allclose(x, xo, atol=1e-8)
        zb, dl, _ = S.cnf_delta_logp(xo, net, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(zb, zo, atol=1e-8); np.testing.assert_allclose(dl, dlo, atol=1e-8)
        r = S.eloc(xo, nup, ndn, net, 2.0, rtol=1e-9, atol=1e-11)
        assert r["stats"][3] == 0
        np.testing.assert_allclose(r["eloc"], ref["eloc"], rtol=1e-7)
        np.testing.assert_allclose(r["grad"], ref["grad"], atol=1e-7)
        gx, gp, st = S.cnf_adjoint(zo, az, ad, net, rtol=1e-9, atol=1e-11)
        assert st[3] == 0
        np.testing.assert_allclose(gx, gxo, atol=1e-7)
        np.testing.assert_allclose(gp, gpo, atol=2e-7 * max(1.0, np.abs(gpo).max()))


@pytest.mark.parametrize("kind", ["mfma", "rows", "columns"])
def test_three_local_energy_kernels_agree(golden, kind, monkeypatch):
    """FF_ELOC_KERNEL selects the matrix-core, row-layout or column-sweep local-energy kernel: the same numbers from all three."""
    import subprocess, sys, json, os
    code = ("import numpy as np, json; from tests.hostsim import simlib as S; from tests.common import net_arrays;"
            "G=np.load('tests/golden/g5_gsvmc.npz'); eta,mu=net_arrays(G,'z2_nt_');"
            "r=S.eloc(G['z2_nt_x'][:4],3,3,S.Net(eta,mu,table=True),2.0,rtol=1e-9,atol=1e-11);"
            "print(json.dumps([r['eloc'].tolist(), r['lap'].tolist(), r['stats'].tolist()]))")
    env = dict(os.environ, FF_ELOC_KERNEL=kind)
    out = subprocess.check_output([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    el, lap, st = json.loads(out.decode().strip().splitlines()[-1])
    G = golden["g5_gsvmc"]
    assert st[3] == 0
    np.testing.assert_allclose(el, G["z2_nt_Eloc"][:4], rtol=1e-8)
    np.testing.assert_allclose(lap, G["z2_nt_lap"][:4], rtol=1e-7, atol=1e-6)


@pytest.mark.parametrize("nup,ndn,B", [(1, 1, 5), (2, 1, 2), (2, 2, 3), (3, 2, 2)])
def test_matrix_core_kernel_every_block_count(nup, ndn, B):
    """ff_eloc_mfma_kernel for 2 ... 5 particles (one to three 4 x 4 blocks per side, coordinates that do not fill the last block,
    one or two radius slots per lane, walker groups with idle slots) against the oracle; 6 particles: the test above."""
    import subprocess, sys, json, os
    code = ("import numpy as np, json; from tests.hostsim import simlib as S; from oracle import oracle as O;"
            f"nup,ndn,B={nup},{ndn},{B}; n=nup+ndn; rng=np.random.default_rng(7*n+B);"
            "eta=[rng.normal(size=10)*0.5, rng.normal(size=10)*0.3, rng.normal(size=10)*0.06];"
            "mu=[rng.normal(size=6)*0.5, rng.normal(size=6)*0.3, rng.normal(size=6)*0.06];"
            "x=rng.normal(size=(B,n,2))*1.2;"
            "ref=O.eloc(x,nup,ndn,O.Net(eta,mu),2.0,rtol=1e-11,atol=1e-13);"
            "out=[]\n"
            "for table in (True, False):\n"
            "    r=S.eloc(x,nup,ndn,S.Net(eta,mu,table=table),2.0,rtol=1e-9,atol=1e-11);"
            "    out.append([r['eloc'].tolist(), r['grad'].tolist(), r['lap'].tolist(), r['stats'].tolist()])\n"
            "print(json.dumps([out, ref['eloc'].tolist(), ref['grad'].tolist(), ref['lap'].tolist()]))")
    env = dict(os.environ, FF_ELOC_KERNEL="mfma")
    out = subprocess.check_output([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    res, el, gr, lap = json.loads(out.decode().strip().splitlines()[-1])
    for r in res:
        assert r[3][3] == 0
        np.testing.assert_allclose(r[0], el, rtol=1e-7)
        np.testing.assert_allclose(r[1], gr, atol=1e-7)
        np.testing.assert_allclose(r[2], lap, rtol=1e-6, atol=1e-6)


def test_rejected_first_steps_do_not_couple_the_walkers_of_a_wave():
    """The four walkers of a matrix-core wave advance in lockstep (csrc/ff_eloc_mfma.h): when one of them rejects a step, all pass through
    stage 0 again -- which must not change anybody's numbers.  Nine walkers of 3 + 3 particles opened with the WHOLE interval as first
    step (rejected for most of them) in one call against the same nine, one call each: every output identical, rejections and attempted
    steps counted once; also with an explicit order and as a batch of five; and against the oracle.  (Written for round 6's retry queue
    -- docs/attic/retry_queue_r06.patch, DESIGN.md 3p -- whose retried integrations had to be the in-place ones bit for bit; kept as the
    invariant it checks.)"""
    import subprocess, sys, json, os
    code = ("import numpy as np, json; from tests.hostsim import simlib as S; from tests.common import net_arrays; from oracle import oracle as O;"
            "G=np.load('tests/golden/g5_gsvmc.npz'); eta,mu=net_arrays(G,'z2_nt_'); x=G['z2_nt_x'][:9]; net=S.Net(eta,mu,table=True);"
            "ref=O.eloc(x,3,3,O.Net(eta,mu),2.0,rtol=1e-11,atol=1e-13); keys=('eloc','logp','lap','grad','z','dlogp','glogp0')\n"
            "def run(xx, order=None):\n"
            "    h=np.ones(len(xx)); steps=np.zeros(len(xx),dtype=np.int32); S.warm(h_init=h, h_scale=1.0)\n"
            "    try:\n"
            "        r=S.eloc(xx,3,3,net,2.0,rtol=1e-7,atol=1e-9,steps=steps,order=order)\n"
            "    finally:\n"
            "        S.warm()\n"
            "    return [[r[k].tolist() for k in keys], r['stats'].tolist(), steps.tolist()]\n"
            "batch=run(x); ordered=run(x, order=np.array([8,3,5,0,7,1,6,2,4],dtype=np.int32)); five=run(x[:5]); singles=[run(x[i:i+1]) for i in range(9)]\n"
            "print(json.dumps([batch, ordered, five, singles, ref['eloc'].tolist()]))")
    env = dict(os.environ, FF_ELOC_KERNEL="mfma")
    out = subprocess.check_output([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    batch, ordered, five, singles, el = json.loads(out.decode().strip().splitlines()[-1])
    assert batch[1][3] == 0 and batch[1][2] >= 3, batch[1]                        # no failure; at least three first steps were rejected
    for k in range(7):
        for i in range(9):
            assert batch[0][k][i] == singles[i][0][k][0], (k, i)                  # bit for bit the in-place retry
            assert ordered[0][k][i] == batch[0][k][i], (k, i)
        for i in range(5):
            assert five[0][k][i] == batch[0][k][i], (k, i)
    assert batch[1][2] == sum(sg[1][2] for sg in singles)                        # rejected steps: counted once per walker
    assert batch[2] == [sg[2][0] for sg in singles]                              # attempted steps per walker (walker_cost)
    np.testing.assert_allclose(batch[0][0], el, rtol=2e-6)


def test_local_energy_routing_by_cost_class():
    """launch_mfma (csrc/ff_cnf_fwd.hip): with cost classes the walkers of class >= ff_ode.heavy_class (default 12 below 12 coordinates) are integrated by the
    one-walker-per-wave kernel (csrc/ff_wide.hip), the others by the four-walkers-per-wave matrix-core kernel.  Every walker is
    integrated exactly once and agrees with the oracle; a walker's result depends on its own class only -- not on the order of
    work, not on the rest of the batch."""
    import subprocess, sys, json, os
    code = ("import numpy as np, json, os; from tests.hostsim import simlib as S; from oracle import oracle as O;"
            "rng=np.random.default_rng(5); eta=[rng.normal(size=6)*0.5, rng.normal(size=6)*0.3, rng.normal(size=6)*0.06];"
            "mu=[rng.normal(size=6)*0.5, rng.normal(size=6)*0.3, rng.normal(size=6)*0.06];"
            "x=rng.normal(size=(6,2,2))*1.2; cls=np.array([3,14,5,12,2,20],dtype=np.int32);"
            "ref=O.eloc(x,1,1,O.Net(eta,mu),2.0,rtol=1e-11,atol=1e-13); net=S.Net(eta,mu,table=True); out=[]\n"
            "def run(xx, cc, order=None, heavy_class=0):\n"
            "    S.warm(wclass=cc, sens_tol=1.0, sens_class=0, heavy_class=heavy_class)\n"
            "    try:\n"
            "        r=S.eloc(xx,1,1,net,2.0,rtol=1e-9,atol=1e-11,order=order)\n"
            "    finally:\n"
            "        S.warm()\n"
            "    return [r['eloc'].tolist(), r['grad'].tolist(), r['stats'].tolist()]\n"
            "out.append(run(x, cls))\n"
            "out.append(run(x, cls, order=np.array([5,1,3,2,0,4],dtype=np.int32)))\n"
            "out.append(run(x[[1,4]], cls[[1,4]]))\n"
            "out.append(run(x, cls, heavy_class=-1))\n"
            "print(json.dumps([out, ref['eloc'].tolist(), ref['grad'].tolist()]))")
    env = dict(os.environ, FF_ELOC_KERNEL="mfma")
    out = subprocess.check_output([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    (plain, ordered, subset, unrouted), el, gr = json.loads(out.decode().strip().splitlines()[-1])
    for r in (plain, ordered, unrouted):
        assert r[2][3] == 0
        np.testing.assert_allclose(r[0], el, rtol=1e-7)
        np.testing.assert_allclose(r[1], gr, atol=1e-7)
    assert plain[0] == ordered[0] and plain[1] == ordered[1]                       # the order of work is invisible
    assert subset[0] == [plain[0][1], plain[0][4]]                                 # ... and so is the rest of the batch
    light, heavy = [0, 2, 4], [1, 3, 5]
    assert [plain[0][i] for i in light] == [unrouted[0][i] for i in light]         # light walkers: the same kernel either way
    assert any(plain[0][i] != unrouted[0][i] for i in heavy)                       # heavy ones did go through the other kernel


def test_state_sums_of_the_finite_temperature_estimator():
    """ff_state_sums: per-state sums over the sorted state list (src/VMC.py:164-169), including empty states and the
    everything-in-state-0 case of beta = 10."""
    rng = np.random.default_rng(0)
    for ws in (np.sort(rng.integers(0, 21, 777)), np.zeros(300, dtype=int), np.full(5, 20)):
        e = rng.normal(size=len(ws)) + 30
        s, c = S.state_sums(e, ws, 21)
        want_c = np.bincount(ws, minlength=21)
        want_s = np.array([e[ws == k].sum() for k in range(21)])
        assert (c == want_c).all()
        np.testing.assert_allclose(s, want_s, rtol=1e-13, atol=1e-12)


HO3D_E = np.array([s + 1.5 for s in range(8) for nx in range(s + 1) for ny in range(s + 1 - nx)])


def test_ho3d_logprob_and_sampler(golden):
    """Groundwork for the 3-D trap (no upstream code): ff_logprob3d against the oracle's jets, the known-answer
    test of tests/test_basedist.py:5-60 one dimension up (E_loc == sum of orbital energies at random points), and the
    d = 3 Metropolis chain bit for bit against the oracle on explicit noise."""
    rng = np.random.RandomState(3)
    for nup, ndn in ((1, 0), (4, 0), (3, 6), (10, 10)):
        iu = np.sort(rng.choice(20, nup, replace=False)); idn = np.sort(rng.choice(20, ndn, replace=False)) if ndn else None
        x = rng.randn(5, nup + ndn, 3)
        lp, g, lap = S.logprob3d(x, nup, ndn, tab_up=iu, tab_dn=idn)
        lpo, go, lapo = O.logprob3d(x, nup, ndn, tab_up=iu, tab_dn=idn)
        np.testing.assert_allclose(lp, lpo, atol=1e-11)
        np.testing.assert_allclose(g, go, rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(lap, lapo, rtol=1e-8, atol=1e-6)
        eloc = -0.25 * lap - 0.125 * (g ** 2).sum(axis=(1, 2)) + 0.5 * (x ** 2).sum(axis=(1, 2))
        want = HO3D_E[iu].sum() + (HO3D_E[idn].sum() if ndn else 0.0)
        np.testing.assert_allclose(eloc, want, rtol=1e-8)
    # (the sampler runs sixteen lanes per determinant: every column of the LU is a handful of emulated lane exchanges, i.e.
    #  workgroup barriers of 64 host threads -- small batches here, the GPU tests run it at size)
    B, Ssteps = 5, 8
    g0, g, u = rng.randn(B, 5, 3), rng.randn(Ssteps, B, 5, 3), rng.rand(Ssteps, B)
    x, lp, acc = S.mcmc_noise3d(g0, g, u, 3, 2)
    xo, lpo, acco = O.mcmc_noise3d(g0, g, u, 3, 2)
    assert (acc == acco).all() and (x == xo).all()
    np.testing.assert_allclose(lp, lpo, atol=1e-12)
    # orbitals of the fifth shell and beyond: Hermite degrees >= 4, past the four the sampler tabulates per lane in LDS (round 5:
    # those are re-run through the recurrence) -- same chain as the oracle's, bit for bit, in d = 3 and in d = 2
    iu, idn = np.array([0, 3, 21, 30]), np.array([1, 25, 34])
    g0, g, u = rng.randn(3, 7, 3), rng.randn(4, 3, 7, 3), rng.rand(4, 3)
    x, lp, acc = S.mcmc_noise3d(g0, g, u, 4, 3, tab_up=iu, tab_dn=idn)
    xo, lpo, acco = O.mcmc_noise3d(g0, g, u, 4, 3, tab_up=iu, tab_dn=idn)
    assert (acc == acco).all() and (x == xo).all()
    np.testing.assert_allclose(lp, lpo, atol=1e-12)
    iu, idn = np.array([0, 2, 5, 9, 12, 17, 27]), np.array([1, 4, 10, 14, 20, 22])
    g0, g, u = rng.randn(2, 13, 2), rng.randn(3, 2, 13, 2), rng.rand(3, 2)
    x, lp, acc = S.mcmc_noise(g0, g, u, 7, 6, tab_up=iu, tab_dn=idn)
    xo, lpo, acco = O.mcmc_noise(g0, g, u, 7, 6, tab_up=iu, tab_dn=idn)
    assert (acc == acco).all() and (x == xo).all()
    np.testing.assert_allclose(lp, lpo, atol=1e-12)
    xs, lps, cnt = S.mcmc3d(6, 4, 4, 8, 99)          # Philox sampler: closed shells 0..1 for both spins
    assert np.isfinite(xs).all() and 0 < cnt.sum() <= 6 * 8
    np.testing.assert_allclose(lps, O.logprob3d(xs, 4, 4, derivs=False), atol=1e-11)
    # ... and it is the noise-fed chain on its own stream (ff_rng_fill3d), walkers and accept counts bit for bit -- also at the
    # largest walker the entry point admits: 11 + 11 and 12 + 12 particles in d = 3 are 66 / 72 coordinates, more than the 64 normals
    # the kernel's LDS staging held until round 5 (walker 0's normals overwrote walker 1's)
    for nup, ndn, B, steps in ((4, 4, 6, 8), (11, 11, 3, 3), (12, 12, 2, 2)):
        h0, h, hu = S.rng_fill(B, nup + ndn, steps, 99, dim=3)
        x1, lp1, a1 = S.mcmc_noise3d(h0, h, hu, nup, ndn)
        x2, lp2, c2 = S.mcmc3d(B, nup, ndn, steps, 99)
        assert (x1 == x2).all() and (a1.sum(0) == c2).all(), (nup, ndn)


def test_fp32_backflow_error_against_fp64(golden):
    """The fp32 instantiation (ff_backflow_v_div_f32) against the fp64 kernel on the benchmark's weights: relative error
    of v and div v at the 1e-6 level expected of single precision (reported in DESIGN.md 7)."""
    G = golden["g3_backflow"]
    eta, mu = net_arrays(G, "c1_")
    x = G["c1_x"]
    v64, d64 = S.backflow(x, S.Net(eta, mu))
    v32, d32 = S.backflow_f32(x, S.Net(eta, mu))
    ev = np.abs(v32 - v64).max() / np.abs(v64).max(); ed = np.abs(d32 - d64).max() / np.abs(d64).max()
    assert 1e-9 < ev < 2e-5 and ed < 2e-5, (ev, ed)


@pytest.mark.parametrize("nup,ndn,B", [(2, 2, 7), (1, 1, 11), (2, 1, 3), (4, 0, 2)])
def test_local_energy_in_three_dimensions(nup, ndn, B):
    """d = 3 through the fused kernels (flow, log-density and adjoint templates with D = 3, row-layout sensitivities, the
    d = 3 Slater finish): zero flow and Z = 0 give E_loc = sum of HO3D orbital energies exactly; with a flow, E_loc, grad,
    the flow itself and the parameter gradient agree with the oracle."""
    n = nup + ndn
    rng = np.random.default_rng(10 * nup + ndn)
    He = Hm = 8
    zero = [np.zeros(He), np.zeros(He), np.zeros(He)]
    x = rng.normal(size=(B, n, 3))
    want = HO3D_E[:nup].sum() + HO3D_E[:ndn].sum()
    for table in (False, True):
        r = S.eloc3d(x, nup, ndn, S.Net(zero, zero, table=table), 0.0)
        assert r["stats"][3] == 0
        np.testing.assert_allclose(r["eloc"], want, rtol=1e-9)
    eta = [rng.normal(size=He) * 0.5, rng.normal(size=He) * 0.3, rng.normal(size=He) * 0.05]
    mu = [rng.normal(size=Hm) * 0.5, rng.normal(size=Hm) * 0.3, rng.normal(size=Hm) * 0.05]
    onet = O.Net(eta, mu)
    xo, _ = O.cnf_generate(x, onet, rtol=1e-11, atol=1e-13)
    ref = O.eloc3d(xo, nup, ndn, onet, 2.0, rtol=1e-11, atol=1e-13)
    zo, dlo, _ = O.cnf_delta_logp(xo, onet, rtol=1e-11, atol=1e-13)
    az, ad = rng.normal(size=x.shape), rng.normal(size=B)
    gxo, gpo, _ = O.cnf_adjoint(zo, dlo, az, ad, onet, rtol=1e-11, atol=1e-13)
    for table in (False, True):
        net = S.Net(eta, mu, table=table)
        xs, st = S.cnf_generate(x, net, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(xs, xo, atol=1e-8)
        r = S.eloc3d(xo, nup, ndn, net, 2.0, rtol=1e-9, atol=1e-11)
        assert r["stats"][3] == 0
        np.testing.assert_allclose(r["eloc"], ref["eloc"], rtol=1e-7)
        np.testing.assert_allclose(r["grad"], ref["grad"], atol=1e-7)
        np.testing.assert_allclose(r["logp"], ref["logp"], atol=1e-8)
        gx, gp, st = S.cnf_adjoint(zo, az, ad, net, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(gx, gxo, atol=1e-7)
        np.testing.assert_allclose(gp, gpo, atol=2e-7 * max(1.0, np.abs(gpo).max()))


def test_sensitivity_tolerance_and_predictive_step_bound(golden):
    """ff_ode.walker_class / sens_tol in the kernels' host build: factor 1 (and anything below), or no walker at or below
    the class threshold, is bit-identical to no policy; a factor of 10 on the sensitivity components takes fewer evaluations
    on ordinary walkers and leaves E_loc within 1e-6 of a tight solve (bar: 1e-5); the choice is per walker.  Second half: a walker with a particle that ENDS next to the origin (where mu(|x|) x is only
    C^1) -- the step sizes shrink geometrically towards t0; with the predictive bound of ff_stepper::decide (Gustafsson) the
    controller follows the trend instead of failing every other step (11 rejected of 26 before, <= 5 now)."""
    G = golden["g5_gsvmc"]
    eta, mu = net_arrays(G, "z2_nt_")
    net = S.Net(eta, mu, table=True)
    x = G["z2_nt_x"][:7]
    B = len(x)
    tight = S.eloc(x, 3, 3, net, 2.0, rtol=1e-10, atol=1e-12)["eloc"]
    try:
        r0 = S.eloc(x, 3, 3, net, 2.0)
        cls = np.full(B, 3, dtype=np.int32)
        S.warm(wclass=cls, sens_tol=1.0, sens_class=8)
        r1 = S.eloc(x, 3, 3, net, 2.0)
        S.warm(wclass=cls, sens_tol=10.0, sens_class=2)        # nobody is at or below class 2
        r1b = S.eloc(x, 3, 3, net, 2.0)
        S.warm(wclass=cls, sens_tol=10.0, sens_class=8)
        r2 = S.eloc(x, 3, 3, net, 2.0)
        mixed = np.where(np.arange(B) % 2 == 0, 3, 9).astype(np.int32)
        S.warm(wclass=mixed, sens_tol=10.0, sens_class=8)
        r3 = S.eloc(x, 3, 3, net, 2.0)
    finally:
        S.warm()
    assert np.array_equal(r0["eloc"], r1["eloc"]) and np.array_equal(r0["eloc"], r1b["eloc"]) and r0["stats"][0] == r1["stats"][0]
    assert r2["stats"][0] < r0["stats"][0] and r2["stats"][3] == 0
    assert np.abs(r2["eloc"] / tight - 1).max() < 1e-6 and np.abs(r0["eloc"] / tight - 1).max() < 1e-6
    assert np.array_equal(r3["eloc"][1::2], r0["eloc"][1::2]) and np.array_equal(r3["eloc"][0::2], r2["eloc"][0::2])   # per walker
    # a particle that ends 2e-4 from the origin
    z = G["z2_nt_x"][:1].copy()
    z[0, 0] = 2e-4 * np.array([0.6, 0.8])
    xh, _ = S.cnf_generate(z, net)
    rh = S.eloc(xh, 3, 3, net, 2.0)
    th = S.eloc(xh, 3, 3, net, 2.0, rtol=1e-10, atol=1e-12)["eloc"]
    assert rh["stats"][3] == 0 and rh["stats"][2] <= 5 and rh["stats"][1] >= 10, rh["stats"]
    assert abs(rh["eloc"][0] / th[0] - 1) < 1e-6


def test_energy_estimator_and_energy_seeded_adjoint(golden):
    """ff_reduce_energy + ff_energy_finish == mean / centred sum of squares / mean(logp (e - E)) (src/VMC.py:56-59), for any
    shift; ff_cnf_adjoint_energy == ff_cnf_adjoint on the seeds w * glogp0, -w with w = (e - E) / n formed by the caller."""
    G = golden["g5_gsvmc"]
    eta, mu = net_arrays(G, "z2_nt_")
    net = S.Net(eta, mu, table=True)
    x = G["z2_nt_x"][:9]
    r = S.eloc(x, 3, 3, net, 2.0)
    e, lp = r["eloc"], r["logp"]
    n = len(e)
    for shift in (0.0, float(e.mean()) + 0.3, 1e3):
        sums = S.reduce_energy(e, lp, shift)
        est = S.energy_finish(sums, shift, n)
        np.testing.assert_allclose(est[0], e.mean(), rtol=1e-13)
        np.testing.assert_allclose(est[1], ((e - e.mean()) ** 2).sum(), rtol=1e-9 if shift < 100 else 1e-6)
        np.testing.assert_allclose(est[2], (lp * (e - e.mean())).mean(), rtol=1e-9 if shift < 100 else 1e-6, atol=1e-12)
    # two "ranks": the sums add up
    sa, sb = S.reduce_energy(e[:4], lp[:4], 2.0), S.reduce_energy(e[4:], lp[4:], 2.0)
    np.testing.assert_allclose(S.energy_finish(sa + sb, 2.0, n), S.energy_finish(S.reduce_energy(e, lp, 2.0), 2.0, n), rtol=1e-12)
    E = float(e.mean())
    w = (e - E) / n
    gx0, gp0, st0 = S.cnf_adjoint(r["z"], w[:, None, None] * r["glogp0"], -w, net)
    gx1, gp1, st1 = S.cnf_adjoint_energy(r["z"], r["glogp0"], e, E, 1.0 / n, net)
    assert st0[3] == 0 and st1[3] == 0
    np.testing.assert_allclose(gx1, gx0, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(gp1, gp0, rtol=1e-12, atol=1e-15)


def test_finite_temperature_estimator_kernels(golden):
    """ff_beta_state_partials + ff_beta_finish == the formulas of BetaVMC.forward (src/VMC.py:146-171) evaluated with numpy;
    ff_cnf_adjoint_energy with a per-state baseline == ff_cnf_adjoint on the seeds (e - mean_e[state]) / n."""
    rng = np.random.default_rng(11)
    ns, B, beta = 7, 403, 3.0
    logits = rng.normal(size=ns)
    ws = np.sort(rng.choice(ns, size=B, p=[0.5, 0.2, 0.1, 0.1, 0.05, 0.05, 0.0])).astype(np.int32)     # state 6 is empty
    e = 30.0 + rng.normal(size=B) * 3.0
    lp = -20.0 + rng.normal(size=B)
    est, gphi, mean_e, lpa = S.beta_estimator(e, lp, ws, logits, beta, shift=29.0)
    lsm = logits - (np.log(np.exp(logits - logits.max()).sum()) + logits.max())
    np.testing.assert_allclose(lpa, lsm, rtol=1e-13, atol=1e-14)
    f = e + lsm[ws] / beta
    cnt = np.bincount(ws, minlength=ns).astype(float)
    sums = np.bincount(ws, weights=e, minlength=ns)
    me = sums / np.maximum(cnt, 1.0)
    np.testing.assert_allclose(mean_e, me, rtol=1e-13)
    F = f.mean()
    want = [e.mean(), ((e - e.mean()) ** 2).sum(), F, ((f - F) ** 2).sum(), -(lsm[ws]).mean(), -(lsm * np.exp(lsm)).sum(),
            (lsm[ws] * (f - F)).mean(), (lp * (e - me[ws])).mean()]
    np.testing.assert_allclose(est, want, rtol=1e-9, atol=1e-10)
    cF = (sums + cnt * lsm / beta - cnt * F) / B
    np.testing.assert_allclose(gphi, cF - np.exp(lsm) * cF.sum(), rtol=1e-10, atol=1e-13)
    # per-state baseline inside the adjoint
    G = golden["g5_gsvmc"]
    eta, mu = net_arrays(G, "z2_nt_")
    net = S.Net(eta, mu, table=True)
    x = G["z2_nt_x"][:6]
    r = S.eloc(x, 3, 3, net, 2.0)
    st = np.array([0, 0, 1, 1, 1, 2], dtype=np.int32)
    m3 = np.array([r["eloc"][:2].mean(), r["eloc"][2:5].mean(), r["eloc"][5]])
    w = (r["eloc"] - m3[st]) / 6
    gx0, gp0, _ = S.cnf_adjoint(r["z"], w[:, None, None] * r["glogp0"], -w, net)
    gx1, gp1, _ = S.cnf_adjoint_energy(r["z"], r["glogp0"], r["eloc"], m3, 1.0 / 6, net, mean_index=st)
    np.testing.assert_allclose(gx1, gx0, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(gp1, gp0, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("nup,ndn,d,B,force", [(2, 1, 2, 2, True), (3, 2, 3, 1, True), (7, 6, 2, 1, False)])
def test_one_walker_per_workgroup_kernels(nup, ndn, d, B, force):
    """csrc/ff_wide.hip and csrc/ff_adj_wide.h in the host simulator (multi-wave workgroups; the 16x16x4 matrix instruction
    emulated per wave): flow, log-density, the matrix-core local-energy kernel and the adjoint (table and direct variants)
    against the oracle -- forced onto small systems (ff_set_kernel_family) and where they are the only kernels (13 particles)."""
    n = nup + ndn
    rng = np.random.default_rng(7 + n)
    He = Hm = 8
    eta = [rng.normal(size=He) * 0.5, rng.normal(size=He) * 0.3, rng.normal(size=He) * 0.05]
    mu = [rng.normal(size=Hm) * 0.5, rng.normal(size=Hm) * 0.3, rng.normal(size=Hm) * 0.05]
    x = rng.normal(size=(B, n, d)) * (1.0 if n < 10 else 0.7)
    onet = O.Net(eta, mu)
    tol = dict(rtol=1e-8, atol=1e-10)
    xo, _ = O.cnf_generate(x, onet, rtol=1e-11, atol=1e-13)
    zo, dlo, _ = O.cnf_delta_logp(xo, onet, rtol=1e-11, atol=1e-13)
    ref = (O.eloc3d if d == 3 else O.eloc)(xo, nup, ndn, onet, 2.0, rtol=1e-11, atol=1e-13)
    az, ad = rng.normal(size=x.shape), rng.normal(size=B)
    gxo, gpo, _ = O.cnf_adjoint(zo, dlo, az, ad, onet, rtol=1e-11, atol=1e-13)
    prev = S.lib().ff_set_kernel_family(1 if force else 0)
    try:
        for table in ((True, False) if n < 10 and d == 2 else (True,)):      # (the direct-evaluation variants: once, on the smallest system)
            net = S.Net(eta, mu, table=table)
            xs, st = S.cnf_generate(x, net, **tol)
            assert st[3] == 0
            np.testing.assert_allclose(xs, xo, atol=1e-7)
            zs, dls, st = S.cnf_delta_logp(xo, net, **tol)
            np.testing.assert_allclose(zs, zo, atol=1e-7)
            np.testing.assert_allclose(dls, dlo, atol=1e-7)
            r = (S.eloc3d if d == 3 else S.eloc)(xo, nup, ndn, net, 2.0, **tol)
            assert r["stats"][3] == 0
            np.testing.assert_allclose(r["eloc"], ref["eloc"], rtol=1e-6)
            np.testing.assert_allclose(r["grad"], ref["grad"], atol=1e-6 * max(1.0, np.abs(ref["grad"]).max()))
            np.testing.assert_allclose(r["logp"], ref["logp"], atol=1e-7)
            # ff_ode.compact_finish: the same kernel finishes its walkers in its epilogue -- against the separate finish kernels
            rc = S.eloc_nd(xo, nup, ndn, net, 2.0, compact=True, **tol)
            assert rc["stats"][3] == 0
            assert rc["workspace_bytes"] == (8 * (B * (n * d + 1) + 2) if n * d > 24 else S.lib().ff_eloc_workspace_bytes(C.c_int64(B), n, d))
            for k in ("logp", "grad", "lap", "V", "eloc", "glogp0", "z", "dlogp"):
                np.testing.assert_allclose(rc[k], r[k], rtol=1e-10, atol=1e-10 * max(1.0, np.abs(r[k]).max()), err_msg=k)
            gx, gp, st = S.cnf_adjoint(zo, az, ad, net, **tol)
            np.testing.assert_allclose(gx, gxo, atol=1e-6 * max(1.0, np.abs(gxo).max()))
            np.testing.assert_allclose(gp, gpo, atol=1e-6 * max(1.0, np.abs(gpo).max()))
    finally:
        S.lib().ff_set_kernel_family(prev)


def test_wide_adjoint_far_radii_take_the_overflow_list():
    """A pair 9-10 apart lies beyond the LDS part of the wide adjoint's deposit table (r >= 8): its deposits wait in the step's
    overflow list and reach the launch's global table when the step is accepted -- same gradient as the oracle's, and the tabulated
    kernel (not the direct fallback) must have served the call (same RHS-evaluation count as with the pair close)."""
    rng = np.random.default_rng(3)
    He = Hm = 8
    eta = [rng.normal(size=He) * 0.5, rng.normal(size=He) * 0.3, rng.normal(size=He) * 0.05]
    mu = [rng.normal(size=Hm) * 0.5, rng.normal(size=Hm) * 0.3, rng.normal(size=Hm) * 0.05]
    z = rng.normal(size=(2, 3, 2)) * 0.7
    z[1, 0, 0] += 9.5
    onet = O.Net(eta, mu)
    dl = np.zeros(2)
    az, ad = rng.normal(size=z.shape), rng.normal(size=2)
    gxo, gpo, _ = O.cnf_adjoint(z, dl, az, ad, onet, rtol=1e-11, atol=1e-13)
    prev = S.lib().ff_set_kernel_family(1)
    try:
        gx, gp, st = S.cnf_adjoint(z, az, ad, S.Net(eta, mu, table=True), rtol=1e-8, atol=1e-10)
    finally:
        S.lib().ff_set_kernel_family(prev)
    assert st[3] == 0
    np.testing.assert_allclose(gx, gxo, atol=1e-6 * max(1.0, np.abs(gxo).max()))
    np.testing.assert_allclose(gp, gpo, atol=1e-6 * max(1.0, np.abs(gpo).max()))


def test_adam_step_equals_torch_adam():
    """ff_adam_step (one launch for all tensors) against torch.optim.Adam's default implementation (src/FermionHO2D.py:61), fp64, over
    several steps, with and without weight decay, tensors of different sizes including more than one launch's worth (16)."""
    import torch
    rng = np.random.default_rng(4)
    for wd, nt in ((0.0, 6), (0.01, 19)):
        shapes = [(int(rng.integers(1, 70)),) for _ in range(nt)]
        p0 = [rng.normal(size=sh) for sh in shapes]
        tp = [torch.tensor(a.copy(), dtype=torch.float64, requires_grad=True) for a in p0]
        opt = torch.optim.Adam(tp, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
        mine = [a.copy() for a in p0]
        m = [np.zeros_like(a) for a in p0]; v = [np.zeros_like(a) for a in p0]
        for step in range(1, 8):
            gs = [rng.normal(size=sh) * (10.0 if step == 3 else 1.0) for sh in shapes]
            for t, g in zip(tp, gs):
                t.grad = torch.tensor(g.copy())
            opt.step()
            S.adam_step(mine, gs, m, v, 1e-2, 0.9, 0.999, 1e-8, wd, step)
            for a, t in zip(mine, tp):
                np.testing.assert_allclose(a, t.detach().numpy(), rtol=2e-15, atol=1e-17)
        st = opt.state[tp[0]]
        np.testing.assert_allclose(m[0], st["exp_avg"].numpy(), rtol=1e-15, atol=1e-18)
        np.testing.assert_allclose(v[0], st["exp_avg_sq"].numpy(), rtol=1e-15, atol=1e-18)
    with pytest.raises(RuntimeError):
        S.adam_step(mine, gs, m, v, 1e-2, 0.9, 0.999, 1e-8, 0.0, 0)      # step counts from 1


def test_opening_steps_rounded_to_equal_steps():
    """ff_ode.walker_h_equal (ABI 108): the flow and adjoint passes round the step a warm-started walker opens with -- walker_h_init x
    walker_h_scale -- DOWN to t_span / k.  Equal, bit for bit, to passing the rounded steps themselves: narrow adjoint kernel (in its
    walker prologue), one-walker-per-workgroup adjoint kernels (ff_open_steps_kernel in front of the launch; per-walker and uniform
    entry), flow kernel."""
    rng = np.random.default_rng(11)
    He = Hm = 8
    eta = [rng.normal(size=He) * 0.5, rng.normal(size=He) * 0.3, rng.normal(size=He) * 0.05]
    mu = [rng.normal(size=Hm) * 0.5, rng.normal(size=Hm) * 0.3, rng.normal(size=Hm) * 0.05]
    net = S.Net(eta, mu, table=True)
    B = 6
    z = rng.normal(size=(B, 3, 2)) * 0.8
    az, ad = rng.normal(size=z.shape), rng.normal(size=B)
    h = np.array([0.38, 0.46, 0.55, 0.31, 1.4, 0.26])
    want = np.where(h * 1.1 < 1.0, 1.0 / np.ceil(1.0 / (h * 1.1) - 1e-9), h * 1.1)      # 1/3 1/2 1/2 1/3 1.54 1/4
    assert np.allclose(want, [1 / 3, 0.5, 0.5, 1 / 3, 1.54, 0.25])

    def adjoint(family, **w):
        prev = S.lib().ff_set_kernel_family(family)
        S.warm(**w)
        try:
            return S.cnf_adjoint(z, az, ad, net, rtol=1e-7, atol=1e-9)
        finally:
            S.warm()
            S.lib().ff_set_kernel_family(prev)
    for family in (0, 1):
        a = adjoint(family, h_init=h, h_scale=1.1, h_equal=True)
        b = adjoint(family, h_init=want, h_scale=1.0)
        c = adjoint(family, h_init=h, h_scale=1.1)
        assert a[2][3] == 0 and (a[0] == b[0]).all() and (a[2] == b[2]).all(), family
        # (the parameter gradient: the two waves of a narrow workgroup deposit into one table in whatever order the host simulator's
        # threads arrive -- an ulp between two identical calls)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(a[1], c[1], rtol=1e-5, atol=1e-8)           # the same gradient to the solver's tolerance
        # uniform entry (ff_ode.walker_h_uniform): one step for every walker
        u = adjoint(family, h_init=np.array([0.46]), h_scale=1.1, h_equal=True, uniform=True)
        v = adjoint(family, h_init=np.array([0.5]), h_scale=1.0, uniform=True)
        assert (u[0] == v[0]).all() and (u[2] == v[2]).all(), family
        np.testing.assert_allclose(u[1], v[1], rtol=1e-13, atol=1e-15)
    # flow pass
    S.warm(h_init=np.array([0.46]), h_scale=1.1, h_equal=True, uniform=True)
    try:
        xa, sa = S.cnf_generate(z, net)
    finally:
        S.warm()
    S.warm(h_init=np.array([0.5]), h_scale=1.0, uniform=True)
    try:
        xb, sb = S.cnf_generate(z, net)
    finally:
        S.warm()
    assert (xa == xb).all() and (sa == sb).all()


def test_one_launch_estimator_and_schedule_with_mean():
    """Round-4 launch diet.  ff_energy_estimate: the four estimator sums from many workgroups, joined in segment order by the
    workgroup that finishes last, and -- single rank -- E, the centred sum of squares and the surrogate in the same launch: equal
    to ff_reduce_energy + ff_energy_finish; the workspace counter is back at zero after every call (ragged and tiny batches too).
    ff_walker_order_mean: the cost-ordered schedule plus the mean of a per-walker array from the same two launches."""
    rng = np.random.default_rng(3)
    for B in (1, 5, 1024, 1025, 5000):
        e = rng.normal(size=B) * 7 + 30; lp = rng.normal(size=B) * 3 - 20
        for shift in (0.0, 29.5, float("nan")):
            want = S.reduce_energy(e, lp, shift)
            sums, est, ws = S.energy_estimate(e, lp, shift, B)
            np.testing.assert_allclose(sums, want, rtol=1e-12, atol=1e-9)
            np.testing.assert_allclose(est, S.energy_finish(want, shift, B), rtol=1e-12, atol=1e-9)
            assert ws.view(np.uint32)[0] == 0
            sums2, est2, _ = S.energy_estimate(e, lp, shift, 0, ws=ws)            # second call on the same workspace, sums only
            assert (sums2 == sums).all() and est2 is None
        assert abs(est[0] - e.mean()) < 1e-12 * abs(e.mean()) and abs(est[1] - ((e - e.mean()) ** 2).sum()) < 1e-9 * B
    for B in (7, 2048, 4100):
        cost = rng.integers(0, 40, size=B).astype(np.int32); h = rng.random(B)
        order, hm = S.walker_order(cost, hval=h)
        assert (order == S.walker_order(cost)).all() and abs(hm - h.mean()) < 1e-14
        # ff_walker_schedule: the same order and mean, plus the first step of every walker from the factor table of its cost class --
        # and the table follows the previous pass: classes of which > 10 % rejected their first step (he < hs) shrink by 0.93; classes
        # with < 5 % grow by 1.02 if 70 % of their voters (without an interval: every walker) accepted a step of the plan one shorter
        # (without an interval: 1.25 x the opening step); within [0.25, 1]; classes with fewer than 64 walkers and walkers without a
        # step keep theirs
        def rule(tab, cls, hs, he, interval, shrink_at=0.10):
            want = tab.copy()
            ok = (he > 0) & (hs > 0)
            if interval > 0:
                k = np.rint(interval / np.where(hs > 0, hs, 1.0))
                vote = ok & (k >= 3)
                yes = vote & (he >= 0.999 * interval / np.maximum(k - 1, 1))
            else:
                vote, yes = ok, ok & (he >= 1.25 * hs)
            for c in range(32):
                m = cls == c
                n_c, r_c, v_c, y_c = int((m & ok).sum()), int((m & ok & (he < 0.999 * hs)).sum()), int((m & vote).sum()), int((m & yes).sum())
                if n_c >= 64:
                    f = 0.93 if r_c / n_c > shrink_at else (1.02 if (r_c / n_c < 0.5 * shrink_at and v_c >= 16 and y_c >= 0.7 * v_c) else 1.0)
                    want[c] = min(1.0, max(0.25, tab[c] * f))
            return want
        tab = np.where(np.arange(32) <= 6, 0.9, 0.6)
        o2, hm2, hs, tab1 = S.walker_schedule(cost, h, tab)
        assert (o2 == order).all() and hm2 == hm and (tab1 == tab).all()
        np.testing.assert_array_equal(hs, h * tab[np.minimum(cost, 31)])
        he = hs.copy()
        cls = np.minimum(cost, 31)
        rej = (cls == 3) | ((cls == 5) & (rng.random(B) < 0.075)) | (cls == 31)      # (class 5: between the two thresholds, give or take)
        he[rej] *= 0.5
        he[(cls <= 1) | (cls == 6)] *= 1.3                   # these accepted a step well beyond the one they opened with: room to grow
        he[(cls == 9) & (rng.random(B) < 0.5)] *= 1.3        # only half of this class did: it stays (and so do 2, 4, 8: no rejections, no evidence)
        he[cls == 7] = 0.0                                   # a class whose walkers report no accepted step: no evidence
        _, _, hs2, tab2 = S.walker_schedule(cost, h, tab1, prev=(cost, hs, he))
        want = rule(tab, cls, hs, he, 0.0)
        np.testing.assert_allclose(tab2, want, rtol=1e-15)
        np.testing.assert_array_equal(hs2, h * tab2[cls])    # the update is applied at once: this pass opens with what the last one taught
        # the same update from the statistics counted separately (ff_scale_counts: what a data-parallel run all-reduces)
        cnts = S.scale_counts(cost, hs, he)
        assert cnts[:32].sum() == ((he > 0) & (hs > 0)).sum() and cnts[32:64].sum() == ((he > 0) & (he < 0.999 * hs)).sum()
        assert (cnts[64:96] == cnts[:32]).all() and cnts[96:].sum() == ((he > 0) & (he >= 1.25 * hs)).sum()
        _, _, hs2c, tab2c = S.walker_schedule(cost, h, tab1, counts=cnts)
        assert (tab2c == tab2).all() and (hs2c == hs2).all()
        # interval > 0: steps rounded down to interval / k
        o3, _, hs3, _ = S.walker_schedule(cost, h, tab, interval=1.0)
        hq = h * tab[cls]
        # ... and the work is ordered by class + 4 x (planned equal steps beyond two): walkers that will take a third step sit together
        kk = np.where(hq >= 1.0, 1, np.ceil(1.0 / hq - 1e-9)).astype(int)
        key = np.minimum(31, cls + 4 * np.maximum(0, np.minimum(kk, 8) - 2))
        assert sorted(o3.tolist()) == list(range(B)) and (np.diff(key[o3]) <= 0).all()
        np.testing.assert_allclose(hs3, np.where(hq < 1.0, 1.0 / np.ceil(1.0 / hq - 1e-9), hq), rtol=1e-15)
        assert (hs3 <= hq * (1 + 1e-12)).all()
        # with an interval the voters are the walkers planned for k >= 3 equal steps, and the evidence a step >= interval / (k - 1)
        k3 = np.rint(1.0 / hs3)
        he3 = hs3.copy()
        he3[(cls == 12) | (cls == 20)] = (1.0 / np.maximum(k3 - 1, 1))[(cls == 12) | (cls == 20)]      # (k = 2 walkers among them report 1.0: no vote)
        he3[cls == 15] *= 1.2                                # larger steps, but short of the shorter plan's (k <= 5: 1.25 x at least)
        he3[(cls == 15) & (k3 > 5)] = hs3[(cls == 15) & (k3 > 5)]
        _, _, hs4, tab4 = S.walker_schedule(cost, h, tab, prev=(cost, hs3, he3), interval=1.0)
        want4 = rule(tab, cls, hs3, he3, 1.0)
        np.testing.assert_allclose(tab4, want4, rtol=1e-15)
        # the caller's threshold (ff_walker_schedule's shrink_at; 0 = 0.10): 0.25, as for kernels that integrate one walker per wave
        _, _, _, tab5 = S.walker_schedule(cost, h, tab1, prev=(cost, hs, he), shrink_at=0.25)
        np.testing.assert_allclose(tab5, rule(tab, cls, hs, he, 0.0, 0.25), rtol=1e-15)
        cnts3 = S.scale_counts(cost, hs3, he3, interval=1.0)
        assert cnts3[64:96].sum() == ((k3 >= 3) & (hs3 > 0)).sum()
        _, _, hs4c, tab4c = S.walker_schedule(cost, h, tab, counts=cnts3, interval=1.0)
        assert (tab4c == tab4).all() and (hs4c == hs4).all()
        if B >= 4100:      # (~100 walkers per class)
            assert tab2[3] == tab[3] * 0.93 and tab2[0] == min(1.0, tab[0] * 1.02) and tab2[6] == min(1.0, tab[6] * 1.02)
            assert tab2[7] == tab[7] and tab2[9] == tab[9] and tab2[2] == tab[2]
            assert tab4[12] == tab[12] * 1.02 and tab4[20] == tab[20] * 1.02 and tab4[15] == tab[15] and tab4[3] == tab[3]
