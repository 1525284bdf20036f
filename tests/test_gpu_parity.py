"""Parity tests proper: the hipcc-built kernels on a real MI355X, called through the C ABI (fermiflow_amd.native /
the reference-named classes), against the golden vectors from the reference and against the oracle.

Bars (BASELINE.json north_star): MCMC acceptance indices bit-exact; E_loc within 1e-5 relative in fp64."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.common import mcmc_noise_from_seed, net_arrays, cnf_param_grads, gsvmc_param_grads, GSVMC_PG

pytestmark = pytest.mark.gpu
ELOC_RTOL = 1e-5     # the bar of the north star; observed ~1e-9


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the -m gpu tests need an MI355X"
    return torch.device("cuda:0")


def T(a, dev, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def _oracle_net(model):
    """the oracle's view of a model's flow (the same weights as numpy arrays)"""
    v = model.cnf.v_wrapper.v
    return O.Net(tuple(N(t) for t in (v.eta.fc1.weight, v.eta.fc1.bias, v.eta.fc2.weight)),
                 tuple(N(t) for t in (v.mu.fc1.weight, v.mu.fc1.bias, v.mu.fc2.weight)))


def make_mlp(w, dev):
    import fermiflow_amd as ff
    m = ff.MLP(1, len(w[1]))
    with torch.no_grad():
        m.fc1.weight.copy_(torch.as_tensor(w[0]).reshape(-1, 1))
        m.fc1.bias.copy_(torch.as_tensor(w[1]))
        m.fc2.weight.copy_(torch.as_tensor(w[2]).reshape(1, -1))
    return m.to(dev)


def make_flow(eta, mu, dev):
    import fermiflow_amd as ff
    v = ff.Backflow(make_mlp(eta, dev), mu=make_mlp(mu, dev) if mu is not None else None)
    return ff.CNF(v, (0.0, 1.0))


# ------------------------------------------------------------------------------------------------ Slater
def test_slater_fwd_bwd_laplacian(golden, dev):
    import fermiflow_amd as ff
    from fermiflow_amd import native
    G = golden["g2_slater"]
    h = ff.HO2D()
    for n in (1, 3, 5, 6, 10):
        orbs = tuple(h.orbitals[k] for k in G[f"n{n}_orb"])
        x = T(G[f"n{n}_x"], dev).requires_grad_(True)
        y = ff.LogAbsSlaterDet.apply(orbs, x)
        y.sum().backward()
        np.testing.assert_allclose(N(y), G[f"n{n}_logabsdet"], atol=1e-12)
        np.testing.assert_allclose(N(x.grad), G[f"n{n}_grad"], rtol=1e-9, atol=1e-9)
        tab = native.orbital_table(G[f"n{n}_orb"], dev)
        lp, g, lap = native.logprob(tab, None, n, 0, x.detach(), derivs=True)
        np.testing.assert_allclose(N(lap) / 2, G[f"n{n}_lap"], rtol=1e-9, atol=1e-7)
    # several batch dims (src/slater.py:28) and both spins
    bd = ff.FreeFermion(device=dev)
    up = tuple(h.orbitals[k] for k in G["lp_up"]); dn = tuple(h.orbitals[k] for k in G["lp_dn"])
    x = T(G["lp_x"], dev)
    lp = bd.log_prob(up, dn, x.reshape(4, 5, 9, 2))
    np.testing.assert_allclose(N(lp).reshape(-1), G["lp_logp"], atol=1e-11)
    _, g, lap = ff.y_grad_laplacian(ff.utils.freefermion_logp(bd, up, dn), x)
    np.testing.assert_allclose(N(g), G["lp_grad"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(N(lap), G["lp_lap"], rtol=1e-9, atol=1e-6)
    # multi-state primitive
    states, _ = h.fermion_states(3, 0, 2.0)
    coll = dict(zip(G["ms_keys"].tolist(), G["ms_counts"].tolist()))
    x = T(G["ms_x"], dev).requires_grad_(True)
    y = bd.log_prob_multstates(states, coll, x)
    y.sum().backward()
    np.testing.assert_allclose(N(y), G["ms_logp"], atol=1e-12)
    np.testing.assert_allclose(N(x.grad), G["ms_grad"], rtol=1e-9, atol=1e-10)


def test_known_answer_eigenfunctions_full_size(dev):
    """reference tests/test_basedist.py:5-60 at BASELINE size: E_loc == sum of orbital energies for all
    65536 random points (size-independent property)."""
    from fermiflow_amd import native
    torch.manual_seed(0)
    x = torch.randn(65536, 9, 2, dtype=torch.float64, device=dev)
    iu, idn = [0, 2, 5], [0, 1, 3, 4, 7, 9]
    Es = [n + 1 for n in range(8) for _ in range(n + 1)]
    lp, g, lap = native.logprob(native.orbital_table(iu, dev), native.orbital_table(idn, dev), 3, 6, x, derivs=True)
    eloc = -0.25 * lap - 0.125 * (g ** 2).sum(dim=(1, 2)) + 0.5 * (x ** 2).sum(dim=(1, 2))
    want = sum(Es[k] for k in iu) + sum(Es[k] for k in idn)
    assert (eloc - want).abs().max().item() < 1e-6 * want


# ------------------------------------------------------------------------------------------------ MCMC
@pytest.mark.parametrize("nup", [2, 3, 4, 5, 6])
def test_particle_split_metropolis_kernel_vs_oracle(dev, nup):
    """One spin species (the finite-temperature runs): ff_mcmc_pair_kernel, two lanes per walker splitting the particles.
    Noise-fed chain == the oracle bit for bit (walkers, accept masks), also with one orbital set per walker; the Philox path
    == the noise path on the materialised stream; 4097 walkers: the last workgroup is ragged."""
    from fermiflow_amd import native
    rng = np.random.default_rng(100 + nup)
    B, steps = 4097, 25
    g0 = rng.normal(size=(B, nup, 2)); g = rng.normal(size=(steps, B, nup, 2)); u = rng.random((steps, B))
    tab1 = native.orbital_table(list(range(nup)), dev)
    x, lp, acc = native.mcmc_sample_noise(tab1, None, nup, 0, T(g0, dev), T(g, dev), T(u, dev), 0.1)
    xo, lpo, acco = O.mcmc_noise(g0, g, u, nup, 0)
    assert (N(x) == xo).all() and (N(acc) == acco).all()
    np.testing.assert_allclose(N(lp), lpo, rtol=1e-12, atol=1e-12)
    sets = np.stack([np.sort(rng.choice(15, size=nup, replace=False)) for _ in range(5)]).astype(np.int32)
    ws = np.sort(rng.integers(0, 5, size=B)).astype(np.int32)
    tabs = native.orbital_table([list(map(int, r)) for r in sets], dev)
    x, lp, acc = native.mcmc_sample_noise(tabs, None, nup, 0, T(g0, dev), T(g, dev), T(u, dev), 0.1,
                                          walker_state=torch.as_tensor(ws, device=dev))
    xo, lpo, acco = O.mcmc_noise(g0, g, u, nup, 0, tab_up=sets, wstate=ws)
    assert (N(x) == xo).all() and (N(acc) == acco).all()
    h0, h, hu = native.rng_fill(B, nup, steps, 4242, dev, walker_offset=7)
    x1, _, a1 = native.mcmc_sample_noise(tab1, None, nup, 0, h0, h, hu, 0.1)
    x2, _, cnt = native.mcmc_sample(tab1, None, nup, 0, B, steps, 0.1, 4242, dev, walker_offset=7)
    assert torch.equal(x1, x2) and torch.equal(a1.sum(0).to(torch.int32), cnt.to(torch.int32))


@pytest.mark.parametrize("name", ["u3d3", "u6d0", "u6d6", "u1d0", "u10d0", "u3d3_b512"])
def test_mcmc_bit_exact_vs_reference(golden, dev, name):
    import fermiflow_amd as ff
    G = golden["g1_mcmc"]
    nup, ndn, g0, g, u, accept = mcmc_noise_from_seed(G, name)
    h = ff.HO2D()
    bd = ff.FreeFermion(device=dev)
    x, logp, acc = bd.sample_with_noise(h.orbitals[:nup], h.orbitals[:ndn], T(g0, dev), T(g, dev), T(u, dev))
    assert (N(acc) == accept).all(), "acceptance indices differ from the reference"
    assert (N(x) == G[name + "_x"]).all(), "final walkers are not bit-identical"
    np.testing.assert_allclose(N(logp), G[name + "_logp"], atol=1e-12)
    if name == "u3d3_b512":     # BASELINE.md 2 anchor: seed 7, 512 walkers
        import hashlib
        assert hashlib.sha256(np.ascontiguousarray(N(x)).tobytes()).hexdigest().startswith("d7799a21de62365d")
        assert abs(N(acc).mean() - 0.7527) < 5e-5


def test_mcmc_full_size_properties(dev):
    """65536 walkers x 100 steps: fused Philox kernel == noise-fed kernel bit for bit; oracle agrees on the
    same noise (sampled rows); sharding does not change any walker; acceptance rate in the reference's range."""
    from fermiflow_amd import native
    B, S = 65536, 100
    tu = native.orbital_table([0, 1, 2], dev)
    x, logp, cnt = native.mcmc_sample(tu, tu, 3, 3, B, S, 0.1, 99, dev)
    g0, g, u = native.rng_fill(B, 6, S, 99, dev)
    x2, logp2, acc = native.mcmc_sample_noise(tu, tu, 3, 3, g0, g, u)
    assert torch.equal(x, x2) and torch.equal(acc.sum(0).to(torch.int32), cnt)
    rate = cnt.double().mean().item() / S
    assert abs(rate - 0.7527) < 0.002, rate          # reference: 0.7527 for (3,3) (tests/golden g1_mcmc u3d3_b512)
    rows = slice(1000, 1064)
    xo, lo, ao = O.mcmc_noise(N(g0[rows]), N(g[:, rows]), N(u[:, rows]), 3, 3)
    assert (ao == N(acc[:, rows])).all() and (xo == N(x[rows])).all()
    xa, _, _ = native.mcmc_sample(tu, tu, 3, 3, B // 2, S, 0.1, 99, dev, walker_offset=0)
    xb, _, _ = native.mcmc_sample(tu, tu, 3, 3, B // 2, S, 0.1, 99, dev, walker_offset=B // 2)
    assert torch.equal(torch.cat([xa, xb]), x)
    assert abs(g.mean().item()) < 1e-3 and abs(g.std().item() - 1) < 1e-3 and abs(u.mean().item() - 0.5) < 1e-3


@pytest.mark.parametrize("nup,ndn", [(4, 3), (7, 6), (2, 5), (0, 7)])
def test_sixteen_lane_sampler_philox_equals_noise_path(dev, nup, ndn):
    """ff_mcmc_rows_kernel (csrc/ff_ho3d.hip: every shape outside the register-resident template list, and d = 3): its Philox-fed
    chain -- Philox blocks dealt over the walker's 32 lanes, determinant-ratio accept test on the polynomial parts -- gives the
    walkers of its noise-fed chain (the reference's arithmetic, == the oracle bit for bit) on the materialised stream."""
    from fermiflow_amd import native
    B, S, n = 1001, 30, nup + ndn
    tu = native.orbital_table(list(range(nup)), dev) if nup else None
    td = native.orbital_table(list(range(ndn)), dev) if ndn else None
    g0, g, u = native.rng_fill(B, n, S, 321, dev, walker_offset=11)
    x1, lp1, a1 = native.mcmc_sample_noise(tu, td, nup, ndn, g0, g, u)
    x2, lp2, cnt = native.mcmc_sample(tu, td, nup, ndn, B, S, 0.1, 321, dev, walker_offset=11)
    assert torch.equal(x1, x2) and torch.equal(a1.sum(0).to(torch.int32), cnt.to(torch.int32))
    assert torch.equal(lp1, lp2)
    rows = slice(100, 116)
    xo, lo, ao = O.mcmc_noise(N(g0[rows]), N(g[:, rows]), N(u[:, rows]), nup, ndn)
    assert (ao == N(a1[:, rows])).all() and (xo == N(x2[rows])).all()


def test_philox_stream_is_normal_and_symmetric(dev):
    """csrc/ff_rng.h: the proposal normals are Box-Muller on 32-bit Philox words evaluated with the hardware fp32
    transcendentals (v_log_f32, v_sqrt_f32, v_sin_f32, v_cos_f32) and promoted to fp64; each normal takes its sign from a bit of
    its own, so the proposal is symmetric exactly -- all a Metropolis chain needs to be exact.  Kolmogorov-Smirnov tests of the
    materialised stream (2 M normals: marginal, and the pair radius r^2 / 2 ~ Exp(1)), moments to the 4th, the tail, sign
    balance, lag-1 independence along a chain, and the uniforms."""
    from scipy import stats
    from fermiflow_amd import native
    B, S = 16384, 10
    g0, g, u = native.rng_fill(B, 6, S, 2024, dev)
    z = N(g).reshape(-1)
    assert np.isfinite(z).all() and np.abs(z).max() < 6.8
    assert abs(z.mean()) < 3e-3 and abs(z.std() - 1) < 2e-3
    assert abs((z ** 3).mean()) < 8e-3 and abs((z ** 4).mean() - 3.0) < 3e-2
    sub = z[::2][:400000]                                   # KS at 4e5 samples resolves 2e-3 in the CDF
    assert stats.kstest(sub, "norm").pvalue > 1e-3
    pairs = z.reshape(-1, 2)
    assert stats.kstest(0.5 * (pairs ** 2).sum(1)[:400000], "expon").pvalue > 1e-3
    ang = np.arctan2(pairs[:, 1], pairs[:, 0])[:400000]
    assert stats.kstest((ang + np.pi) / (2 * np.pi), "uniform").pvalue > 1e-3
    assert abs((z > 0).mean() - 0.5) < 1.5e-3 and abs((pairs[:, 0] * pairs[:, 1] > 0).mean() - 0.5) < 2e-3
    assert abs(np.mean(np.abs(z) > 3.0) - 2.6998e-3) < 2e-4
    gg = N(g)                                               # (S, B, 6, 2): consecutive steps of one coordinate
    assert abs(np.corrcoef(gg[:-1].reshape(-1), gg[1:].reshape(-1))[0, 1]) < 3e-3
    assert stats.kstest(N(u).reshape(-1), "uniform").pvalue > 1e-3
    assert stats.kstest(N(g0).reshape(-1), "norm").pvalue > 1e-3


# ------------------------------------------------------------------------------------------------ backflow
def test_backflow_mlp_potentials(golden, dev):
    import fermiflow_amd as ff
    G = golden["g3_backflow"]
    for k in range(int(G["ncase"])):
        n, d, He, Hm = G[f"c{k}_cfg"]
        eta, mu = net_arrays(G, f"c{k}_", Hm > 0)
        v = ff.Backflow(make_mlp(eta, dev), mu=make_mlp(mu, dev) if mu is not None else None)
        x = T(G[f"c{k}_x"], dev)
        np.testing.assert_allclose(N(v(x)), G[f"c{k}_v"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(N(v.divergence(x)), G[f"c{k}_div"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(N(ff.HO().V(x)), G[f"c{k}_Vho"], rtol=1e-13)
        np.testing.assert_allclose(N(ff.CoulombPairPotential(2.0).V(x)), G[f"c{k}_Vc"], rtol=1e-12)
    m = make_mlp(net_arrays(G, "c0_")[0], dev)
    r = T(G["mlp_r"], dev)[:, None]
    np.testing.assert_allclose(N(m(r)).reshape(-1), G["mlp_eta"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(N(m.grad(r)).reshape(-1), G["mlp_deta"], rtol=1e-12, atol=1e-15)
    # permutation equivariance (reference tests/test_equivariant_funs.py:20-23), n = 10, d = 3
    eta, mu = net_arrays(G, "c3_")
    v = ff.Backflow(make_mlp(eta, dev), mu=make_mlp(mu, dev))
    x = torch.randn(1000, 10, 3, dtype=torch.float64, device=dev)
    P = torch.randperm(10, device=dev)
    assert torch.allclose(v(x[:, P, :]), v(x)[:, P, :], rtol=1e-10, atol=1e-12)


# ------------------------------------------------------------------------------------------------ CNF
@pytest.mark.parametrize("tag,rt,at,tol", [("tol6", 1e-6, 1e-8, 2e-6), ("tol10", 1e-10, 1e-12, 5e-10)])
def test_cnf_generate_logp_adjoint(golden, dev, tag, rt, at, tol):
    G = golden["g4_cnf"]
    eta, mu = net_arrays(G, "")
    cnf = make_flow(eta, mu, dev)
    cnf.rtol, cnf.atol = rt, at
    x = cnf.generate(T(G[tag + "_z"], dev))
    np.testing.assert_allclose(N(x), G[tag + "_x"], atol=tol)
    xg = T(G[tag + "_x"], dev).requires_grad_(True)
    z, dl = cnf.delta_logp(xg, params_require_grad=True)
    np.testing.assert_allclose(N(z), G[tag + "_zback"], atol=tol)
    np.testing.assert_allclose(N(dl), G[tag + "_dlogp"], atol=tol)
    loss = (T(G[tag + "_cz"], dev) * z).sum() + (T(G[tag + "_cd"], dev) * dl).sum()
    grads = torch.autograd.grad(loss, [xg] + list(cnf.parameters()))
    np.testing.assert_allclose(N(grads[0]), G[tag + "_gx"], atol=10 * tol)
    ref = cnf_param_grads(G, tag)
    got = np.concatenate([N(g).reshape(-1) for g in grads[1:]])
    np.testing.assert_allclose(got, ref, atol=10 * tol * np.abs(ref).max())


def test_cnf_reversibility_full_size(golden, dev):
    """generate followed by delta_logp returns to z (src/flow.py:57-69 check_reversibility) on 65536 walkers."""
    G = golden["g4_cnf"]
    cnf = make_flow(*net_arrays(G, ""), dev)
    torch.manual_seed(1)
    z = torch.randn(65536, 6, 2, dtype=torch.float64, device=dev)
    x = cnf.generate(z)
    zb, dl = cnf.delta_logp(x)
    assert (zb - z).abs().max().item() < 2e-5
    assert torch.isfinite(dl).all()
    # a chunk against the oracle (its own batch-global step control): tolerance-level agreement
    net = O.Net(*net_arrays(G, ""))
    xo, _ = O.cnf_generate(N(z[:64]), net)
    np.testing.assert_allclose(N(x[:64]), xo, atol=5e-6)


# ------------------------------------------------------------------------------------------------ E_loc
@pytest.mark.parametrize("name", ["z0_zero", "z2_zero", "z05_nt", "z2_nt", "u6_nt", "z2_nomu", "u6d6_nt"])
def test_local_energy_vs_reference(golden, dev, name):
    import fermiflow_amd as ff
    G = golden["g5_gsvmc"]
    nup, ndn, B, seed = (int(v) for v in G[name + "_cfg"])
    use_mu = bool(G[name + "_use_mu"])
    eta, mu = net_arrays(G, name + "_", use_mu)
    cnf = make_flow(eta, mu, dev)
    model = ff.GSVMC(nup, ndn, ff.HO2D(), ff.FreeFermion(device=dev), cnf,
                     ff.CoulombPairPotential(float(G[name + "_Z"])), sp_potential=ff.HO())
    x = T(G[name + "_x"], dev)
    r = model.local_energy(x, want_stats=True)                     # reference default tolerances
    assert int(r["stats"][3]) == 0
    el = G[name + "_Eloc"]
    assert (np.abs(N(r["eloc"]) - el) / np.abs(el)).max() < ELOC_RTOL
    np.testing.assert_allclose(N(r["eloc"]), el, rtol=1e-7)         # what is actually achieved
    np.testing.assert_allclose(N(r["logp"]), G[name + "_logp"], atol=1e-6)
    np.testing.assert_allclose(N(r["grad"]), G[name + "_grad"], atol=1e-6)
    np.testing.assert_allclose(N(r["lap"]), G[name + "_lap"], rtol=1e-7, atol=1e-5)
    np.testing.assert_allclose(N(r["V"]), G[name + "_V"], rtol=1e-12)
    logp, grad, lap = ff.y_grad_laplacian(model.logp, x)            # the reference's call, src/VMC.py:48
    assert torch.equal(lap, r["lap"])
    # theta-gradient of the surrogate from these walkers (src/VMC.py:58, FermionHO2D.py:71)
    from fermiflow_amd import native
    w = (r["eloc"] - r["eloc"].mean()) / B
    _, gp = native.cnf_adjoint(cnf.v_wrapper.v.net(), r["z"], w[:, None, None] * r["glogp0"], -w, 0.0, 1.0, 1e-6, 1e-8,
                               need_gx=False)
    ref = gsvmc_param_grads(G, name, use_mu)
    np.testing.assert_allclose(N(gp), ref, atol=1e-5 * max(np.abs(ref).max(), 1e-8))   # zero-flow case: gradient is rounding noise
    np.testing.assert_allclose(((r["logp"] * w).sum()).item(), float(G[name + "_gradE"]), rtol=1e-5, atol=1e-12)


def test_local_energy_full_size_known_answer(dev):
    """65536 walkers, zero-initialised flow (the driver default, src/FermionHO2D.py:40-43), Z = 0: every
    walker's E_loc is exactly the free-fermion energy 10 (nup = ndown = 3)  -> E = 10, E_std ~ 1e-14."""
    import __graft_entry__ as Gm
    model = Gm._model(dev, 3, 3, 0.0)
    for p in model.parameters():
        torch.nn.init.zeros_(p)
    torch.manual_seed(3)
    g = model(65536)
    g.backward()
    assert abs(model.E - 10.0) < 1e-9 and model.E_std < 1e-9
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


def test_gsvmc_iteration_statistics_and_oracle(golden, dev):
    """A full sweep from fresh walkers: per-walker E_loc agrees with the oracle on the very walkers the GPU
    produced; E is consistent with the reference's value for this configuration (30.43 +- 4.29/sqrt(B))."""
    import __graft_entry__ as Gm
    G = golden["g5_gsvmc"]
    model = Gm._model(dev, 3, 3, 2.0)
    torch.manual_seed(5)
    B = 8192
    gE = model(B)
    gE.backward()
    net = O.Net(*net_arrays(G, "z2_nt_"))
    idx = slice(0, 128)
    ref = O.eloc(N(model.x[idx]), 3, 3, net, 2.0, rtol=1e-10, atol=1e-12)
    rel = np.abs(N(model.Eloc[idx]) - ref["eloc"]) / np.abs(ref["eloc"])
    assert rel.max() < ELOC_RTOL, rel.max()
    assert abs(model.E - float(G["z2_nt_E"])) < 6 * 4.5 / np.sqrt(32)     # golden E is itself a 32-walker estimate
    assert 3.0 < model.E_std < 12.0       # heavy-tailed (Coulomb): 4.9 .. 7.4 seen over seeds / noise streams at this B
    np.testing.assert_allclose(model.E, model.Eloc.mean().item(), rtol=1e-13)
    np.testing.assert_allclose(model.E_std, model.Eloc.std().item(), rtol=1e-10)


@pytest.mark.parametrize("name,rt,at,vtol,gtol", [("z2_nt", 1e-10, 1e-12, 1e-7, 1e-6), ("z2_nt", 1e-6, 1e-8, 1e-5, 1e-5),
                                                  ("z05_nt", 1e-10, 1e-12, 1e-7, 1e-6), ("u6_nt", 1e-10, 1e-12, 1e-7, 1e-6),
                                                  ("z2_nomu", 1e-10, 1e-12, 1e-7, 1e-6), ("u6d6_nt", 1e-10, 1e-12, 1e-7, 1e-6),
                                                  ("u6d6_nt", 1e-6, 1e-8, 1e-5, 1e-5)])
def test_gsvmc_forward_backward_vs_reference(golden, dev, name, rt, at, vtol, gtol):
    """GSVMC.forward -> .backward() END TO END (src/VMC.py:40-59, src/FermionHO2D.py:69-72) on the reference's base
    walkers z, through the production sweep (flow with cost classes, cost-ordered local-energy pass, step-size warm
    start between the three integrations, one-pass moments, _ScalarWithParamGrads): E, E_std, gradE and the .grad of
    every parameter against the reference's.  Tolerances: values 1e-7 relative at rtol 1e-10 (1e-5, the north-star
    bar, at the reference's default tolerance), gradients 1e-6 (1e-5) of their largest entry."""
    import fermiflow_amd as ff
    G = golden["g5_gsvmc"]
    nup, ndn, B, seed = (int(v) for v in G[name + "_cfg"])
    use_mu = bool(G[name + "_use_mu"])
    eta, mu = net_arrays(G, name + "_", use_mu)
    cnf = make_flow(eta, mu, dev)
    cnf.rtol, cnf.atol = rt, at
    model = ff.GSVMC(nup, ndn, ff.HO2D(), ff.FreeFermion(device=dev), cnf,
                     ff.CoulombPairPotential(float(G[name + "_Z"])), sp_potential=ff.HO())
    assert model.warm_start and model.sens_tol == (1.0 if nup + ndn <= 6 else 10.0) and model.adaptive_h      # the production settings: warm start, one tolerance (a factor beyond 12 coordinates), learned first steps
    for sweep in range(2):      # the second sweep opens its flow pass with the first one's mean step (warm start across sweeps)
        gradE = model.forward_from(T(G[name + "_z"], dev))
        model.zero_grad()
        gradE.backward()
        np.testing.assert_allclose(N(model.x), G[name + "_x"], atol=100 * rt)
        np.testing.assert_allclose(model.E, float(G[name + "_E"]), rtol=vtol)
        np.testing.assert_allclose(model.E_std, float(G[name + "_E_std"]), rtol=10 * vtol)
        np.testing.assert_allclose(gradE.item(), float(G[name + "_gradE"]), rtol=100 * vtol, atol=1e-9)
        ref = gsvmc_param_grads(G, name, use_mu)
        got = np.concatenate([N(p.grad).reshape(-1) for p in model.parameters()])
        np.testing.assert_allclose(got, ref, atol=gtol * np.abs(ref).max())


@pytest.mark.parametrize("tag,rt,at,vtol", [("boltz", 1e-10, 1e-12, 1e-7), ("hot", 1e-10, 1e-12, 1e-7), ("rand", 1e-10, 1e-12, 1e-7),
                                            ("boltz", 1e-6, 1e-8, 1e-5)])
def test_betavmc_forward_backward_vs_reference(golden, dev, tag, rt, at, vtol):
    """BetaVMC.forward -> both .backward()s END TO END (src/VMC.py:114-171, src/BetaFermionHO2D.py:72-79) on the
    reference's base walkers and state assignment: E, E_std, F, F_std, S, S_analytical, logp_states_all, gradF_phi,
    gradF_theta, d/d(log_state_weights) and the six flow-parameter gradients.  Values 1e-7 relative, gradients 1e-5 of
    their largest entry."""
    import fermiflow_amd as ff
    G = golden["g6_betavmc"]
    eta, mu = net_arrays(G, "")
    cnf = make_flow(eta, mu, dev)
    cnf.rtol, cnf.atol = rt, at      # (1e-6 / 1e-8: the reference's default, with the sweep's tolerance policy and warm start on)
    model = ff.BetaVMC(float(G[tag + "_beta"]), 3, 0, float(G[tag + "_dE"]), tag != "rand", ff.HO2D(), ff.FreeFermion(device=dev), cnf,
                       ff.CoulombPairPotential(2.0), sp_potential=ff.HO())
    model.to(dev)
    with torch.no_grad():
        model.log_state_weights.copy_(T(G[tag + "_logits"], dev))
    ws = np.repeat(G[tag + "_keys"], G[tag + "_counts"])
    gphi, gtheta = model.forward_from(T(G[tag + "_z"], dev), ws)
    model.zero_grad()
    (gphi + gtheta).backward()
    np.testing.assert_allclose(N(model.x), G[tag + "_x"], atol=max(1e-8, 100 * rt))
    for k in ("E", "F", "S", "S_analytical"):
        np.testing.assert_allclose(getattr(model, k), float(G[f"{tag}_{k}"]), rtol=vtol, err_msg=k)
    for k in ("E_std", "F_std"):
        np.testing.assert_allclose(getattr(model, k), float(G[f"{tag}_{k}"]), rtol=10 * vtol, err_msg=k)
    np.testing.assert_allclose(N(model.logp_states_all), G[tag + "_logp_states_all"], atol=1e-12)
    np.testing.assert_allclose(gphi.item(), float(G[tag + "_gphi"]), rtol=max(1e-5, 100 * vtol), atol=1e-10)
    np.testing.assert_allclose(gtheta.item(), float(G[tag + "_gtheta"]), rtol=max(1e-5, 100 * vtol), atol=1e-10)
    ref = G[tag + "_pg_log_state_weights"]
    np.testing.assert_allclose(N(model.log_state_weights.grad), ref, atol=1e-5 * max(np.abs(ref).max(), 1e-12))
    ref = np.concatenate([G[f"{tag}_pg_{k}"] for k in GSVMC_PG])
    got = np.concatenate([N(p.grad).reshape(-1) for p in cnf.parameters()])
    np.testing.assert_allclose(got, ref, atol=1e-5 * np.abs(ref).max())
    assert dict(model.state_indices_collection) == dict(zip(G[tag + "_keys"].tolist(), G[tag + "_counts"].tolist()))


def test_betavmc_vs_reference(golden, dev):
    import fermiflow_amd as ff
    G = golden["g6_betavmc"]
    eta, mu = net_arrays(G, "")
    for tag in ("boltz", "hot"):
        cnf = make_flow(eta, mu, dev)
        model = ff.BetaVMC(float(G[tag + "_beta"]), 3, 0, float(G[tag + "_dE"]), True, ff.HO2D(), ff.FreeFermion(device=dev), cnf,
                           ff.CoulombPairPotential(2.0), sp_potential=ff.HO())
        model.to(dev)
        assert model.Nstates == 21
        assert (np.array([[o.k for o in s[0]] for s in model.states]) == G[tag + "_states"]).all()
        ws = torch.as_tensor(np.repeat(G[tag + "_keys"], G[tag + "_counts"]), dtype=torch.int32, device=dev)
        r = model.local_energy(T(G[tag + "_x"], dev), ws)
        el = G[tag + "_Eloc"]
        assert (np.abs(N(r["eloc"]) - el) / np.abs(el)).max() < ELOC_RTOL
        np.testing.assert_allclose(N(r["logp"]), G[tag + "_logp"], atol=1e-6)
        # a full native forward/backward runs and gives sane thermodynamics
        torch.manual_seed(0)
        gphi, gtheta = model(512)
        (gphi + gtheta).backward()
        np.testing.assert_allclose(model.S_analytical, float(G[tag + "_S_analytical"]), rtol=1e-12)
        np.testing.assert_allclose(N(model.logp_states_all), G[tag + "_logp_states_all"], atol=1e-12)
        assert np.isfinite([model.E, model.F, model.S]).all()
        assert model.log_state_weights.grad is not None and all(p.grad is not None for p in cnf.parameters())
        # gradient wrt the state logits against autograd on the same walkers (src/VMC.py:162)
        ws2 = model._walker_state(dev).to(torch.int64)
        lw = model.log_state_weights.detach().clone().requires_grad_(True)
        lps = torch.log_softmax(lw, dim=0)[ws2]
        Floc = model.Eloc + lps.detach() / model.beta
        (lps * (Floc - model.F)).mean().backward()
        gtheta_before = [p.grad.clone() for p in cnf.parameters()]
        model.zero_grad()
        gphi.backward()
        np.testing.assert_allclose(N(model.log_state_weights.grad), N(lw.grad), atol=1e-12 * max(1.0, lw.grad.abs().max().item()))
        assert all(p.grad is None or (p.grad == 0).all() for p in cnf.parameters())      # gradF_phi does not touch the flow


# ------------------------------------------------------------------------------------------------ radial tables
def test_radial_table_equals_direct_evaluation_and_is_deterministic(golden, dev):
    """The tabulated eta/mu path (default) against the direct sigmoid evaluation on 8192 fresh walkers: local
    energies, flow and parameter gradient agree far below the solver tolerance; the gradient is bit-reproducible."""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    model = Gm._model(dev, 3, 3, 2.0)
    v = model.cnf.v_wrapper.v
    tu, td = model._tables(dev)
    torch.manual_seed(11)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (8192,))
    res = {}
    for mode in ("table", "exact"):
        net = v.net(radial=mode)
        x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8)
        r = native.eloc(tu, td, 3, 3, net, x if mode == "table" else res["table"][0], 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True)
        w = (r["eloc"] - r["eloc"].mean()) / 8192
        gx, gp = native.cnf_adjoint(net, r["z"], w[:, None, None] * r["glogp0"], -w, 0.0, 1.0, 1e-6, 1e-8)
        gx2, gp2 = native.cnf_adjoint(net, r["z"], w[:, None, None] * r["glogp0"], -w, 0.0, 1.0, 1e-6, 1e-8)
        assert torch.equal(gp, gp2) and torch.equal(gx, gx2), f"{mode}: adjoint not reproducible"
        assert int(r["stats"][3]) == 0
        res[mode] = (x, r, gx, gp)
    (xt, rt, gxt, gpt), (xe, re, gxe, gpe) = res["table"], res["exact"]
    assert (xt - xe).abs().max().item() < 1e-11
    rel = ((rt["eloc"] - re["eloc"]).abs() / re["eloc"].abs()).max().item()
    assert rel < 1e-9, rel
    assert (gxt - gxe).abs().max().item() < 1e-9 * gxe.abs().max().item()
    assert (gpt - gpe).abs().max().item() < 1e-9 * gpe.abs().max().item()


def test_config4_six_plus_six(golden, dev):
    """BASELINE config 4 shape (nup = ndown = 6, n = 12): local energy and theta-gradient on the GPU vs the oracle
    (no reference vectors at this size: a reference E_loc here costs minutes per walker)."""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    G = golden["g5_gsvmc"]
    model = Gm._model(dev, 6, 6, 2.0)
    torch.manual_seed(21)
    B = 512
    g = model(B)
    g.backward()
    assert np.isfinite(model.E) and 0 < model.E_std < 60
    net = O.Net(*net_arrays(G, "z2_nt_"))
    xs = N(model.x[:6])
    ref = O.eloc(xs, 6, 6, net, 2.0, rtol=1e-9, atol=1e-11)
    rel = np.abs(N(model.Eloc[:6]) - ref["eloc"]) / np.abs(ref["eloc"])
    assert rel.max() < ELOC_RTOL, rel.max()
    # gradient of the surrogate on these 6 walkers: GPU adjoint vs oracle adjoint
    v = model.cnf.v_wrapper.v
    tu, td = model._tables(dev)
    r = native.eloc(tu, td, 6, 6, v.net(), model.x[:6], 0.0, 1.0, 1e-6, 1e-8, 2.0, True)
    w = (r["eloc"] - r["eloc"].mean()) / 6
    _, gp = native.cnf_adjoint(v.net(), r["z"], w[:, None, None] * r["glogp0"], -w, 0.0, 1.0, 1e-6, 1e-8, need_gx=False)
    zo, dlo, _ = O.cnf_delta_logp(xs, net, rtol=1e-9, atol=1e-11)
    _, g0o, _ = O.logprob(zo, 6, 6)
    wo = N(w)
    _, gpo, _ = O.cnf_adjoint(zo, dlo, wo[:, None, None] * g0o, -wo, net, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(N(gp), gpo, atol=2e-5 * np.abs(gpo).max())


@pytest.mark.parametrize("nup,ndn", [(2, 0), (2, 2), (5, 0), (4, 4), (10, 0), (1, 0), (4, 3), (5, 4), (6, 5)])
def test_other_particle_numbers_vs_oracle(golden, dev, nup, ndn):
    """every particle number 1..12 has fused kernels (odd n: row-layout local-energy kernel): flow, local energy and adjoint vs the oracle."""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    G = golden["g5_gsvmc"]
    n = nup + ndn
    model = Gm._model(dev, nup, ndn, 1.0)
    net = O.Net(*net_arrays(G, "z2_nt_"))
    torch.manual_seed(100 + n)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (6,))
    v = model.cnf.v_wrapper.v
    x = native.cnf_generate(v.net(), z, 0.0, 1.0, 1e-8, 1e-10)
    xo, _ = O.cnf_generate(N(z), net, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(N(x), xo, atol=1e-7)
    r = model.local_energy(x, want_stats=True)
    assert int(r["stats"][3]) == 0
    ref = O.eloc(N(x), nup, ndn, net, 1.0, rtol=1e-9, atol=1e-11)
    assert (np.abs(N(r["eloc"]) - ref["eloc"]) / np.abs(ref["eloc"])).max() < ELOC_RTOL
    np.testing.assert_allclose(N(r["grad"]), ref["grad"], atol=2e-5 * max(1.0, np.abs(ref["grad"]).max()))
    w = (r["eloc"] - r["eloc"].mean()) / 6
    _, gp = native.cnf_adjoint(v.net(), r["z"], w[:, None, None] * r["glogp0"], -w, 0.0, 1.0, 1e-8, 1e-10, need_gx=False)
    zo, dlo, _ = O.cnf_delta_logp(N(x), net, rtol=1e-10, atol=1e-12)
    _, g0o, _ = O.logprob(zo, nup, ndn)
    _, gpo, _ = O.cnf_adjoint(zo, dlo, N(w)[:, None, None] * g0o, -N(w), net, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(N(gp), gpo, atol=2e-5 * np.abs(gpo).max())


# ------------------------------------------------------------------------------------------------ walker schedule
def test_walker_schedule_is_invisible_in_the_results(dev):
    """ff_ode.walker_cost / walker_order, ff_walker_order: the cost classes the flow pass reports put the walkers in
    descending order of cost; running the local-energy pass and the adjoint in that order leaves every per-walker
    result bit-identical (order changes timing only) and the parameter gradient equal to rounding."""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    model = Gm._model(dev, 3, 3, 2.0)
    net = model.cnf.v_wrapper.v.net()
    tu, td = model._tables(dev)
    B = 20000
    torch.manual_seed(5)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    cost = torch.full((B,), -1, dtype=torch.int32, device=dev)
    x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, walker_cost=cost)
    assert torch.equal(x, native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8))
    assert int(cost.min()) >= 1 and int(cost.max()) <= 31
    order = native.walker_order(cost)
    assert torch.equal(order.long().sort().values, torch.arange(B, device=dev))
    c = cost[order.long()]
    assert bool((c[:-1] >= c[1:]).all())
    assert torch.equal(order, native.walker_order(cost))
    steps0, steps1 = torch.empty_like(cost), torch.empty_like(cost)
    r0 = native.eloc(tu, td, 3, 3, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, walker_cost=steps0)
    r1 = native.eloc(tu, td, 3, 3, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, walker_cost=steps1, walker_order=order)
    for k in ("logp", "grad", "lap", "V", "eloc", "z", "dlogp", "glogp0"):
        assert torch.equal(r0[k], r1[k]), k
    assert torch.equal(steps0, steps1)
    # the predictor does its job: the walkers the flow pass marks expensive are the ones the sensitivity pass works on longest
    # (steps0: attempted steps of the sensitivity pass; most walkers take the same few, the tail takes up to ten times as many)
    heavy = (steps0 >= steps0.median() + 2).nonzero().squeeze(1)
    rank = torch.empty(B, dtype=torch.long, device=dev)
    rank[order.long()] = torch.arange(B, device=dev)
    assert heavy.numel() > 0 and (rank[heavy] < B // 20).double().mean() > 0.7
    w = (r0["eloc"] - r0["eloc"].mean()) / B
    gx0, gp0 = native.cnf_adjoint(net, r0["z"], w[:, None, None] * r0["glogp0"], -w, 0.0, 1.0, 1e-6, 1e-8)
    gx1, gp1 = native.cnf_adjoint(net, r0["z"], w[:, None, None] * r0["glogp0"], -w, 0.0, 1.0, 1e-6, 1e-8,
                                  walker_order=native.walker_order(steps0))
    assert torch.equal(gx0, gx1)
    assert (gp0 - gp1).abs().max().item() <= 1e-12 * gp0.abs().max().item()


def test_off_table_radii_are_served_by_the_direct_kernels(golden, dev):
    """Off-table protocol of the forward kernels: with the particles in two clusters 34 apart some pair
    distance exceeds the table's 32, the table kernel leaves its launch id in the table header and the direct kernel queued
    behind it redoes the call -- so the table net and the exact net return bit-identical results (same kernel), for
    the flow, the local energy (one lane and two lanes per direction) and the adjoint.  Walkers inside the table are
    NOT bit-identical between the two nets (different arithmetic), which shows the fallback did run above."""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    for nup, ndn in ((3, 3), (4, 4)):
        n = nup + ndn
        model = Gm._model(dev, nup, ndn, 2.0)
        v = model.cnf.v_wrapper.v
        tu, td = model._tables(dev)
        tab, exact = v.net(radial="table"), v.net(radial="exact")
        g = torch.Generator().manual_seed(3)
        near = torch.randn(40, n, 2, generator=g, dtype=torch.float64).to(dev)
        far = near.clone()
        far[:, ::2, 0] += 17.0
        far[:, 1::2, 0] -= 17.0
        assert torch.cdist(far, far).max() > 33
        for z, same in ((far, True), (near, False)):
            xt = native.cnf_generate(tab, z, 0.0, 1.0, 1e-6, 1e-8)
            xe = native.cnf_generate(exact, z, 0.0, 1.0, 1e-6, 1e-8)
            rt = native.eloc(tu, td, nup, ndn, tab, z, 0.0, 1.0, 1e-6, 1e-8, 2.0, True)
            re = native.eloc(tu, td, nup, ndn, exact, z, 0.0, 1.0, 1e-6, 1e-8, 2.0, True)
            if same:
                assert torch.equal(xt, xe)
                assert torch.isfinite(rt["z"]).all() and torch.isfinite(rt["dlogp"]).all()
                for k in ("z", "dlogp", "grad", "lap"):      # (the flow carries some walkers to |r| ~ 38, where exp(-r^2/2)
                    assert torch.allclose(rt[k], re[k], rtol=0.0, atol=0.0, equal_nan=True), k   # underflows: NaN on both sides)
            else:
                assert not torch.equal(xt, xe) and (xt - xe).abs().max() < 1e-10
                assert not torch.equal(rt["lap"], re["lap"])


def test_step_size_warm_start(dev):
    """ff_ode.walker_h_init/_scale/_out: opening every integration with the step size the previous one
    along the same trajectory settled on saves the probe evaluation and the two tiny first steps of the cold start --
    a quarter of the RHS evaluations -- while the per-step error control, hence the accuracy, stays what it was:
    warm and cold results agree far inside the solver tolerance and are equally close to a tight-tolerance solve."""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    model = Gm._model(dev, 3, 3, 2.0)
    net = model.cnf.v_wrapper.v.net()
    tu, td = model._tables(dev)
    B = 8192
    torch.manual_seed(9)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    f = dict(dtype=torch.float64, device=dev)
    hg, he = torch.zeros(B, **f), torch.zeros(B, **f)
    x, st_c = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, want_stats=True, walker_h_out=hg)
    assert 0.0 < float(hg.min()) and float(hg.max()) <= 1.0
    # entries <= 0 start cold: bit-identical to no warm start at all
    x0 = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, walker_h_init=torch.zeros(B, **f), walker_h_scale=1.0)
    assert torch.equal(x0, x)
    xw, st_w = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, want_stats=True, walker_h_init=hg, walker_h_scale=0.75)
    assert int(st_w[0]) < 0.75 * int(st_c[0]) and (xw - x).abs().max().item() < 1e-6
    tight = native.eloc(tu, td, 3, 3, net, x, 0.0, 1.0, 1e-11, 1e-13, 2.0, True)
    cold = native.eloc(tu, td, 3, 3, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True)
    warm = native.eloc(tu, td, 3, 3, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True,
                       walker_h_init=hg, walker_h_scale=0.6, walker_h_out=he)
    assert int(warm["stats"][0]) < 0.8 * int(cold["stats"][0]) and int(warm["stats"][3]) == 0
    rel = lambda r: ((r["eloc"] - tight["eloc"]).abs() / tight["eloc"].abs()).max().item()
    assert rel(warm) < ELOC_RTOL / 10 and rel(warm) < 10 * rel(cold) + 1e-8, (rel(warm), rel(cold))
    w = (tight["eloc"] - tight["eloc"].mean()) / B
    args = (tight["z"], w[:, None, None] * tight["glogp0"], -w, 0.0, 1.0)
    _, g_t = native.cnf_adjoint(net, *args, 1e-11, 1e-13, need_gx=False)
    _, g_c, sc = native.cnf_adjoint(net, *args, 1e-6, 1e-8, need_gx=False, want_stats=True)
    _, g_w, sw = native.cnf_adjoint(net, *args, 1e-6, 1e-8, need_gx=False, want_stats=True, walker_h_init=he, walker_h_scale=1.25)
    assert int(sw[0]) < 0.75 * int(sc[0])
    err = lambda g: ((g - g_t).norm() / g_t.norm()).item()
    assert err(g_w) < 1e-6 and err(g_w) < 10 * err(g_c) + 1e-8, (err(g_w), err(g_c))
    # ff_ode.walker_h_equal (ABI 108; what the sweeps open the adjoint with since round 5: 1.1 x the FLOW pass's step, rounded down to
    # equal steps of the interval): the same call as with the rounded steps passed explicitly, as many evaluations as the old rule on
    # these weights (two steps either way; the gain is on trained flows: DESIGN.md 4), and as accurate
    hr = 1.0 / torch.ceil(1.0 / (hg * 1.1) - 1e-9).clamp(min=1.0)
    _, g_e, se = native.cnf_adjoint(net, *args, 1e-6, 1e-8, need_gx=False, want_stats=True, walker_h_init=hg, walker_h_scale=1.1, walker_h_equal=True)
    _, g_r, sr = native.cnf_adjoint(net, *args, 1e-6, 1e-8, need_gx=False, want_stats=True, walker_h_init=hr, walker_h_scale=1.0)
    assert torch.equal(se[:4], sr[:4]) and torch.equal(g_e, g_r)
    assert int(se[0]) <= 1.01 * int(sw[0]) and int(se[3]) == 0 and err(g_e) < 1e-6 and err(g_e) < 10 * err(g_c) + 1e-8, (se[:4], sw[:4], err(g_e))
    # the sweep uses it by default; switching it off changes the estimate only at the solver-tolerance level
    Es = []
    for flag in (True, False):
        m = Gm._model(dev, 3, 3, 2.0)
        m.warm_start = flag
        torch.manual_seed(21)
        m(4096); m(4096)                      # second sweep: the flow pass is warm too
        Es.append((m.E, m.E_std))
    assert abs(Es[0][0] - Es[1][0]) < 1e-6 * abs(Es[1][0]) and abs(Es[0][1] - Es[1][1]) < 1e-5 * Es[1][1]


@pytest.mark.parametrize("nup,ndn", [(3, 3), (2, 1)])
def test_local_energy_routing_by_cost_class(dev, nup, ndn, monkeypatch):
    """launch_routed (csrc/ff_cnf_fwd.hip): with cost classes the walkers of class >= ff_ode.heavy_class (12 here; the default at 12
    coordinates is 16 since round 5) run on the one-walker-per-wave kernel (at
    0.3 x the tolerances) beside the throughput kernel that takes everyone else.  Light walkers are untouched by the routing, the
    heavy ones stay within 1e-6 of a tight solve (over seeds: 5e-7 against 2.5e-6 without it), and a walker's result depends on its own class only: not on the
    order of work, not on the rest of the batch.  (6 particles: the matrix-core kernel; 3: the column kernel.)"""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    model = Gm._model(dev, nup, ndn, 2.0)
    net = model.cnf.v_wrapper.v.net()
    tu, td = model._tables(dev)
    B = 30000
    torch.manual_seed(17)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    f = dict(dtype=torch.float64, device=dev)
    hg, cost = torch.zeros(B, **f), torch.zeros(B, dtype=torch.int32, device=dev)
    x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, walker_cost=cost, walker_h_out=hg)
    heavy = cost >= 12
    assert 0 < int(heavy.sum()) < B // 20
    kw = dict(walker_h_scale=model._h_scale_eloc, sens_tol=model.sens_tol, sens_tol_class=model.sens_tol_class,
              walker_h_scale_loose=model._h_scale_loose)
    run = lambda xx, hh, cc, heavy_class=12, **extra: native.eloc(tu, td, nup, ndn, net, xx, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True,
                                                                  walker_h_init=hh, walker_class=cc, heavy_class=heavy_class, **kw, **extra)
    tight = native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-11, 1e-13, 2.0, True)["eloc"]
    routed = run(x, hg, cost)
    ordered = run(x, hg, cost, walker_order=native.walker_order(cost))
    plain = run(x, hg, cost, heavy_class=-1)           # ff_ode.heavy_class < 0: no routing
    assert int(routed["stats"][3]) == 0 and int(plain["stats"][3]) == 0
    for k in ("eloc", "grad", "lap", "z", "dlogp"):
        assert torch.equal(routed[k], ordered[k]), k                      # the order of work is invisible
        assert torch.equal(routed[k][~heavy], plain[k][~heavy]), k        # light walkers: the same kernel, the same numbers
    assert not torch.equal(routed["eloc"][heavy], plain["eloc"][heavy])  # the heavy ones did go through the other kernel
    er, ep = (routed["eloc"] / tight - 1).abs(), (plain["eloc"] / tight - 1).abs()
    assert er[heavy].max().item() < 1e-6 and er[heavy].max().item() <= 2 * ep[heavy].max().item() + 1e-7, (er[heavy].max(), ep[heavy].max())
    # a sub-batch: the same walkers, the same numbers
    idx = torch.cat([heavy.nonzero().squeeze(1)[:7], (~heavy).nonzero().squeeze(1)[:1000]])
    sub = run(x[idx].contiguous(), hg[idx].contiguous(), cost[idx].contiguous())
    assert torch.equal(sub["eloc"], routed["eloc"][idx]) and torch.equal(sub["grad"], routed["grad"][idx])
    # the library's default threshold (ff_ode.heavy_class = 0): 12 below 12 coordinates, 16 at 12 (csrc/ff_cnf_fwd.hip)
    dflt = run(x, hg, cost, heavy_class=0)
    want = run(x, hg, cost, heavy_class=16 if (nup + ndn) * 2 >= 12 else 12)
    assert all(torch.equal(dflt[k], want[k]) for k in ("eloc", "grad", "z"))


def test_rejected_first_steps_do_not_couple_the_walkers_of_a_wave(dev, capsys):
    """The four walkers of a matrix-core wave advance in lockstep: a rejected step of one sends all of them through stage 0 again, which
    must not change anybody's numbers.  16 384 walkers of a TRAINED flow (tests/golden/trained_weights.npz "trained") opened with half
    the interval -- most first steps rejected -- in one call, against the same walkers in calls of three: every output identical, the
    rejections counted once, the call reproducible.  (Written for round 6's retry queue -- docs/attic/retry_queue_r06.patch, DESIGN.md 3p
    -- whose retried integrations had to be the in-place ones bit for bit; kept as the invariant it checks.)"""
    import os
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    model = Gm._model(dev, 3, 3, 2.0)
    _load_weight_set(model, np.load(os.path.join(os.path.dirname(__file__), "golden", "trained_weights.npz")), "trained")
    net = model.cnf.v_wrapper.v.net()
    tu, td = model._tables(dev)
    B = 16384
    torch.manual_seed(23)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8)
    h = torch.full((B,), 0.5, dtype=torch.float64, device=dev)
    run = lambda xx, hh: native.eloc(tu, td, 3, 3, net, xx, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True, walker_h_init=hh, walker_h_scale=1.0,
                                     walker_cost=torch.zeros(xx.shape[0], dtype=torch.int32, device=dev))
    a, b = run(x, h), run(x, h)
    rej = int(a["stats"][2])
    assert int(a["stats"][3]) == 0 and rej > B // 4, a["stats"][:4]
    keys = ("eloc", "logp", "lap", "grad", "z", "dlogp", "glogp0")
    for k in keys:
        assert torch.equal(a[k], b[k]), k
    n3 = 1500      # the first 1 500 walkers three at a time
    rej3 = 0
    for i in range(0, n3, 3):
        s3 = run(x[i:i + 3].contiguous(), h[i:i + 3].contiguous())
        rej3 += int(s3["stats"][2])
        for k in keys:
            assert torch.equal(s3[k], a[k][i:i + 3]), (k, i)
    sub = run(x[:n3].contiguous(), h[:n3].contiguous())
    assert int(sub["stats"][2]) == rej3
    tight = native.eloc(tu, td, 3, 3, net, x, 0.0, 1.0, 1e-11, 1e-13, 2.0, True)["eloc"]
    err = ((a["eloc"] - tight).abs() / tight.abs()).max().item()
    with capsys.disabled():
        print(f"\n[rejected first steps] {B} walkers, {rej} rejected steps; bit-identical to the calls of three; max rel. E_loc error vs a 1e-11 solve {err:.1e}")
    assert err < 1e-5


def test_heavy_walker_route_vs_oracle(dev, capsys):
    """the walkers of cost class >= 12 (a particle passing the origin: 0.4-0.6 % of a batch, and the ones
    with the largest E_loc error) leave the throughput kernel for the one-walker-per-wave kernel at 0.3 x the tolerances
    (launch_routed, csrc/ff_cnf_fwd.hip; the threshold passed explicitly: since round 5 the default at 12 coordinates is 16, a
    tenth as many walkers).  Here EVERY one of them of a 65 536-walker batch -- the production call with the
    sweep's policy -- is compared with the oracle's generic jet arithmetic at rtol 1e-10 (oracle/ff_oracle.c), E_loc within the
    north-star bar, together with the same number of the heaviest walkers that stay on the throughput kernel (classes 9-11)."""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    model = Gm._model(dev, 3, 3, 2.0)
    net = model.cnf.v_wrapper.v.net()
    tu, td = model._tables(dev)
    B = 65536
    torch.manual_seed(29)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    f = dict(dtype=torch.float64, device=dev)
    hg, cost = torch.zeros(B, **f), torch.zeros(B, dtype=torch.int32, device=dev)
    x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, walker_cost=cost, walker_h_out=hg)
    r = native.eloc(tu, td, 3, 3, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True, walker_order=native.walker_order(cost),
                    walker_h_init=hg, walker_h_scale=model._h_scale_eloc, walker_class=cost, sens_tol=model.sens_tol,
                    sens_tol_class=model.sens_tol_class, walker_h_scale_loose=model._h_scale_loose, heavy_class=12)
    assert int(r["stats"][3]) == 0
    heavy = (cost >= 12).nonzero().squeeze(1)
    assert heavy.numel() >= 40, heavy.numel()
    mid = ((cost >= 9) & (cost < 12)).nonzero().squeeze(1)[:heavy.numel()]
    onet = _oracle_net(model)
    out = {}
    for name, idx in (("routed (class >= 12)", heavy), ("throughput kernel, classes 9-11", mid)):
        ref = O.eloc(N(x[idx]), 3, 3, onet, 2.0, rtol=1e-10, atol=1e-12)
        rel = np.abs(N(r["eloc"][idx]) - ref["eloc"]) / np.abs(ref["eloc"])
        gerr = np.abs(N(r["grad"][idx]) - ref["grad"]).max() / np.abs(ref["grad"]).max()
        lerr = np.abs(N(r["logp"][idx]) - ref["logp"]).max()
        out[name] = (idx.numel(), rel.max(), np.median(rel), gerr, lerr)
        assert rel.max() < ELOC_RTOL, (name, rel.max())
        assert gerr < 1e-5 and lerr < 1e-6, (name, gerr, lerr)
    with capsys.disabled():
        for name, (cnt, mx, med, gerr, lerr) in out.items():
            print(f"\n[heavy route vs oracle] {name}: {cnt} walkers, E_loc rel. error max {mx:.2e} median {med:.2e}; grad logp {gerr:.2e}; logp {lerr:.2e}")
    # ff_ode.compact_finish on the routed pass: the heavy route's one-wave kernel finishes its walkers in its own epilogue instead of
    # leaving their sensitivities to the two filtered finish kernels -- same integration, every output to rounding
    rc = native.eloc(tu, td, 3, 3, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True, walker_order=native.walker_order(cost),
                     walker_h_init=hg, walker_h_scale=model._h_scale_eloc, walker_class=cost, sens_tol=model.sens_tol,
                     sens_tol_class=model.sens_tol_class, walker_h_scale_loose=model._h_scale_loose, heavy_class=12, compact=True)
    assert int(rc["stats"][3]) == 0 and int(rc["stats"][0]) == int(r["stats"][0])
    assert torch.equal(rc["z"], r["z"]) and torch.equal(rc["dlogp"], r["dlogp"])
    light = (cost < 12).nonzero().squeeze(1)
    for k in ("eloc", "logp", "lap", "V", "grad", "glogp0"):
        assert torch.equal(rc[k][light], r[k][light]), k                      # the throughput kernel's walkers: the same kernel
        sc = max(1.0, r[k][heavy].abs().max().item())
        assert (rc[k][heavy] - r[k][heavy]).abs().max().item() < 1e-9 * sc, k


@pytest.mark.parametrize("nup,ndn,B", [(3, 3, 65536), (6, 6, 8192)])
def test_sensitivity_tolerance_mechanism(dev, nup, ndn, B):
    """ff_ode.walker_class / sens_tol / sens_tol_class: walkers whose flow-pass cost class is <= sens_tol_class integrate
    the sensitivity components at sens_tol x rtol/atol, the others at rtol/atol.  The sweeps of rounds 2-4 ran it at 10 x / class <= 8;
    since round 5 they pass sens_tol = 1 (one tolerance: on trained flows the factor shows one for one in the worst walkers' E_loc,
    test_headline_policy_error_over_seeds_and_weight_sets) -- the mechanism stays in the ABI and is checked here at 10 x / class <= 8 on
    the synthetic weights: a third fewer RHS evaluations, the strict walkers untouched, the loose ones below 1e-6."""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    model = Gm._model(dev, nup, ndn, 2.0)
    assert model.sens_tol == (1.0 if nup + ndn <= 6 else 10.0)
    net = model.cnf.v_wrapper.v.net()
    tu, td = model._tables(dev)
    torch.manual_seed(13)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    f = dict(dtype=torch.float64, device=dev)
    hg, cost = torch.zeros(B, **f), torch.zeros(B, dtype=torch.int32, device=dev)
    x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, walker_cost=cost, walker_h_out=hg)
    loose = cost <= 8
    assert 0.85 < loose.double().mean().item() < 1.0
    tight = native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-11, 1e-13, 2.0, True)["eloc"]
    a = native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True, walker_h_init=hg,
                    walker_h_scale=model._h_scale_eloc)
    b = native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True, walker_h_init=hg,
                    walker_h_scale=model._h_scale_eloc, walker_class=cost, sens_tol=10.0, sens_tol_class=8, walker_h_scale_loose=1.0)
    assert int(a["stats"][3]) == 0 and int(b["stats"][3]) == 0
    assert int(b["stats"][0]) < 0.8 * int(a["stats"][0]), (a["stats"][:3], b["stats"][:3])
    ra, rb = (a["eloc"] / tight - 1).abs(), (b["eloc"] / tight - 1).abs()
    # (the strict walkers: the plain solve's error -- up to 1.2e-6 in classes 12-15, which the heavy route tightened until its default
    # threshold at 12 coordinates moved to 16 in round 5; the headline bar for every walker is 3e-6)
    assert rb[loose].max().item() < 1e-6 and rb.max().item() < 3e-6, (rb[loose].max(), rb.max())
    assert torch.equal(a["eloc"][~loose], b["eloc"][~loose]) or (rb[~loose].max() <= 2 * ra[~loose].max() + 1e-9)
    assert abs(b["eloc"].mean().item() / a["eloc"].mean().item() - 1) < 1e-7
    Es = []
    for tol in (10.0, 1.0):
        m = Gm._model(dev, nup, ndn, 2.0)
        m.sens_tol, m.sens_tol_class = tol, 8
        torch.manual_seed(21)
        m(4096); m(4096)
        Es.append(m.E)
    assert abs(Es[0] / Es[1] - 1) < 1e-7


def _load_weight_set(model, W, tag):
    v = model.cnf.v_wrapper.v
    with torch.no_grad():
        for nm, m in (("eta", v.eta), ("mu", v.mu)):
            m.fc1.weight.copy_(torch.as_tensor(W[f"{tag}_{nm}_w1"]).reshape(-1, 1))
            m.fc1.bias.copy_(torch.as_tensor(W[f"{tag}_{nm}_b1"]))
            m.fc2.weight.copy_(torch.as_tensor(W[f"{tag}_{nm}_w2"]).reshape(1, -1))


@pytest.mark.parametrize("tag", ["head", "trained", "driver", "driver1000", "soak3000"])
def test_headline_policy_error_over_seeds_and_weight_sets(dev, tag, capsys):
    """VERDICT r04 next #1c.  The production sweep at BASELINE.json configs[1] -- one tolerance for every component, first steps by cost
    class from the learned table, routing of the heavy walkers; GSVMC.forward_from, several sweeps so that the table has settled --
    against a 1e-11 solve of the same walkers: the MAXIMUM relative E_loc error over 5 seeds x 65 536 walkers, on the benchmark's
    synthetic weights ("head") and on three TRAINED weight sets (tests/golden/trained_weights.npz, written by
    tools/probes/policy_error.py on the GPU: head + 300 iterations at lr 1e-4; init_zeros() + 300 and + 1000 iterations of the
    reference's loop at lr 1e-2, src/FermionHO2D.py:40-43,61-72).  Bar of the north star: 1e-5.
    Measured (tools/probes/policy_sweep.py): 8.1e-7 / 1.7e-6 / 4.4e-7 / 6.7e-6 -- and that is the error of a plain rtol = 1e-6 solve
    with a per-walker norm, which is what the sweep now is (the 10 x / class <= 8 policy of rounds 2-4: 4.3e-7 / 1.0e-5 / 4.0e-6 /
    1.3e-5; 5 x / class <= 6: 8.1e-7 / 1.0e-5 / 4.4e-7 / 2.7e-5).  Asserted: 3e-6 (driver1000: 1e-5, where the plain solve itself is at
    6.7e-6), and never worse than the stand-alone one-tolerance call on the same walkers.
    Round 6: "soak3000" -- the strongest flow the repo itself produces, init_zeros() + 3000 iterations of the
    reference's loop at lr 1e-2 on 65 536 walkers (tools/probes/train_fixtures_r06.py; max|w1| = 0.77, 31 evaluations per walker).
    Measured there: sweep 1.6e-6, plain call 3.0e-6 (profiles/r06_a_policy_error_shape_trained.txt); asserted 5e-6."""
    import os
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    model = Gm._model(dev, 3, 3, 2.0)
    if tag != "head":
        _load_weight_set(model, np.load(os.path.join(os.path.dirname(__file__), "golden", "trained_weights.npz")), tag)
    assert model.sens_tol == 1.0 and model.heavy_class == 0 and model.warm_start and model.adaptive_h
    tu, td = model._tables(dev)
    worst, plain, evals = [], [], []
    for seed in range(500, 505):
        torch.manual_seed(seed)
        with torch.no_grad():
            z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (65536,))
        for _ in range(3 if seed == 500 else 1):
            model.forward_from(z)
        model.profile = {"stages": False}
        model.forward_from(z)
        pr, model.profile = model.profile, None
        net = model.cnf.v_wrapper.v.net()
        tight = native.eloc(tu, td, 3, 3, net, model.x, 0.0, 1.0, 1e-11, 1e-13, 2.0, True)["eloc"]
        rel = (model.Eloc - tight).abs() / tight.abs()
        worst.append(rel.max().item())
        one = native.eloc(tu, td, 3, 3, net, model.x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True)["eloc"]      # stand-alone: cold start, no classes
        plain.append(((one - tight).abs() / tight.abs()).max().item())
        evals.append(int(pr["eloc_stats"][0][0].item()) / 65536)
        assert abs(model.Eloc.mean().item() / tight.mean().item() - 1) < 1e-8
    with capsys.disabled():
        print(f"\n[policy error, {tag} weights] max rel. E_loc error vs a 1e-11 solve by seed: " + " ".join(f"{w:.1e}" for w in worst) +
              " | plain one-tolerance call: " + " ".join(f"{w:.1e}" for w in plain) + f"; RHS evaluations per walker {np.mean(evals):.1f}; " +
              "first-step factors by class 2..12: " + " ".join(f"{v:.2f}" for v in model._h_tab[model._h_tab_cur][2:13].tolist()))
    assert max(worst) < {"driver1000": 1e-5, "soak3000": 5e-6}.get(tag, 3e-6), worst
    assert max(worst) <= max(1.5 * max(plain), 3e-6), (worst, plain)


@pytest.mark.parametrize("shape", ["n12", "c5", "c5_f32"])
def test_loosened_sensitivity_tolerance_on_shape_trained_flows(dev, shape, capsys):
    """the sweeps of BASELINE.json configs[3] (6 + 6 particles, d = 2) and configs[4] (10 + 10, d = 3) control the
    SENSITIVITY components of the walkers of flow cost class <= 8 at 10 x / 5 x the reference's rtol / atol
    (fermiflow_amd/VMC.py _init_sweep; the reference has one tolerance: src/NeuralODE/nnModule.py:161-162).  Round 5 showed on config 2 that
    such a factor can pass on the weights it was tuned on only, so here it is pinned on flows TRAINED AT THESE SHAPES with the reference's
    loop -- init_zeros(), Adam, 1000 iterations (src/FermionHO2D.py:40-43,61-72; lr 1e-2 / 3: at 1e-2 both shapes leave the basin in
    the second iteration; tests/golden/trained_weights.npz entries n12_1000 / c5_1000, written by tools/probes/train_fixtures_r06.py) --
    AND on the benchmark's synthetic weights: the production sweep (forward_from, learned first steps settled) against a 1e-11 solve of the
    same walkers with fp64 sensitivity matrices, 3 seeds x 32 768 (16 384) walkers.  "c5_f32": the fp32 sensitivity matrices configs[4] names.
    Measured (profiles/r06_a_policy_error_shape_trained.txt): n12 4.8e-7 (the plain one-tolerance call: 3.6e-7; the largest errors sit in
    classes 12-15, which keep one tolerance), c5 4.1e-7 (plain 9e-8); synthetic weights (round 5): 7.6e-7 / 1.2e-6.
    Asserted: 3e-6 on the synthetic weights (config 2's bound), 1.5e-6 on the shape-trained ones -- 7 x under the 1e-5 bar -- and the
    mean energy to 1e-8."""
    import os
    import __graft_entry__ as Gm
    import fermiflow_amd as ff
    from fermiflow_amd import native
    nup, ndn, dim, B = (6, 6, 2, 32768) if shape == "n12" else (10, 10, 3, 16384)
    bits = 32 if shape == "c5_f32" else 64
    W = np.load(os.path.join(os.path.dirname(__file__), "golden", "trained_weights.npz"))
    rows = []
    try:
        for tag in ("head", "n12_1000" if dim == 2 else "c5_1000"):
            if dim == 2:
                model = Gm._model(dev, nup, ndn, 2.0)
            else:
                gs = Gm._model(dev, 2, 2, 2.0)
                model = ff.GSVMC(nup, ndn, ff.HO3D(), ff.FreeFermion(device=dev), gs.cnf, ff.CoulombPairPotential(2.0), sp_potential=ff.HO())
                model.to(dev)
            if tag != "head":
                _load_weight_set(model, W, tag)
            assert model.sens_tol == (10.0 if dim == 2 else 5.0) and model.sens_tol_class == 8      # the policy under test
            tu, td = model._tables(dev)
            worst, plain = [], []
            for seed in range(600, 603):
                torch.manual_seed(seed)
                with torch.no_grad():
                    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
                native.set_sens_precision(bits)
                for _ in range(3 if seed == 600 else 2):
                    model.forward_from(z)
                e = model.Eloc.clone()
                net = model.cnf.v_wrapper.v.net()
                native.set_sens_precision(64)
                tight = native.eloc(tu, td, nup, ndn, net, model.x, 0.0, 1.0, 1e-11, 1e-13, 2.0, True)["eloc"]
                one = native.eloc(tu, td, nup, ndn, net, model.x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True)["eloc"]
                loose = model.walker_cost <= 8
                assert 0.5 < loose.double().mean().item() <= 1.0
                rel = (e - tight).abs() / tight.abs()
                worst.append(rel.max().item())
                plain.append(((one - tight).abs() / tight.abs()).max().item())
                assert abs(e.mean().item() / tight.mean().item() - 1) < 1e-8
            rows.append((tag, worst, plain))
            assert max(worst) < (3e-6 if tag == "head" else 1.5e-6), (tag, worst, plain)
    finally:
        native.set_sens_precision(64)
    with capsys.disabled():
        for tag, worst, plain in rows:
            print(f"\n[loosened sensitivity tolerance, {shape}, {tag} weights] max rel. E_loc error vs a 1e-11 solve by seed: " +
                  " ".join(f"{w:.1e}" for w in worst) + " | plain one-tolerance call: " + " ".join(f"{w:.1e}" for w in plain))


def test_first_step_table_settles_on_a_trained_flow(dev, capsys):
    """The learned first-step table (ff_walker_schedule) on a flow whose sensitivities need three steps of 1/3 per walker (the "trained"
    set of tests/golden/trained_weights.npz): without a growth condition every class cycled -- no rejections at three steps, x 1.02 for
    four passes until two steps of 1/2 were planned, 40-70 % of them rejected, x 0.93, and again: one pass in five at 24 evaluations per
    walker instead of 21 (round 5, tools/probes/h_table_drift.py).  Growth now needs evidence that the shorter plan would hold
    (ff_scale_update); here: 40 sweeps of fresh walkers on fixed weights, and over the last 20 the evaluations per walker stay within
    +- 2 % of their mean and no factor of a populated class grows."""
    import os
    import __graft_entry__ as Gm
    model = Gm._model(dev, 3, 3, 2.0)
    _load_weight_set(model, np.load(os.path.join(os.path.dirname(__file__), "golden", "trained_weights.npz")), "trained")
    torch.manual_seed(77)
    evals, tabs = [], []
    for it in range(40):
        model.profile = {"stages": False}
        with torch.no_grad():
            model(65536)
        pr, model.profile = model.profile, None
        evals.append(int(pr["eloc_stats"][0][0].item()) / 65536)
        tabs.append(model._h_tab[model._h_tab_cur][2:9].clone())
    late = np.array(evals[20:])
    steps = torch.stack(tabs[20:])[1:] - torch.stack(tabs[20:])[:-1]
    grown, shrunk = int((steps > 1e-12).sum()), int((steps < -1e-12).sum())
    with capsys.disabled():
        print(f"\n[first-step table on a trained flow] evaluations per walker, sweeps 20-39: mean {late.mean():.2f}, min {late.min():.2f}, max {late.max():.2f}; "
              f"factor updates of classes 2..8 in that window: {grown} up, {shrunk} down")
    assert late.max() - late.min() < 0.04 * late.mean(), late
    # (a class whose rejection rate sits at the 20 % threshold may still take a step down on some batch; what must not happen is the
    # climb back towards a plan its walkers have shown no room for)
    assert grown == 0 and shrunk <= 2, (grown, shrunk)


@pytest.mark.parametrize("shape", ["c2", "c5"])
def test_throughput_sampler_distribution_vs_parity_sampler(dev, shape, capsys):
    """VERDICT r04 next #8.  The production Metropolis kernels (FreeFermion.sample: Philox, Box-Muller on the fp32 transcendentals,
    determinant-ratio accept test -- no longer the reference's arithmetic) against the parity-mode kernels
    (sample_with_noise: the reference's arithmetic operation for operation, src/base_dist.py:62-70) fed fp64 torch noise: the
    DISTRIBUTION they sample must be the same.  Five independent batches each; E and E_std of one flow and <sum r^2> of the base
    walkers agree within 4 standard errors of the difference (walkers are independent chains: the standard error of a batch mean is
    sigma / sqrt(B) exactly).  c2: BASELINE.json configs[1], 65 536 walkers of 3 + 3 particles (ff_mcmc_spin_philox_kernel<3>);
    c5: configs[4]'s sixteen-lane sampler, 16 384 walkers of 10 + 10 particles in d = 3."""
    import __graft_entry__ as Gm
    import fermiflow_amd as ff
    if shape == "c2":
        B, n, d = 65536, 6, 2
        model = Gm._model(dev, 3, 3, 2.0)
    else:
        B, n, d = 16384, 20, 3
        gs = Gm._model(dev, 2, 2, 2.0)
        model = ff.GSVMC(10, 10, ff.HO3D(), ff.FreeFermion(device=dev), gs.cnf, ff.CoulombPairPotential(2.0), sp_potential=ff.HO())
    bd, up, dn = model.basedist, model.orbitals_up, model.orbitals_down
    stats = {"philox": [], "noise": []}
    for seed in range(5):
        torch.manual_seed(900 + seed)
        z1 = bd.sample(up, dn, (B,))
        gen = torch.Generator(device=dev); gen.manual_seed(7000 + seed)
        g0 = torch.randn(B, n, d, dtype=torch.float64, device=dev, generator=gen)
        g = torch.randn(100, B, n, d, dtype=torch.float64, device=dev, generator=gen)
        u = torch.rand(100, B, dtype=torch.float64, device=dev, generator=gen)
        z2, _, acc = bd.sample_with_noise(up, dn, g0, g, u)
        del g0, g, u
        for key, z in (("philox", z1), ("noise", z2)):
            model.forward_from(z)
            r2 = (z ** 2).sum(dim=(1, 2))
            stats[key].append((model.E, model.E_std, r2.mean().item(), r2.std().item()))
    a, b = np.array(stats["philox"]), np.array(stats["noise"])
    nE = 5 * B
    sig_E = np.sqrt((a[:, 1] ** 2).mean() + (b[:, 1] ** 2).mean()) / np.sqrt(nE)          # s.e. of the difference of the two 5-batch means
    sig_r = np.sqrt((a[:, 3] ** 2).mean() + (b[:, 3] ** 2).mean()) / np.sqrt(nE)
    dE, dr = a[:, 0].mean() - b[:, 0].mean(), a[:, 2].mean() - b[:, 2].mean()
    with capsys.disabled():
        print(f"\n[sampler distribution, {shape}] E philox {a[:, 0].mean():.5f} vs noise-fed {b[:, 0].mean():.5f}: difference {dE:+.2e} = {dE / sig_E:+.2f} s.e.; "
              f"E_std {a[:, 1].mean():.4f} vs {b[:, 1].mean():.4f}; <sum r^2> {a[:, 2].mean():.5f} vs {b[:, 2].mean():.5f}: {dr / sig_r:+.2f} s.e.")
    assert abs(dE) < 4 * sig_E and abs(dr) < 4 * sig_r, (dE, sig_E, dr, sig_r)
    # E_std: a heavy-tailed quantity (Coulomb cusp), its batch-to-batch spread is the yardstick
    spread = np.sqrt(a[:, 1].var(ddof=1) + b[:, 1].var(ddof=1)) / np.sqrt(5) + 1e-12
    assert abs(a[:, 1].mean() - b[:, 1].mean()) < 4 * spread + 0.02 * b[:, 1].mean()


def test_backward_short_cut_respects_parameter_hooks(dev):
    """VMC._SweepScalar: `gradE.backward()` hands the adjoint's gradient views straight to .grad -- unless a parameter carries a
    tensor hook (register_hook / register_post_accumulate_grad_hook: DDP reducers, hook-based clipping), in which case the ordinary
    autograd path runs and the hook fires; both paths give the same gradients."""
    import __graft_entry__ as Gm
    model = Gm._model(dev, 3, 3, 2.0)
    torch.manual_seed(3)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (2048,))
    g = model.forward_from(z); g.backward()
    fast = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    seen = []
    p0 = next(model.parameters())
    h1 = p0.register_hook(lambda gr: seen.append("pre") or gr)
    h2 = p0.register_post_accumulate_grad_hook(lambda p: seen.append("post"))
    g = model.forward_from(z); g.backward()
    h1.remove(); h2.remove()
    assert seen == ["pre", "post"]
    for a, p in zip(fast, model.parameters()):      # (two sweeps: the second one opens with the first one's step sizes -- solver-tolerance level)
        np.testing.assert_allclose(N(p.grad), N(a), rtol=1e-5, atol=1e-7 * float(a.abs().max()))


def test_fused_adam_equals_torch_adam(dev):
    """fermiflow_amd.utils.FusedAdam (ff_adam_step: every tensor of the flow in one launch) against torch.optim.Adam's default
    implementation on the same gradients (src/FermionHO2D.py:61 builds the latter): parameters to 2e-15 over ten steps, a changed lr
    honoured (param_groups), and the optimizer state moves between the two classes in both directions."""
    import __graft_entry__ as Gm
    from fermiflow_amd.utils import FusedAdam, make_adam
    ma, mb = Gm._model(dev, 3, 3, 2.0), Gm._model(dev, 3, 3, 2.0)
    oa, ob = make_adam(ma.parameters(), lr=1e-2), torch.optim.Adam(mb.parameters(), lr=1e-2)
    assert isinstance(oa, FusedAdam)
    gen = torch.Generator(device="cpu").manual_seed(5)
    for it in range(10):
        if it == 6:
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 3e-3
        for pa, pb in zip(ma.parameters(), mb.parameters()):
            g = torch.randn(pa.shape, generator=gen, dtype=torch.float64).to(dev)
            pa.grad, pb.grad = g.clone(), g.clone()
        oa.step(); ob.step()
        for pa, pb in zip(ma.parameters(), mb.parameters()):
            assert torch.allclose(pa, pb, rtol=2e-15, atol=1e-17), it
    # state_dicts are interchangeable: continue each run with the OTHER class
    oc, od = torch.optim.Adam(ma.parameters(), lr=1e-2), FusedAdam(mb.parameters(), lr=1e-2)
    oc.load_state_dict(oa.state_dict()); od.load_state_dict(ob.state_dict())
    for it in range(3):
        for pa, pb in zip(ma.parameters(), mb.parameters()):
            g = torch.randn(pa.shape, generator=gen, dtype=torch.float64).to(dev)
            pa.grad, pb.grad = g.clone(), g.clone()
        oc.step(); od.step()
        for pa, pb in zip(ma.parameters(), mb.parameters()):
            assert torch.allclose(pa, pb, rtol=2e-15, atol=1e-17), it
    # a checkpoint of PyTorch's FUSED Adam (rounds 2-4 of this package: `step` lives on the device there) loads too, its step counts brought
    # to the host
    of = torch.optim.Adam(ma.parameters(), lr=1e-2, fused=True)
    for pa in ma.parameters():
        pa.grad = torch.ones_like(pa)
    of.step()
    oe = FusedAdam(ma.parameters(), lr=1e-2); oe.load_state_dict(of.state_dict())
    assert all(not st["step"].is_cuda and float(st["step"]) == 1.0 for st in oe.state.values())
    oe.step()
    assert all(float(st["step"]) == 2.0 for st in oe.state.values())
    # what it does not serve it refuses
    q = torch.nn.Parameter(torch.zeros(3, dtype=torch.float32, device=dev)); q.grad = torch.ones_like(q)
    with pytest.raises(RuntimeError):
        FusedAdam([q], lr=1e-2).step()
    assert isinstance(make_adam([q], lr=1e-2), torch.optim.Adam)
    # an amsgrad / maximize state is refused (not run as plain Adam), and a refused step moves no step count
    oam = torch.optim.Adam(ma.parameters(), lr=1e-2, amsgrad=True)
    oam.step()
    with pytest.raises(RuntimeError):
        FusedAdam(ma.parameters(), lr=1e-2).load_state_dict(oam.state_dict())
    good = torch.nn.Parameter(torch.zeros(3, dtype=torch.float64, device=dev)); good.grad = torch.ones_like(good)
    mixed = FusedAdam([good, q], lr=1e-2)
    with pytest.raises(RuntimeError):
        mixed.step()
    assert len(mixed.state[good]) == 0      # validated before anything was counted


def test_sweep_scalar_backward_short_cut_equals_autograd(dev):
    """The scalar a sweep returns hands its pre-computed gradient to .grad directly when the training loop calls `.backward()` on it
    (VMC._SweepScalar); `model.fast_backward = False` -- what a model wrapped in DistributedDataParallel sets -- takes the
    ordinary autograd path: the same gradients, and a schedule state on another device follows the walkers (_h_counts)."""
    import __graft_entry__ as Gm
    grads = []
    for fast in (True, False):
        model = Gm._model(dev, 3, 3, 2.0)
        model.fast_backward = fast
        torch.manual_seed(9)
        g = model(4096)
        assert hasattr(g, "_ff_fast") == fast
        g.backward()
        grads.append([p.grad.clone() for p in model.parameters()])
    for a, b in zip(*grads):
        assert torch.equal(a, b)
    from fermiflow_amd import native
    cost = torch.zeros(64, dtype=torch.int32, device=dev); hg = torch.full((64,), 0.5, dtype=torch.float64, device=dev)
    tab = torch.ones(2, 32, dtype=torch.float64, device=dev)
    with pytest.raises(ValueError):      # counts on the host: refused, not handed to a kernel as a device pointer
        native.walker_schedule(cost, hg, tab[0], tab[1], None, interval=1.0, counts=torch.zeros(native.SCALE_COUNTS, dtype=torch.float64), shrink_at=0.1)


def test_walker_prefetch_changes_nothing_but_the_schedule(dev):
    """GSVMC.prefetch_walkers (default on): the Metropolis kernels of the next two iterations run on a side stream, released behind
    this iteration's adjoint kernel (ff_ode.after_main_event).  Same seeds in the same order -> the same walkers: three training
    iterations with and without it end in bit-identical energies and parameters; the prefetched walkers are part of the checkpoint
    state."""
    import __graft_entry__ as Gm
    from fermiflow_amd.utils import make_adam
    out = []
    for flag in (True, False):
        m = Gm._model(dev, 3, 3, 2.0)
        m.prefetch_walkers = flag
        opt = make_adam(m.parameters(), lr=1e-3)
        torch.manual_seed(31)
        Es = []
        for it in range(3):
            opt.zero_grad()
            m(4096).backward()
            opt.step()
            Es.append((m.E, m.E_std))
        torch.cuda.synchronize()
        out.append((Es, [p.detach().clone() for p in m.parameters()], m))
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert torch.equal(a, b)
    st_on, st_off = out[0][2].get_extra_state(), out[1][2].get_extra_state()
    assert len(st_on["z_queue"]) == 2 and all(e["z"].shape == (4096, 6, 2) for e in st_on["z_queue"]) and "z_queue" not in st_off      # two batches ahead
    # re-seeding between iterations is honoured: the prefetched walkers are dropped when torch's generator was touched
    m = out[0][2]
    torch.manual_seed(77); m(4096); e1 = m.E
    torch.manual_seed(77); m(4096); e2 = m.E            # same walkers again (the warm-start step sizes differ: not the same bits)
    m(4096); e3 = m.E                                   # the next, prefetched, batch: other walkers
    assert abs(e1 - e2) < 1e-7 * abs(e1) and abs(e3 - e1) > 1e-4 * abs(e1)


def test_persistent_walkers_opt_in(dev):
    """SURVEY 8(f).1, off by default: ff_mcmc_continue is the same chain as ff_mcmc_sample_noise fed the walkers and the
    materialised Philox stream; a GSVMC that keeps its walkers and advances them 10 steps per sweep samples the same
    distribution (energy agrees with fresh 100-step walkers within the statistical error)."""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    model = Gm._model(dev, 3, 3, 2.0)
    tu, td = model._tables(dev)
    B, steps = 4096, 25
    torch.manual_seed(2)
    x0 = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    _, g, u = native.rng_fill(B, 6, steps, 77, dev, walker_offset=5)
    xr, lr, acc = native.mcmc_sample_noise(tu, td, 3, 3, x0, g, u, 0.1)
    xc, lc, cnt = native.mcmc_continue(tu, td, 3, 3, x0, steps, 0.1, 77, walker_offset=5)
    assert torch.equal(xc, xr) and torch.equal(lc, lr) and torch.equal(cnt.long(), acc.long().sum(0))
    Es = {}
    for flag in (False, True):
        m = Gm._model(dev, 3, 3, 2.0)
        m.persistent_walkers = flag
        torch.manual_seed(4)
        vals = []
        for _ in range(6):
            m(16384)
            vals.append((m.E, m.E_std))
        Es[flag] = vals
    e_fresh = sum(v[0] for v in Es[False]) / 6
    e_keep = sum(v[0] for v in Es[True][1:]) / 5
    sigma = max(v[1] for v in Es[False]) / (16384 ** 0.5)
    assert abs(e_keep - e_fresh) < 6 * sigma, (e_keep, e_fresh, sigma)


def test_adjoint_is_linear_in_its_seeds_full_size(dev):
    """Size-independent property of the theta-gradient adjoint at the benchmark's walker count: the result is linear
    in the incoming gradients (a_z, a_Delta) up to the solver tolerance, and invariant under a permutation of the
    walkers (a sum over walkers) to rounding."""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    model = Gm._model(dev, 3, 3, 2.0)
    net = model.cnf.v_wrapper.v.net()
    B = 65536
    torch.manual_seed(13)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    g = torch.Generator(device="cpu").manual_seed(1)
    a1, a2 = (torch.randn(B, 6, 2, generator=g, dtype=torch.float64).to(dev) / B for _ in range(2))
    d1, d2 = (torch.randn(B, generator=g, dtype=torch.float64).to(dev) / B for _ in range(2))
    run = lambda zz, az, ad: native.cnf_adjoint(net, zz, az, ad, 0.0, 1.0, 1e-9, 1e-11, need_gx=False)[1]
    g1, g2, g12 = run(z, a1, d1), run(z, a2, d2), run(z, a1 + 2.0 * a2, d1 + 2.0 * d2)
    scale = (g1.abs() + 2.0 * g2.abs()).max().item()
    assert (g12 - (g1 + 2.0 * g2)).abs().max().item() < 1e-7 * scale
    perm = torch.randperm(B, generator=g).to(dev)
    gp = run(z[perm].contiguous(), a1[perm].contiguous(), d1[perm].contiguous())
    assert (gp - g1).abs().max().item() < 1e-11 * g1.abs().max().item()


def test_local_energy_is_invariant_under_same_spin_exchange(dev):
    """Domain property at full size: |psi|^2 of the flow-transformed Slater state is symmetric under the exchange of two
    same-spin particles (the backflow field is permutation equivariant, the determinant changes sign only), so logp and
    E_loc are unchanged -- through the flow solve, its sensitivities and the Slater contraction."""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    model = Gm._model(dev, 3, 3, 2.0)
    net = model.cnf.v_wrapper.v.net()
    tu, td = model._tables(dev)
    B = 32768
    torch.manual_seed(17)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8)
    xs = x.clone()
    xs[:, [0, 2]] = x[:, [2, 0]]          # two spin-up particles
    xs[:, [3, 4]] = x[:, [4, 3]]          # two spin-down particles
    r = native.eloc(tu, td, 3, 3, net, x, 0.0, 1.0, 1e-9, 1e-11, 2.0, True)
    rs = native.eloc(tu, td, 3, 3, net, xs, 0.0, 1.0, 1e-9, 1e-11, 2.0, True)
    assert ((r["logp"] - rs["logp"]).abs() / r["logp"].abs()).max().item() < 1e-7
    assert ((r["eloc"] - rs["eloc"]).abs() / r["eloc"].abs()).max().item() < 1e-6
    assert (r["grad"][:, [2, 1, 0, 4, 3, 5]] - rs["grad"]).abs().max().item() < 1e-5 * r["grad"].abs().max().item()


@pytest.mark.parametrize("n,d,B", [(6, 2, 1000), (12, 2, 129), (6, 3, 64), (7, 2, 333)])
def test_potential_kernels_against_the_formula(dev, n, d, B):
    """ff_potential (HO.V + CoulombPairPotential.V, src/potentials.py:13,23-47): the HBM-streaming instantiations and the
    generic kernel (n = 7 has no instantiation) against the plain formula, ragged batch sizes."""
    from fermiflow_amd import native
    g = torch.Generator().manual_seed(n * 100 + d)
    x = torch.randn(B, n, d, generator=g, dtype=torch.float64).to(dev)
    rij = (x[:, :, None] - x[:, None]).norm(dim=-1)
    iu = torch.triu_indices(n, n, 1, device=dev)
    ref = 1.7 / rij[:, iu[0], iu[1]]
    np.testing.assert_allclose(N(native.potential(x, 1.7, True)), N(ref.sum(1) + 0.5 * (x ** 2).sum((1, 2))), rtol=1e-13)
    np.testing.assert_allclose(N(native.potential(x, 1.7, False)), N(ref.sum(1)), rtol=1e-13)


# ------------------------------------------------------------------------------------------------ checkpoints
@pytest.mark.parametrize("warm", ["1", "0"])
def test_checkpoint_resume_reproduces_the_uninterrupted_run(dev, tmp_path, monkeypatch, warm, capsys):
    """--save / --resume of the ground-state driver: 2 iterations + resume + 1 iteration == 3 iterations, bit for bit
    (parameters, Adam moments, E of the third iteration) -- with the step-size warm start on (its state travels in the
    checkpoint) and off."""
    from fermiflow_amd import FermionHO2D
    monkeypatch.setenv("FERMIFLOW_WARM_START", warm)
    a, b = str(tmp_path / "a.pt"), str(tmp_path / "b.pt")
    common = ["--nup", "3", "--ndown", "3", "--Z", "2.0", "--batch", "2048"]
    torch.manual_seed(11)
    FermionHO2D.main(common + ["--iternum", "3", "--save", a])
    out_a = capsys.readouterr().out
    torch.manual_seed(11)
    FermionHO2D.main(common + ["--iternum", "2", "--save", b])
    torch.manual_seed(999)          # the resumed run must take its RNG streams from the checkpoint, not from here
    FermionHO2D.main(common + ["--iternum", "1", "--resume", b, "--save", b])
    out_b = capsys.readouterr().out
    ca, cb = torch.load(a, weights_only=False), torch.load(b, weights_only=False)
    assert ca["iter"] == cb["iter"] == 3
    for k, v in ca["model"].items():
        if isinstance(v, torch.Tensor):
            assert torch.equal(v, cb["model"][k]), k
    for sa, sb in zip(ca["optimizer"]["state"].values(), cb["optimizer"]["state"].values()):
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
    line3 = lambda o: [l for l in o.splitlines() if l.startswith("iter: 003")][0].split("Instant")[0]
    assert line3(out_a) == line3(out_b)


def test_betavmc_checkpoint_and_two_spin_species(dev, tmp_path, capsys):
    """Finite-temperature driver: --ndown != 0 runs (the reference rejects it), the reference default
    --beta 2.0 is back, and --save/--resume continues the run (E/F of the resumed iteration equal the uninterrupted one's
    to reduction-order noise: the per-state sums use device atomics)."""
    from fermiflow_amd import BetaFermionHO2D
    a, b = str(tmp_path / "a.pt"), str(tmp_path / "b.pt")
    common = ["--nup", "2", "--ndown", "1", "--Z", "1.0", "--deltaE", "1.0", "--boltzmann", "--batch", "1024"]
    torch.manual_seed(5)
    BetaFermionHO2D.main(common + ["--iternum", "3", "--save", a])
    out_a = capsys.readouterr().out
    assert "beta = 2.0" in out_a and "total number of states = 10" in out_a
    torch.manual_seed(5)
    BetaFermionHO2D.main(common + ["--iternum", "2", "--save", b])
    BetaFermionHO2D.main(common + ["--iternum", "1", "--resume", b, "--save", b])
    out_b = capsys.readouterr().out
    ca, cb = torch.load(a, weights_only=False), torch.load(b, weights_only=False)
    for k, v in ca["model"].items():
        if isinstance(v, torch.Tensor):
            np.testing.assert_allclose(N(v), N(cb["model"][k]), rtol=1e-9, atol=1e-12, err_msg=k)
    vals = lambda o: [float(t) for t in [l for l in o.splitlines() if l.startswith("iter: 003")][0].split("Instant")[0].replace(":", " ").split() if t.replace(".", "").replace("-", "").replace("e", "").isdigit()]
    np.testing.assert_allclose(vals(out_a), vals(out_b), rtol=1e-9)


def test_betavmc_two_spin_species_vs_oracle(dev):
    """nup = 2, ndown = 1, 10 states: Metropolis chain bit-exact and local energies within the bar against the oracle,
    every walker in its own (up, down) orbital pair."""
    import fermiflow_amd as ff
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    h = ff.HO2D()
    states, Es = h.fermion_states(2, 1, 1.0)
    up = np.array([[o.k for o in s[0]] for s in states], dtype=np.int32)
    dn = np.array([[o.k for o in s[1]] for s in states], dtype=np.int32)
    B, S = 200, 40
    rng = np.random.RandomState(1)
    ws = np.sort(rng.randint(0, len(states), B)).astype(np.int32)
    g0, g, u = rng.randn(B, 3, 2), rng.randn(S, B, 3, 2), rng.rand(S, B)
    tu, td = native.orbital_table(up, dev), native.orbital_table(dn, dev)
    x, logp, acc = native.mcmc_sample_noise(tu, td, 2, 1, T(g0, dev), T(g, dev), T(u, dev), walker_state=T(ws, dev, torch.int32))
    xo, lo, ao = O.mcmc_noise(g0, g, u, 2, 1, tab_up=up, tab_dn=dn, wstate=ws)
    assert (N(acc) == ao).all() and (N(x) == xo).all()
    gs = Gm._model(dev, 2, 1, 1.0)
    model = ff.BetaVMC(2.0, 2, 1, 1.0, True, h, ff.FreeFermion(device=dev), gs.cnf, ff.CoulombPairPotential(1.0), sp_potential=ff.HO())
    model.to(dev)
    assert model.Nstates == 10
    xs = gs.cnf.generate(x)
    r = model.local_energy(xs, T(ws, dev, torch.int32))
    v = gs.cnf.v_wrapper.v
    net = O.Net(tuple(N(t) for t in (v.eta.fc1.weight, v.eta.fc1.bias, v.eta.fc2.weight)),
                tuple(N(t) for t in (v.mu.fc1.weight, v.mu.fc1.bias, v.mu.fc2.weight)))
    ref = O.eloc(N(xs)[:64], 2, 1, net, 1.0, rtol=1e-10, atol=1e-12, tab_up=up, tab_dn=dn, wstate=ws[:64])
    rel = np.abs(N(r["eloc"])[:64] - ref["eloc"]) / np.abs(ref["eloc"])
    assert rel.max() < ELOC_RTOL, rel.max()
    np.testing.assert_allclose(N(r["logp"])[:64], ref["logp"], atol=1e-6)
    torch.manual_seed(3)
    gphi, gtheta = model(1024)
    (gphi + gtheta).backward()
    assert np.isfinite([model.E, model.F, model.S]).all()


# ------------------------------------------------------------------------------------------------ shapes (VERDICT r01 missing #3)
@pytest.mark.parametrize("He,Hm,radial", [(100, 70, "table"), (100, 70, "exact"), (200, 256, "exact")])
def test_hidden_widths_beyond_64(dev, He, Hm, radial):
    """--Deta / --Dmu other than 50 (src/FermionHO2D.py:24-27): flow, local energy and parameter gradient against the
    oracle; with direct evaluation the adjoint runs one launch per chunk of hidden units."""
    import fermiflow_amd as ff
    from fermiflow_amd import native
    rng = np.random.default_rng(He + Hm)
    sc = 4.0 / np.sqrt(He)
    eta = [rng.normal(size=He) * 0.5, rng.normal(size=He) * 0.3, rng.normal(size=He) * 0.05 * sc]
    mu = [rng.normal(size=Hm) * 0.5, rng.normal(size=Hm) * 0.3, rng.normal(size=Hm) * 0.05 * sc]
    cnf = make_flow(eta, mu, dev)
    v = cnf.v_wrapper.v
    net = v.net(radial=radial)
    onet = O.Net(eta, mu)
    B = 16
    z = rng.normal(size=(B, 6, 2)) * 1.2
    x = native.cnf_generate(net, T(z, dev), 0.0, 1.0, 1e-9, 1e-11)
    xo, _ = O.cnf_generate(z, onet, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(N(x), xo, atol=1e-8)
    tu = native.orbital_table([0, 1, 2], dev)
    r = native.eloc(tu, tu, 3, 3, net, T(xo, dev), 0.0, 1.0, 1e-9, 1e-11, 2.0, True, want_stats=True)
    assert int(r["stats"][3]) == 0
    ref = O.eloc(xo, 3, 3, onet, 2.0, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(N(r["eloc"]), ref["eloc"], rtol=1e-7)
    zo, dlo, _ = O.cnf_delta_logp(xo, onet, rtol=1e-11, atol=1e-13)
    az, ad = rng.normal(size=z.shape), rng.normal(size=B)
    gxo, gpo, _ = O.cnf_adjoint(zo, dlo, az, ad, onet, rtol=1e-11, atol=1e-13)
    gx, gp = native.cnf_adjoint(net, T(zo, dev), T(az, dev), T(ad, dev), 0.0, 1.0, 1e-9, 1e-11)
    np.testing.assert_allclose(N(gx), gxo, atol=1e-7)
    np.testing.assert_allclose(N(gp), gpo, atol=2e-7 * max(1.0, np.abs(gpo).max()))


@pytest.mark.parametrize("kind", ["mfma", "rows", "columns"])
def test_three_local_energy_kernels_agree_on_the_gpu(kind):
    """FF_ELOC_KERNEL = mfma | rows | columns (read once per process, hence a child process): the matrix-core kernel
    (v_mfma_f64_4x4x4), the row-layout kernel and the column sweep give the reference's local energies."""
    import subprocess, sys, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, json, torch; from fermiflow_amd import native; import fermiflow_amd as ff;"
            "from tests.common import net_arrays; from tests.test_gpu_parity import make_flow, T, N;"
            "G=np.load('tests/golden/g5_gsvmc.npz'); dev=torch.device('cuda:0'); eta,mu=net_arrays(G,'z2_nt_');"
            "cnf=make_flow(eta,mu,dev); tu=native.orbital_table([0,1,2],dev);"
            "r=native.eloc(tu,tu,3,3,cnf.v_wrapper.v.net(),T(G['z2_nt_x'],dev),0.0,1.0,1e-9,1e-11,2.0,True,want_stats=True);"
            "print(json.dumps([N(r['eloc']).tolist(), N(r['lap']).tolist(), r['stats'][:4].tolist()]))")
    out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, FF_ELOC_KERNEL=kind), cwd=root, timeout=300)
    el, lap, st = json.loads(out.decode().strip().splitlines()[-1])
    G = np.load(os.path.join(root, "tests", "golden", "g5_gsvmc.npz"))
    assert st[3] == 0
    np.testing.assert_allclose(el, G["z2_nt_Eloc"], rtol=1e-8)
    np.testing.assert_allclose(lap, G["z2_nt_lap"], rtol=1e-7, atol=1e-6)


# ------------------------------------------------------------------------------------------------ configs 3 and 4 at BASELINE size
@pytest.mark.parametrize("beta", [10.0, 1.0])
def test_config3_betavmc_full_size_known_answer(dev, beta):
    """BASELINE.json configs[2] (beta = 10, nup = 3, boltzmann) at 65536 walkers with the driver's zero-initialised flow
    and Z = 0: every walker's local energy is exactly the energy of the many-body state it was drawn in, so
    E = sum_s p_s E_s up to the sampling of the states, F = E - S/beta, and S agrees with the analytic entropy.
    beta = 10 is the BASELINE configuration (the ground state holds almost every walker); beta = 1 populates all 21 states."""
    import fermiflow_amd as ff
    eta, mu = ff.MLP(1, 50), ff.MLP(1, 50)
    eta.init_zeros(); mu.init_zeros()
    cnf = ff.CNF(ff.Backflow(eta, mu=mu), (0.0, 1.0))
    model = ff.BetaVMC(beta, 3, 0, 2.0, True, ff.HO2D(), ff.FreeFermion(device=dev), cnf, ff.CoulombPairPotential(0.0), sp_potential=ff.HO())
    model.to(dev)
    torch.manual_seed(1)
    B = 65536
    gphi, gtheta = model(B)
    (gphi + gtheta).backward()
    Es = model.Es_original.to(dev)
    ws = model._ws.long()
    assert (model.Eloc - Es[ws]).abs().max().item() < 1e-8            # eigenfunction KAT, state by state
    p = torch.softmax(model.log_state_weights.detach(), dim=0)
    E_exact = (p * Es).sum().item()
    assert abs(model.E - E_exact) < 6 * model.E_std / np.sqrt(B)
    assert abs(model.S - model.S_analytical) < 0.02 and abs(model.F - (model.E - model.S / model.beta)) < 1e-9
    counts = torch.bincount(ws, minlength=model.Nstates).double()
    assert ((counts / B - p).abs() < 6 * (p * (1 - p) / B).sqrt() + 1e-12).all()       # the state list is a sample of softmax(logits)
    assert torch.isfinite(model.log_state_weights.grad).all() and all(torch.isfinite(q.grad).all() for q in cnf.parameters())


def test_config4_six_plus_six_full_size_known_answer(dev):
    """BASELINE.json configs[3] per GPU: nup = ndown = 6, 32768 walkers; zero flow, Z = 0 -> E_loc = 2 (1+2+2+3+3+3) = 28
    for every walker (row-layout local-energy kernel, split-determinant Metropolis kernel, 6 x 6 Slater finish)."""
    import __graft_entry__ as Gm
    model = Gm._model(dev, 6, 6, 0.0)
    for p in model.parameters():
        torch.nn.init.zeros_(p)
    torch.manual_seed(2)
    g = model(32768)
    g.backward()
    assert (model.Eloc - 28.0).abs().max().item() < 1e-7
    assert abs(model.E - 28.0) < 1e-9 and model.E_std < 1e-7
    # and with the benchmark's non-trivial flow: a sample of walkers against the oracle
    model = Gm._model(dev, 6, 6, 2.0)
    torch.manual_seed(3)
    model(4096)
    v = model.cnf.v_wrapper.v
    net = O.Net(tuple(N(t) for t in (v.eta.fc1.weight, v.eta.fc1.bias, v.eta.fc2.weight)),
                tuple(N(t) for t in (v.mu.fc1.weight, v.mu.fc1.bias, v.mu.fc2.weight)))
    ref = O.eloc(N(model.x[:24]), 6, 6, net, 2.0, rtol=1e-10, atol=1e-12)
    rel = np.abs(N(model.Eloc[:24]) - ref["eloc"]) / np.abs(ref["eloc"])
    assert rel.max() < ELOC_RTOL, rel.max()


# ------------------------------------------------------------------------------------------------ three dimensions / fp32
def test_ho3d_base_distribution(dev):
    """HO3D orbitals, FreeFermion.log_prob / y_grad_laplacian / sample in d = 3: against the oracle, bit-exact Metropolis on
    explicit noise, and the eigenfunction known-answer test at BASELINE size: nup = ndown = 10 (closed shells 0..2, the
    occupation of configs[4]) on 65536 random points, E_loc == 2 (1.5 + 3*2.5 + 6*3.5) = 60."""
    import fermiflow_amd as ff
    from fermiflow_amd import native
    h = ff.HO3D()
    bd = ff.FreeFermion(device=dev)
    rng = np.random.RandomState(5)
    iu, idn = np.sort(rng.choice(20, 3, replace=False)), np.sort(rng.choice(20, 6, replace=False))
    up, dn = tuple(h.orbitals[k] for k in iu), tuple(h.orbitals[k] for k in idn)
    x = rng.randn(20, 9, 3)
    lpo, go, lapo = O.logprob3d(x, 3, 6, tab_up=iu, tab_dn=idn)
    np.testing.assert_allclose(N(bd.log_prob(up, dn, T(x, dev))), lpo, atol=1e-11)
    lp, g, lap = ff.y_grad_laplacian(ff.utils.freefermion_logp(bd, up, dn), T(x, dev))
    np.testing.assert_allclose(N(g), go, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(N(lap), lapo, rtol=1e-8, atol=1e-6)
    B, S = 64, 50
    g0, g, u = rng.randn(B, 9, 3), rng.randn(S, B, 9, 3), rng.rand(S, B)
    xs, lps, acc = bd.sample_with_noise(up, dn, T(g0, dev), T(g, dev), T(u, dev))
    xo, lpo2, acco = O.mcmc_noise3d(g0, g, u, 3, 6, tab_up=iu, tab_dn=idn)
    assert (N(acc) == acco).all() and (N(xs) == xo).all()
    # full size: closed shells, every point an eigenfunction value
    torch.manual_seed(0)
    xb = torch.randn(65536, 20, 3, dtype=torch.float64, device=dev)
    lp, g, lap = ff.y_grad_laplacian(ff.utils.freefermion_logp(bd, tuple(h.orbitals[:10]), tuple(h.orbitals[:10])), xb)
    eloc = -0.25 * lap - 0.125 * (g ** 2).sum(dim=(1, 2)) + 0.5 * (xb ** 2).sum(dim=(1, 2))
    err = (eloc - 60.0).abs() / 60.0
    # (random points make some 10 x 10 determinants nearly singular: the Laplacian of those few cancels badly)
    assert err.median().item() < 1e-12 and (err > 1e-8).double().mean().item() < 2e-3 and err.max().item() < 1e-2
    z = bd.sample(tuple(h.orbitals[:4]), tuple(h.orbitals[:4]), (4096,), equilibrim_steps=1000)          # Philox sampler, d = 3
    assert z.shape == (4096, 8, 3) and torch.isfinite(z).all()
    # virial theorem of the oscillator: <sum r^2> = E = 2 (1.5 + 3 * 2.5) = 18 over 24 coordinates (the chain starts at 1.0 per
    # coordinate and relaxes in a few hundred steps of tau = 0.1)
    assert abs((z ** 2).mean().item() - 18.0 / 24.0) < 0.04


def test_fp32_backflow_error_report(golden, dev, capsys):
    """ff_backflow_v_div_f32 against the fp64 kernel on the benchmark's weights, 65536 walkers of six particles: the
    single-precision error this path would carry (max over walkers, relative to the largest entry)."""
    from fermiflow_amd import native
    import __graft_entry__ as Gm
    model = Gm._model(dev, 3, 3, 2.0)
    net = model.cnf.v_wrapper.v.net(radial="exact")
    torch.manual_seed(1)
    x = torch.randn(65536, 6, 2, dtype=torch.float64, device=dev) * 1.3
    v64, d64 = native.backflow_v_div(net, x)
    v32, d32 = native.backflow_v_div_f32(net, x)
    ev = ((v32 - v64).abs().max() / v64.abs().max()).item(); ed = ((d32 - d64).abs().max() / d64.abs().max()).item()
    with capsys.disabled():
        print(f"\n[fp32 backflow] max |v32 - v64| / max |v64| = {ev:.2e}, max |div32 - div64| / max |div64| = {ed:.2e}")
    assert 1e-9 < ev < 5e-5 and ed < 5e-5


def test_gsvmc_sweep_in_three_dimensions(dev):
    """A whole VMC iteration with HO3D orbitals (nup = ndown = 2, walkers (B, 4, 3)): the d = 3 sampler, the fused flow /
    sensitivity / adjoint kernels with D = 3 and the d = 3 finish.  Zero flow, Z = 0: E = 2 (1.5 + 2.5) = 8 for every walker;
    with the benchmark's flow: per-walker E_loc against the oracle."""
    import fermiflow_amd as ff
    import __graft_entry__ as Gm

    def build(Z, zero):
        gs = Gm._model(dev, 2, 2, Z)
        if zero:
            for p in gs.parameters():
                torch.nn.init.zeros_(p)
        return ff.GSVMC(2, 2, ff.HO3D(), ff.FreeFermion(device=dev), gs.cnf, ff.CoulombPairPotential(Z), sp_potential=ff.HO())
    model = build(0.0, True)
    torch.manual_seed(4)
    g = model(8192)
    g.backward()
    assert model.x.shape == (8192, 4, 3)
    assert (model.Eloc - 8.0).abs().max().item() < 1e-7 and abs(model.E - 8.0) < 1e-9
    model = build(2.0, False)
    torch.manual_seed(5)
    g = model(4096)
    g.backward()
    v = model.cnf.v_wrapper.v
    net = O.Net(tuple(N(t) for t in (v.eta.fc1.weight, v.eta.fc1.bias, v.eta.fc2.weight)),
                tuple(N(t) for t in (v.mu.fc1.weight, v.mu.fc1.bias, v.mu.fc2.weight)))
    ref = O.eloc3d(N(model.x[:48]), 2, 2, net, 2.0, rtol=1e-10, atol=1e-12)
    rel = np.abs(N(model.Eloc[:48]) - ref["eloc"]) / np.abs(ref["eloc"])
    assert rel.max() < ELOC_RTOL, rel.max()
    assert np.isfinite(model.E) and all(torch.isfinite(p.grad).all() for p in model.parameters())


def test_autograd_through_backflow_and_mlp(dev):
    """The reference's own checks of these two modules run against the drop-in (VERDICT r03 missing #5):
    tests/test_equivariant_funs.py:5-35 -- permutation equivariance and hand-derived divergence == autograd divergence of
    Backflow.forward; tests/test_MLP.py:18-44 -- MLP(15, 40): autograd gradient of forward == MLP.grad, one and two batch
    dimensions.  The checker is the plain torch restatement on the same weights (fc2(sigmoid(fc1(.))) through the modules' own
    nn.Linear layers); the product's forward and backward are the native kernels (ff_backflow_v_div / ff_backflow_vjp /
    ff_mlp_eval_nd)."""
    import fermiflow_amd as ff
    torch.manual_seed(3)
    eta, mu = ff.MLP(1, 100).to(dev), ff.MLP(1, 200).to(dev)
    v = ff.Backflow(eta, mu=mu)
    batch, n, dim = 300, 10, 3
    x = torch.randn(batch, n, dim, dtype=torch.float64, device=dev, requires_grad=True)
    out = v(x)
    assert out.shape == (batch, n, dim) and out.requires_grad
    P = torch.randperm(n)
    assert torch.allclose(v(x[:, P, :]), out[:, P, :])

    def plain(m, r):
        return m.fc2(torch.sigmoid(m.fc1(r)))

    def v_plain(xx):      # src/equivariant_funs.py:17-62 restated with torch ops
        rij = xx[:, :, None] - xx[:, None]
        dij = (rij + torch.eye(n, dtype=xx.dtype, device=dev)[..., None]).norm(dim=-1, keepdim=True)
        ee = ((plain(eta, dij) * rij) * (1 - torch.eye(n, dtype=xx.dtype, device=dev))[..., None]).sum(dim=-2)
        return ee + plain(mu, xx.norm(dim=-1, keepdim=True)) * xx
    xr = x.detach().clone().requires_grad_(True)
    outr = v_plain(xr)
    assert torch.allclose(out, outr, atol=1e-12)
    w = torch.randn_like(out)
    gx, = torch.autograd.grad(out, x, grad_outputs=w, retain_graph=True)
    gxr, = torch.autograd.grad(outr, xr, grad_outputs=w, retain_graph=True)
    assert torch.allclose(gx, gxr, atol=1e-11), (gx - gxr).abs().max()
    # the reference's utils.divergence (src/utils.py:4-21): sum_i d v_i / d x_i by autograd, against the hand-derived formula
    xf = x.flatten(start_dim=1)
    yf = v(xf.view_as(x)).flatten(start_dim=1)
    ones = torch.ones(batch, dtype=torch.float64, device=dev)
    div = sum(torch.autograd.grad(yf[:, i], xf, grad_outputs=ones, retain_graph=True)[0][:, i] for i in range(n * dim))
    div_direct = v.divergence(x)
    assert div_direct.shape == (batch,) and torch.allclose(div, div_direct, atol=1e-11)
    # ... and the divergence itself is differentiable with respect to x
    gd, = torch.autograd.grad(div_direct.sum(), x)
    divr = sum(torch.autograd.grad(outr.flatten(start_dim=1)[:, i].sum(), xr, create_graph=True)[0].flatten(start_dim=1)[:, i] for i in range(n * dim))
    gdr, = torch.autograd.grad(divr.sum(), xr)
    assert torch.allclose(gd, gdr, atol=1e-9), (gd - gdr).abs().max()
    # MLP of any input dimension
    mlp = ff.MLP(15, 40).to(dev)
    for shape in ((100,), (46, 87)):
        xm = torch.randn(*shape, 15, dtype=torch.float64, device=dev, requires_grad=True)
        y = mlp(xm)
        assert y.shape == (*shape, 1) and torch.allclose(y, plain(mlp, xm), atol=1e-13)
        g_auto, = torch.autograd.grad(y, xm, grad_outputs=torch.ones_like(y))
        g_direct = mlp.grad(xm)
        g_plain, = torch.autograd.grad(plain(mlp, xm), xm, grad_outputs=torch.ones_like(y))
        assert g_auto.shape == (*shape, 15) and torch.allclose(g_auto, g_direct, atol=1e-13) and torch.allclose(g_auto, g_plain, atol=1e-13)
