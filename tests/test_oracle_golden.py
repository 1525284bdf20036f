"""The oracle (oracle/ff_oracle.c) against the golden vectors produced by the REFERENCE itself
(tests/golden/make_golden.py) and against the reference's known-answer tests.  CPU only."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.common import mcmc_noise_from_seed, net_arrays, cnf_param_grads, gsvmc_param_grads


def test_orbitals_and_slater(golden):
    G = golden["g2_slater"]
    v = O.orbitals(np.arange(36), G["orb_pts"])
    np.testing.assert_allclose(v, G["orb_vals"], rtol=1e-14, atol=1e-16)
    for n in (1, 3, 5, 6, 10):
        lp, g, lap = O.logprob(G[f"n{n}_x"], n, 0, tab_up=G[f"n{n}_orb"])
        np.testing.assert_allclose(lp / 2, G[f"n{n}_logabsdet"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(g / 2, G[f"n{n}_grad"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(lap / 2, G[f"n{n}_lap"], rtol=1e-9, atol=1e-7)
    lp, g, lap = O.logprob(G["lp_x"], 3, 6, tab_up=G["lp_up"], tab_dn=G["lp_dn"])
    np.testing.assert_allclose(lp, G["lp_logp"], atol=1e-11)
    np.testing.assert_allclose(g, G["lp_grad"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(lap, G["lp_lap"], rtol=1e-9, atol=1e-6)
    ws = np.repeat(G["ms_keys"], G["ms_counts"])
    lp, g, lap = O.logprob(G["ms_x"], 3, 0, tab_up=G["ms_states"], wstate=ws)
    np.testing.assert_allclose(lp, G["ms_logp"], atol=1e-12)
    np.testing.assert_allclose(g, G["ms_grad"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(lap, G["ms_lap"], rtol=1e-9, atol=1e-8)


def test_known_answer_eigenfunctions():
    """reference tests/test_basedist.py:5-60: Slater determinants of HO2D orbitals are eigenfunctions,
    E_loc == sum of orbital energies at random points."""
    rng = np.random.RandomState(0)
    Es = np.array([n + 1 for n in range(8) for _ in range(n + 1)])
    for nup, ndn in ((5, 0), (3, 6), (1, 0), (10, 0)):
        iu = np.sort(rng.choice(21, nup, replace=False)); idn = np.sort(rng.choice(21, ndn, replace=False))
        x = rng.randn(20, nup + ndn, 2)
        lp, g, lap = O.logprob(x, nup, ndn, tab_up=iu, tab_dn=idn if ndn else None)
        eloc = -0.25 * lap - 0.125 * (g ** 2).sum(axis=(1, 2)) + 0.5 * (x ** 2).sum(axis=(1, 2))
        np.testing.assert_allclose(eloc, Es[iu].sum() + Es[idn].sum(), rtol=1e-8)


@pytest.mark.parametrize("name", ["u3d3", "u6d0", "u6d6", "u1d0", "u10d0"])
def test_mcmc_bit_exact(golden, name):
    """FreeFermion.sample: accept masks and final walkers bit-identical to the reference."""
    G = golden["g1_mcmc"]
    nup, ndn, g0, g, u, accept = mcmc_noise_from_seed(G, name)
    x, logp, acc = O.mcmc_noise(g0, g, u, nup, ndn)
    assert (acc == accept).all()
    assert (x == G[name + "_x"]).all()
    np.testing.assert_allclose(logp, G[name + "_logp"], atol=1e-13)


def test_mcmc_self_contained_fixture(golden):
    G = golden["g1_mcmc"]
    x, logp, acc = O.mcmc_noise(G["u3d3_g0"], G["u3d3_g10"], G["u3d3_u10"], 3, 3)
    accept = np.unpackbits(G["u3d3_accept"])[:100 * 64].reshape(100, 64)[:10]
    assert (acc == accept).all()


def test_backflow_potentials(golden):
    G = golden["g3_backflow"]
    for k in range(int(G["ncase"])):
        n, d, He, Hm = G[f"c{k}_cfg"]
        eta, mu = net_arrays(G, f"c{k}_", Hm > 0)
        v, div = O.backflow(G[f"c{k}_x"], O.Net(eta, mu))
        np.testing.assert_allclose(v, G[f"c{k}_v"], rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(div, G[f"c{k}_div"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(O.potential(G[f"c{k}_x"], 2.0), G[f"c{k}_Vho"] + G[f"c{k}_Vc"], rtol=1e-13)


@pytest.mark.parametrize("tag,rt,at", [("tol6", 1e-6, 1e-8), ("tol10", 1e-10, 1e-12)])
def test_cnf_same_step_sequence_as_reference(golden, tag, rt, at):
    """generate / delta_logp / adjoint with the reference's batch-global scipy-RK45 control: agreement is at
    rounding level (same nfev), not merely at tolerance level."""
    G = golden["g4_cnf"]
    net = O.Net(*net_arrays(G, ""))
    x, _ = O.cnf_generate(G[tag + "_z"], net, rtol=rt, atol=at)
    np.testing.assert_allclose(x, G[tag + "_x"], atol=1e-13)
    z, dl, _ = O.cnf_delta_logp(G[tag + "_x"], net, rtol=rt, atol=at)
    np.testing.assert_allclose(z, G[tag + "_zback"], atol=1e-13)
    np.testing.assert_allclose(dl, G[tag + "_dlogp"], atol=1e-13)
    gx, gp, _ = O.cnf_adjoint(G[tag + "_zback"], G[tag + "_dlogp"], G[tag + "_cz"], G[tag + "_cd"], net, rtol=rt, atol=at)
    np.testing.assert_allclose(gx, G[tag + "_gx"], atol=1e-12)
    ref = cnf_param_grads(G, tag)
    np.testing.assert_allclose(gp, ref, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("name", ["z0_zero", "z2_zero", "z05_nt", "z2_nt", "u6_nt", "z2_nomu", "u6d6_nt"])
def test_local_energy(golden, name):
    """E_loc, logp, grad logp, laplacian logp per walker vs the reference's nested-adjoint autograd."""
    G = golden["g5_gsvmc"]
    nup, ndn, B, seed = (int(v) for v in G[name + "_cfg"])
    use_mu = bool(G[name + "_use_mu"])
    net = O.Net(*net_arrays(G, name + "_", use_mu))
    r = O.eloc(G[name + "_x"], nup, ndn, net, float(G[name + "_Z"]), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(r["eloc"], G[name + "_Eloc"], rtol=1e-9)
    np.testing.assert_allclose(r["logp"], G[name + "_logp"], atol=1e-9)
    np.testing.assert_allclose(r["grad"], G[name + "_grad"], atol=1e-8)
    np.testing.assert_allclose(r["lap"], G[name + "_lap"], rtol=1e-9, atol=1e-7)
    # reference default tolerance: still far inside the 1e-5 bar
    r6 = O.eloc(G[name + "_x"], nup, ndn, net, float(G[name + "_Z"]), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(r6["eloc"], G[name + "_Eloc"], rtol=1e-7)


def test_gsvmc_gradient(golden):
    """theta-gradient of the surrogate loss (src/VMC.py:58 + FermionHO2D.py:71) from the golden walkers."""
    G = golden["g5_gsvmc"]
    name = "z2_nt"
    net = O.Net(*net_arrays(G, name + "_"))
    x, el = G[name + "_x"], G[name + "_Eloc"]
    B = len(el)
    z, dl, _ = O.cnf_delta_logp(x, net, rtol=1e-10, atol=1e-12)
    lp0, g0, _ = O.logprob(z, 3, 3)
    w = (el - el.mean()) / B
    _, gp, _ = O.cnf_adjoint(z, dl, w[:, None, None] * g0, -w, net, rtol=1e-10, atol=1e-12)
    ref = gsvmc_param_grads(G, name)
    np.testing.assert_allclose(gp, ref, atol=1e-8 * np.abs(ref).max())
    np.testing.assert_allclose(((lp0 - dl) * w).sum(), float(G[name + "_gradE"]), rtol=1e-8)


def test_betavmc_walkers(golden):
    G = golden["g6_betavmc"]
    net = O.Net(*net_arrays(G, ""))
    for tag in ("boltz", "hot"):
        ws = np.repeat(G[tag + "_keys"], G[tag + "_counts"])
        r = O.eloc(G[tag + "_x"], 3, 0, net, 2.0, rtol=1e-10, atol=1e-12, tab_up=G[tag + "_states"], wstate=ws)
        np.testing.assert_allclose(r["eloc"], G[tag + "_Eloc"], rtol=1e-9)
        np.testing.assert_allclose(r["eloc"].mean(), float(G[tag + "_E"]), rtol=1e-10)


def test_mcmc_b512_anchor(golden):
    """BASELINE.md 2 / SURVEY 8c anchor: seed 7, 512 walkers, (3,3): final-x SHA-256 d7799a21de62365d, acceptance 0.7527."""
    import hashlib
    G = golden["g1_mcmc"]
    nup, ndn, g0, g, u, accept = mcmc_noise_from_seed(G, "u3d3_b512")
    x, logp, acc = O.mcmc_noise(g0, g, u, nup, ndn)
    assert (acc == accept).all() and (x == G["u3d3_b512_x"]).all()
    assert hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest().startswith("d7799a21de62365d")
    assert abs(acc.mean() - 0.7527) < 5e-5


def betavmc_estimators(eloc, logp, ws, logits, beta):
    """src/VMC.py:146-171 restated on given per-walker E_loc / logp: returns E, E_std, F, F_std, S, gphi, gtheta, the
    gradient of gphi wrt the state logits and the per-walker weights of the theta-gradient."""
    B = len(eloc)
    m = logits.max()
    lsm = logits - (m + np.log(np.exp(logits - m).sum()))
    lps = lsm[ws]
    floc = eloc + lps / beta
    E, F = eloc.mean(), floc.mean()
    cF = (floc - F) / B
    gphi = (lps * cF).sum()
    onehot = np.zeros((B, len(logits))); onehot[np.arange(B), ws] = 1.0
    g_logits = (cF[:, None] * (onehot - np.exp(lsm)[None, :])).sum(axis=0)
    base = np.array([eloc[ws == s].mean() for s in ws])
    w = (eloc - base) / B
    return dict(E=E, E_std=eloc.std(ddof=1), F=F, F_std=floc.std(ddof=1), S=-lps.mean(), gphi=gphi,
                gtheta=(logp * w).sum(), g_logits=g_logits, w=w)


BETA_PG = ["cnf.v_wrapper.v.eta.fc1.weight", "cnf.v_wrapper.v.eta.fc1.bias", "cnf.v_wrapper.v.eta.fc2.weight",
           "cnf.v_wrapper.v.mu.fc1.weight", "cnf.v_wrapper.v.mu.fc1.bias", "cnf.v_wrapper.v.mu.fc2.weight"]


@pytest.mark.parametrize("tag", ["boltz", "hot", "rand"])
def test_betavmc_estimators_and_gradients(golden, tag):
    """F, F_std, S, gradF_phi, gradF_theta (per-state baseline), d gradF_phi / d logits and the six theta-gradients of
    BetaVMC.forward + backward (src/VMC.py:146-171, src/BetaFermionHO2D.py:72-79) from the reference's walkers."""
    G = golden["g6_betavmc"]
    net = O.Net(*net_arrays(G, ""))
    ws = np.repeat(G[tag + "_keys"], G[tag + "_counts"])
    beta, logits = float(G[tag + "_beta"]), G[tag + "_logits"]
    r = O.eloc(G[tag + "_x"], 3, 0, net, 2.0, rtol=1e-10, atol=1e-12, tab_up=G[tag + "_states"], wstate=ws)
    e = betavmc_estimators(r["eloc"], r["logp"], ws, logits, beta)
    for k in ("E", "E_std", "F", "F_std", "S"):
        np.testing.assert_allclose(e[k], float(G[f"{tag}_{k}"]), rtol=1e-9, err_msg=k)
    np.testing.assert_allclose(e["gphi"], float(G[tag + "_gphi"]), rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(e["gtheta"], float(G[tag + "_gtheta"]), rtol=1e-8, atol=1e-12)
    ref = G[tag + "_pg_log_state_weights"]
    np.testing.assert_allclose(e["g_logits"], ref, atol=1e-11 * max(1.0, np.abs(ref).max()))
    z, dl, _ = O.cnf_delta_logp(G[tag + "_x"], net, rtol=1e-10, atol=1e-12)
    _, g0, _ = O.logprob(z, 3, 0, tab_up=G[tag + "_states"], wstate=ws)
    _, gp, _ = O.cnf_adjoint(z, dl, e["w"][:, None, None] * g0, -e["w"], net, rtol=1e-10, atol=1e-12)
    ref = np.concatenate([G[f"{tag}_pg_{k}"] for k in BETA_PG])
    np.testing.assert_allclose(gp, ref, atol=1e-8 * np.abs(ref).max())


def test_gsvmc_estimator(golden):
    """E, E_std, gradE of GSVMC.forward (src/VMC.py:57-58) from the oracle's local energies of the reference's walkers."""
    G = golden["g5_gsvmc"]
    for name in ("z2_nt", "z05_nt", "u6_nt"):
        nup, ndn, B, seed = (int(v) for v in G[name + "_cfg"])
        net = O.Net(*net_arrays(G, name + "_", bool(G[name + "_use_mu"])))
        r = O.eloc(G[name + "_x"], nup, ndn, net, float(G[name + "_Z"]), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(r["eloc"].mean(), float(G[name + "_E"]), rtol=1e-10)
        np.testing.assert_allclose(r["eloc"].std(ddof=1), float(G[name + "_E_std"]), rtol=1e-8)
        np.testing.assert_allclose((r["logp"] * (r["eloc"] - r["eloc"].mean())).mean(), float(G[name + "_gradE"]), rtol=1e-7, atol=1e-12)


def test_ho3d_known_answer_eigenfunctions():
    """The oracle's HO3D orbitals: orthonormal on a quadrature grid, and Slater determinants of them are eigenfunctions --
    E_loc == sum of orbital energies (shell + 3/2) at random points, the reference's tests/test_basedist.py:5-60 one dimension
    up.  (Since round 5 the d = 3 oracle is ALSO pinned to reference-derived goldens: the g7 tests below.)"""
    E3 = np.array([s + 1.5 for s in range(8) for nx in range(s + 1) for ny in range(s + 1 - nx)])
    ax = np.linspace(-6, 6, 41)
    pts = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    v = O.orbitals3d(np.arange(10), pts)
    np.testing.assert_allclose((v @ v.T) * (12 / 40) ** 3, np.eye(10), atol=1e-12)
    rng = np.random.RandomState(0)
    for nup, ndn in ((1, 0), (4, 0), (3, 6), (10, 10)):
        iu = np.sort(rng.choice(20, nup, replace=False)); idn = np.sort(rng.choice(20, ndn, replace=False)) if ndn else None
        x = rng.randn(8, nup + ndn, 3)
        lp, g, lap = O.logprob3d(x, nup, ndn, tab_up=iu, tab_dn=idn)
        eloc = -0.25 * lap - 0.125 * (g ** 2).sum(axis=(1, 2)) + 0.5 * (x ** 2).sum(axis=(1, 2))
        np.testing.assert_allclose(eloc, E3[iu].sum() + (E3[idn].sum() if ndn else 0.0), rtol=1e-10)


# ---- d = 3 pinned to the REFERENCE (tests/golden/g7_3d.npz; make_golden.py group d3): the reference's dimension-generic code
# ---- (LogAbsSlaterDet, FreeFermion.log_prob, y_grad_laplacian, the Metropolis loop body, Backflow, CNF, GSVMC.logp) run on
# ---- (B, n, 3) walkers with 3-D orbital closures that are products of the reference's own HO2D closures.

def test_3d_orbitals_and_slater_vs_reference(golden):
    G = golden["g7_3d"]
    nxyz = np.array([(nx, ny, s - nx - ny) for s in range(8) for nx in range(s + 1) for ny in range(s + 1 - nx)])
    assert (G["nxyz"] == nxyz).all()                       # same list order as fermiflow_amd.orbitals.HO3D / the oracle
    np.testing.assert_allclose(O.orbitals3d(np.arange(35), G["orb_pts"]), G["orb_vals"], rtol=1e-13, atol=1e-16)
    for n in (1, 4, 10):
        lp, g, lap = O.logprob3d(G[f"n{n}_x"], n, 0, tab_up=G[f"n{n}_orb"])
        np.testing.assert_allclose(lp / 2, G[f"n{n}_logabsdet"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(g / 2, G[f"n{n}_grad"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(lap / 2, G[f"n{n}_lap"], rtol=1e-9, atol=1e-6)
    for tag in ("lp36", "lp1010"):
        iu, idn = G[tag + "_up"], G[tag + "_dn"]
        lp, g, lap = O.logprob3d(G[tag + "_x"], len(iu), len(idn), tab_up=iu, tab_dn=idn)
        np.testing.assert_allclose(lp, G[tag + "_logp"], atol=1e-11)
        np.testing.assert_allclose(g, G[tag + "_grad"], rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(lap, G[tag + "_lap"], rtol=1e-9, atol=1e-5)


@pytest.mark.parametrize("name", ["m2d2", "m10d10", "m4d3"])
def test_3d_mcmc_bit_exact_vs_reference(golden, name):
    """The loop body of FreeFermion.sample (src/base_dist.py:63-70) on (B, n, 3) walkers: accept masks and final walkers
    bit-identical to the reference's."""
    G = golden["g7_3d"]
    nup, ndn, g0, g, u, accept = mcmc_noise_from_seed(G, name, dim=3)
    x, logp, acc = O.mcmc_noise3d(g0, g, u, nup, ndn)
    assert (acc == accept).all()
    assert (x == G[name + "_x"]).all()
    np.testing.assert_allclose(logp, G[name + "_logp"], atol=1e-12)


@pytest.mark.parametrize("name", ["e2d2", "e5d4", "e1d1"])
def test_3d_local_energy_and_gradient_vs_reference(golden, name):
    """E_loc, logp, grad, Laplacian per walker, E, E_std, gradE and the six parameter gradients of a GSVMC iteration on 3-D
    walkers (src/VMC.py:46-58 through y_grad_laplacian's nested adjoints) -- O.eloc3d and the d-generic flow / adjoint."""
    G = golden["g7_3d"]
    nup, ndn, B, seed = (int(v) for v in G[name + "_cfg"])
    Z = float(G[name + "_Z"])
    net = O.Net(*net_arrays(G, ""))
    x = G[name + "_x"]
    xg, _ = O.cnf_generate(G[name + "_z"], net, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(xg, x, atol=1e-12)
    r = O.eloc3d(x, nup, ndn, net, Z, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(r["eloc"], G[name + "_Eloc"], rtol=1e-9)
    np.testing.assert_allclose(r["logp"], G[name + "_logp"], atol=1e-9)
    np.testing.assert_allclose(r["grad"], G[name + "_grad"], atol=1e-8)
    np.testing.assert_allclose(r["lap"], G[name + "_lap"], rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(r["V"], G[name + "_V"], rtol=1e-13)
    np.testing.assert_allclose(r["eloc"].mean(), float(G[name + "_E"]), rtol=1e-10)
    np.testing.assert_allclose(r["eloc"].std(ddof=1), float(G[name + "_E_std"]), rtol=1e-8)
    z, dl, _ = O.cnf_delta_logp(x, net, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(z, G[name + "_zback"], atol=1e-12)
    np.testing.assert_allclose(dl, G[name + "_dlogp"], atol=1e-12)
    lp0, g0, _ = O.logprob3d(z, nup, ndn)
    el = G[name + "_Eloc"]
    w = (el - el.mean()) / B
    np.testing.assert_allclose(((lp0 - dl) * w).sum(), float(G[name + "_gradE"]), rtol=1e-8, atol=1e-12)
    _, gp, _ = O.cnf_adjoint(z, dl, w[:, None, None] * g0, -w, net, rtol=1e-10, atol=1e-12)
    ref = gsvmc_param_grads(G, name)
    np.testing.assert_allclose(gp, ref, atol=1e-8 * np.abs(ref).max())
