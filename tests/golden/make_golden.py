#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz by running the REFERENCE (imported from
/root/reference through ref_shim.py) on seeded inputs.

Run in the build container only:   python tests/golden/make_golden.py [group ...]
Groups: mcmc slater backflow cnf gsvmc betavmc d3   (default: all)

Every fixture stores inputs and the reference's outputs; no reference source text is stored.
Reference call sites are cited next to each group.
"""
import sys, os, time, hashlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def orbital_indices(ho2d, orbs):
    """closures -> integer index k into HO2D().orbitals (identity lookup)."""
    return np.array([next(i for i, o in enumerate(ho2d.orbitals) if o is f) for f in orbs], dtype=np.int32)


def nontrivial_weights(R, H_eta=50, H_mu=50, seeds=(1, 2)):
    """init_gaussian(seed) then fc2 x30, fc1.weight x300 -> O(0.1-1) backflow."""
    torch = R.torch
    eta = R.MLP.MLP(1, H_eta); eta.init_gaussian(seeds[0])
    mu = R.MLP.MLP(1, H_mu); mu.init_gaussian(seeds[1])
    with torch.no_grad():
        for m in (eta, mu):
            m.fc2.weight *= 30.0
            m.fc1.weight *= 300.0
    return eta, mu


def mlp_np(m):
    return (m.fc1.weight.detach().numpy().reshape(-1).copy(),
            m.fc1.bias.detach().numpy().copy(),
            m.fc2.weight.detach().numpy().reshape(-1).copy())


# ---------------------------------------------------------------------------------------------
def g_mcmc(R):
    """FreeFermion.sample, src/base_dist.py:58-71.  RNG draw order: randn(B,n,2); per step
    randn_like(x) then rand_like(p)."""
    torch = R.torch
    ho = R.orbitals.HO2D()
    bd = R.base_dist.FreeFermion()
    out = {}
    for name, (nup, ndown, B, seed, steps) in {
            "u3d3": (3, 3, 64, 7, 100), "u6d0": (6, 0, 64, 11, 100), "u6d6": (6, 6, 32, 13, 100),
            "u1d0": (1, 0, 16, 3, 50), "u10d0": (10, 0, 16, 5, 50),
            "u3d3_b512": (3, 3, 512, 7, 100)}.items():   # SURVEY 8c / BASELINE.md 2 anchor: final-x sha d7799a21de62365d, acceptance 0.7527
        up, dn = ho.orbitals[:nup], ho.orbitals[:ndown]
        torch.manual_seed(seed)
        x_ref = bd.sample(up, dn, (B,), equilibrim_steps=steps)
        # replay to capture noise + per-step accept masks
        torch.manual_seed(seed)
        n = nup + ndown
        g0 = torch.randn(B, n, 2)
        x = g0.clone()
        logp = bd.log_prob(up, dn, x)
        logp0 = logp.clone()
        gs, us, acc = [], [], []
        for _ in range(steps):
            g = torch.randn_like(x)
            new_x = x + 0.1 * g
            new_logp = bd.log_prob(up, dn, new_x)
            p = torch.exp(new_logp - logp)
            u = torch.rand_like(p)
            a = u < p
            x[a] = new_x[a]; logp[a] = new_logp[a]
            gs.append(g.numpy().copy()); us.append(u.numpy().copy()); acc.append(a.numpy().copy())
        assert torch.equal(x, x_ref)
        gs, us, acc = np.stack(gs), np.stack(us), np.stack(acc)
        out[name + "_cfg"] = np.array([nup, ndown, B, seed, steps], dtype=np.int64)
        out[name + "_noise_sha"] = np.array(sha(g0.numpy()) + sha(gs) + sha(us))
        out[name + "_logp0"] = logp0.numpy()
        out[name + "_accept"] = np.packbits(acc.astype(np.uint8), axis=None)
        out[name + "_x"] = x.numpy().copy()
        out[name + "_logp"] = logp.numpy().copy()
        if name in ("u3d3",):  # one fully self-contained case (first 10 steps of noise kept)
            out[name + "_g0"] = g0.numpy()
            out[name + "_g10"] = gs[:10]
            out[name + "_u10"] = us[:10]
        out[name + "_x_sha"] = np.array(sha(x.numpy()))
        print("mcmc", name, "acc-rate", acc.mean(), "x sha", sha(x.numpy())[:16])
        if name == "u3d3_b512":
            assert sha(x.numpy()).startswith("d7799a21de62365d") and abs(acc.mean() - 0.7527) < 5e-5, "BASELINE.md anchor moved"
    np.savez_compressed(os.path.join(HERE, "g1_mcmc.npz"), **out)


# ---------------------------------------------------------------------------------------------
def g_slater(R):
    """LogAbsSlaterDet fwd/bwd (src/slater.py:13-62), MultStates (:85-155), log_prob
    (src/base_dist.py:49-56); grad + Laplacian through utils.y_grad_laplacian (src/utils.py:40-65)."""
    torch = R.torch
    ho = R.orbitals.HO2D()
    rng = np.random.RandomState(123)
    out = {}
    for n in (1, 3, 5, 6, 10):
        idx = np.sort(rng.choice(36 if n < 10 else 21, size=n, replace=False)).astype(np.int32)
        if n == 3:
            idx = np.array([0, 1, 2], dtype=np.int32)
        orbs = tuple(ho.orbitals[i] for i in idx)
        torch.manual_seed(100 + n)
        x = torch.randn(24, n, 2, requires_grad=True)
        y, g, lap = R.utils.y_grad_laplacian(lambda x: R.slater.LogAbsSlaterDet.apply(orbs, x), x)
        out[f"n{n}_orb"] = idx
        out[f"n{n}_x"] = x.detach().numpy().copy()
        out[f"n{n}_logabsdet"] = y.detach().numpy()
        out[f"n{n}_grad"] = g.detach().numpy()
        out[f"n{n}_lap"] = lap.detach().numpy()
        E = sum(ho.Es[i] for i in idx)
        eloc = -0.5 * lap - 0.5 * (g ** 2).sum(dim=(-2, -1)) + 0.5 * (x ** 2).sum(dim=(-2, -1))
        assert torch.allclose(eloc, E * torch.ones(24)), (n, eloc)
    # orbital values on a grid of points (all 36)
    torch.manual_seed(5)
    pts = torch.randn(40, 2) * 1.5
    out["orb_pts"] = pts.numpy()
    out["orb_vals"] = np.stack([o(pts).numpy() for o in ho.orbitals])
    out["orb_Es"] = np.array(ho.Es, dtype=np.int64)
    # full log_prob with two spins + multstates
    bd = R.base_dist.FreeFermion()
    iu = np.array([0, 2, 5], dtype=np.int32); idn = np.array([0, 1, 3, 4, 7, 9], dtype=np.int32)
    up = tuple(ho.orbitals[i] for i in iu); dn = tuple(ho.orbitals[i] for i in idn)
    torch.manual_seed(77)
    x = torch.randn(20, 9, 2, requires_grad=True)
    y, g, lap = R.utils.y_grad_laplacian(lambda x: bd.log_prob(up, dn, x), x)
    out["lp_up"], out["lp_dn"] = iu, idn
    out["lp_x"] = x.detach().numpy().copy(); out["lp_logp"] = y.detach().numpy()
    out["lp_grad"] = g.detach().numpy(); out["lp_lap"] = lap.detach().numpy()
    # multstates, polarized (ndown = 0), fermion_states(3, 0, 2.0) -> 21 states
    states, Es = ho.fermion_states(3, 0, 2.0)
    st_idx = np.stack([orbital_indices(ho, s[0]) for s in states])
    counts = {0: 5, 3: 2, 4: 4, 9: 1, 20: 3}
    Bm = sum(counts.values())
    torch.manual_seed(78)
    x = torch.randn(Bm, 3, 2, requires_grad=True)
    y, g, lap = R.utils.y_grad_laplacian(lambda x: bd.log_prob_multstates(states, counts, x), x)
    out["ms_states"] = st_idx; out["ms_Es"] = np.array(Es, dtype=np.int64)
    out["ms_keys"] = np.array(list(counts.keys()), dtype=np.int64)
    out["ms_counts"] = np.array(list(counts.values()), dtype=np.int64)
    out["ms_x"] = x.detach().numpy().copy(); out["ms_logp"] = y.detach().numpy()
    out["ms_grad"] = g.detach().numpy(); out["ms_lap"] = lap.detach().numpy()
    # state enumeration sizes (src/orbitals.py:14-54)
    for N, dE in ((3, 2), (4, 2), (6, 4), (10, 3)):
        for de in range(dE + 1):
            s, e = ho.fermion_states(N, 0, de)
            out[f"enum_N{N}_dE{de}_E"] = np.array(e, dtype=np.int64)
            out[f"enum_N{N}_dE{de}_idx"] = np.stack([orbital_indices(ho, t[0]) for t in s])
    np.savez_compressed(os.path.join(HERE, "g2_slater.npz"), **out)
    print("slater done")


# ---------------------------------------------------------------------------------------------
def g_backflow(R):
    """Backflow.forward/divergence (src/equivariant_funs.py:83-102), MLP.forward/grad
    (src/MLP.py:30-45), potentials (src/potentials.py:13-47)."""
    torch = R.torch
    out = {}
    k = 0
    for (n, d, He, Hm, use_mu) in ((4, 2, 50, 50, True), (6, 2, 50, 50, True), (6, 2, 50, 0, False),
                                  (10, 3, 100, 200, True), (12, 2, 50, 50, True), (3, 2, 50, 50, True)):
        eta, mu = nontrivial_weights(R, He, max(Hm, 1), seeds=(10 + k, 20 + k))
        v = R.equivariant_funs.Backflow(eta, mu=mu if use_mu else None)
        torch.manual_seed(300 + k)
        x = torch.randn(12, n, d)
        out[f"c{k}_cfg"] = np.array([n, d, He, Hm if use_mu else 0], dtype=np.int64)
        out[f"c{k}_x"] = x.numpy().copy()
        out[f"c{k}_v"] = v(x).detach().numpy()
        out[f"c{k}_div"] = v.divergence(x).detach().numpy()
        for nm, m in (("eta", eta), ("mu", mu)):
            w1, b1, w2 = mlp_np(m)
            out[f"c{k}_{nm}_w1"], out[f"c{k}_{nm}_b1"], out[f"c{k}_{nm}_w2"] = w1, b1, w2
        if k == 0:
            r = torch.linspace(0.0, 4.0, 33)[:, None]
            out["mlp_r"] = r.numpy().reshape(-1)
            out["mlp_eta"] = eta(r).detach().numpy().reshape(-1)
            out["mlp_deta"] = eta.grad(r).detach().numpy().reshape(-1)
        out[f"c{k}_Vho"] = R.potentials.HO().V(x).numpy()
        out[f"c{k}_Vc"] = R.potentials.CoulombPairPotential(2.0).V(x).numpy()
        k += 1
    out["ncase"] = np.array(k)
    np.savez_compressed(os.path.join(HERE, "g3_backflow.npz"), **out)
    print("backflow done")


# ---------------------------------------------------------------------------------------------
def g_cnf(R):
    """CNF.generate / CNF.delta_logp (src/flow.py:42-55) through solve_ivp_nnmodule
    (src/NeuralODE/nnModule.py:161-188, scipy RK45 branch :49-61)."""
    out = {}
    for tag, (rtol, atol) in {"tol6": (1e-6, 1e-8), "tol10": (1e-10, 1e-12)}.items():
        Rr = ref_shim.load(rtol, atol)
        torch = Rr.torch
        eta, mu = nontrivial_weights(Rr)
        v = Rr.equivariant_funs.Backflow(eta, mu=mu)
        cnf = Rr.flow.CNF(v, (0.0, 1.0))
        torch.manual_seed(42)
        z = torch.randn(16, 6, 2)
        x = cnf.generate(z)
        zb, dl = cnf.delta_logp(x)
        out[f"{tag}_z"] = z.numpy().copy(); out[f"{tag}_x"] = x.detach().numpy()
        out[f"{tag}_zback"] = zb.detach().numpy(); out[f"{tag}_dlogp"] = dl.detach().numpy()
        # gradient of sum(logp-ish) wrt x and params through the adjoint (nnModule.py:76-99)
        xg = x.detach().clone().requires_grad_(True)
        zb2, dl2 = cnf.delta_logp(xg, params_require_grad=True)
        torch.manual_seed(43)
        cz = torch.randn_like(zb2); cd = torch.randn_like(dl2)
        loss = (cz * zb2).sum() + (cd * dl2).sum()
        grads = torch.autograd.grad(loss, [xg] + list(cnf.parameters()))
        out[f"{tag}_cz"], out[f"{tag}_cd"] = cz.numpy(), cd.numpy()
        out[f"{tag}_gx"] = grads[0].numpy()
        names = [n for n, _ in cnf.named_parameters()]
        for nme, g in zip(names, grads[1:]):
            out[f"{tag}_g_{nme}"] = g.numpy().reshape(-1)
        out[f"{tag}_pnames"] = np.array(names)
        for nm, m in (("eta", eta), ("mu", mu)):
            w1, b1, w2 = mlp_np(m)
            out[f"{nm}_w1"], out[f"{nm}_b1"], out[f"{nm}_w2"] = w1, b1, w2
        print("cnf", tag, "done")
    np.savez_compressed(os.path.join(HERE, "g4_cnf.npz"), **out)


# ---------------------------------------------------------------------------------------------
def gsvmc_case(R, nup, ndown, Z, weights, B, seed, use_mu=True):
    torch = R.torch
    ho = R.orbitals.HO2D()
    bd = R.base_dist.FreeFermion()
    if weights == "zero":
        eta = R.MLP.MLP(1, 50); eta.init_zeros()
        mu = R.MLP.MLP(1, 50); mu.init_zeros()
    else:
        eta, mu = nontrivial_weights(R)
    v = R.equivariant_funs.Backflow(eta, mu=mu if use_mu else None)
    cnf = R.flow.CNF(v, (0.0, 1.0))
    model = R.VMC.GSVMC(nup, ndown, ho, bd, cnf, R.potentials.CoulombPairPotential(Z),
                        sp_potential=R.potentials.HO())
    # --- replicate GSVMC.forward (src/VMC.py:40-59) statement by statement to capture per-walker data
    torch.manual_seed(seed)
    z, x = model.sample((B,))
    x = x.detach().clone().requires_grad_(True)
    logp_full = model.logp(x, params_require_grad=True)
    logp, grad_logp, lap_logp = R.utils.y_grad_laplacian(model.logp, x)
    kinetic = -1 / 4 * lap_logp - 1 / 8 * (grad_logp ** 2).sum(dim=(-2, -1))
    potential = model.pair_potential.V(x) + model.sp_potential.V(x)
    Eloc = (kinetic + potential).detach()
    E, E_std = Eloc.mean().item(), Eloc.std().item()
    gradE = (logp_full * (Eloc - E)).mean()
    model.zero_grad()
    gradE.backward()
    # --- and the real forward from the same seed must agree
    torch.manual_seed(seed)
    gE2 = model(B)
    assert abs(model.E - E) <= 1e-9 * max(1, abs(E)), (model.E, E)
    o = dict(cfg=np.array([nup, ndown, B, seed], dtype=np.int64), Z=np.array(Z),
             z=z.detach().numpy(), x=x.detach().numpy(), logp=logp.detach().numpy(),
             logp_full=logp_full.detach().numpy(),
             grad=grad_logp.detach().numpy(), lap=lap_logp.detach().numpy(),
             V=potential.detach().numpy(), Eloc=Eloc.numpy(), E=np.array(E), E_std=np.array(E_std),
             gradE=np.array(gradE.item()), use_mu=np.array(int(use_mu)))
    for nm, m in (("eta", eta), ("mu", mu)):
        w1, b1, w2 = mlp_np(m)
        o[f"{nm}_w1"], o[f"{nm}_b1"], o[f"{nm}_w2"] = w1, b1, w2
    for nme, p in model.named_parameters():
        o["pg_" + nme] = (p.grad.numpy().reshape(-1).copy() if p.grad is not None else np.zeros(p.numel()))
    return o


def g_gsvmc(R_unused):
    """GSVMC.forward + backward (src/VMC.py:40-59, src/FermionHO2D.py:69-72)."""
    out = {}
    cases = {
        "z0_zero": (3, 3, 0.0, "zero", 16, 0, True),
        "z2_zero": (3, 3, 2.0, "zero", 16, 1, True),
        "z05_nt": (3, 3, 0.5, "nt", 16, 2, True),
        "z2_nt": (3, 3, 2.0, "nt", 32, 3, True),
        "u6_nt": (6, 0, 0.5, "nt", 8, 4, True),
        "z2_nomu": (3, 3, 2.0, "nt", 8, 5, False),
        "u6d6_nt": (6, 6, 2.0, "nt", 4, 6, True),      # BASELINE.json configs[3] shape (n = 12): the reference's own numbers
    }
    only = os.environ.get("FF_GSVMC_CASES")
    for name, (nup, ndown, Z, w, B, seed, use_mu) in cases.items():
        if only and name not in only.split(","):
            continue
        t = time.time()
        R = ref_shim.load(1e-10, 1e-12)
        o = gsvmc_case(R, nup, ndown, Z, w, B, seed, use_mu)
        for k_, v_ in o.items():
            out[f"{name}_{k_}"] = v_
        print("gsvmc", name, "E", o["E"], "E_std", o["E_std"], "%.1fs" % (time.time() - t), flush=True)
    # the same z2_nt walkers at the reference's default tolerance (measures its own ODE error)
    if not only or "z2_nt_tol6" in only:
        R = ref_shim.load(1e-6, 1e-8)
        o = gsvmc_case(R, 3, 3, 2.0, "nt", 32, 3, True)
        for k_ in ("Eloc", "E", "E_std", "gradE", "logp", "lap", "grad", "x"):
            out[f"z2_nt_tol6_{k_}"] = o[k_]
        for k_ in o:
            if k_.startswith("pg_"):
                out[f"z2_nt_tol6_{k_}"] = o[k_]
    out["names"] = np.array(list(cases.keys()))
    path = os.path.join(HERE, "g5_gsvmc.npz")
    if only and os.path.exists(path):
        old = dict(np.load(path)); old.update(out); out = old
    np.savez_compressed(path, **out)


# ---------------------------------------------------------------------------------------------
def g_betavmc(R_unused):
    """BetaVMC.forward + backward (src/VMC.py:61-171, src/BetaFermionHO2D.py:72-79)."""
    R = ref_shim.load(1e-10, 1e-12)
    torch = R.torch
    import io, contextlib
    ho = R.orbitals.HO2D()
    bd = R.base_dist.FreeFermion()
    eta, mu = nontrivial_weights(R)
    v = R.equivariant_funs.Backflow(eta, mu=mu)
    cnf = R.flow.CNF(v, (0.0, 1.0))
    beta, nup, dE, B, seed = 10.0, 3, 2.0, 24, 9
    out = {}
    for tag, boltz, bta in (("boltz", True, 10.0), ("hot", True, 0.5), ("rand", False, 2.0)):
        torch.manual_seed(100)      # "rand": the state logits are torch.randn(Nstates) (src/VMC.py:83); kept in the fixture
        model = R.VMC.BetaVMC(bta, nup, 0, dE, boltz, ho, bd, cnf,
                              R.potentials.CoulombPairPotential(2.0), sp_potential=R.potentials.HO())
        # capture the per-walker data of this very forward (x, E_loc, logp) by wrapping the pieces it calls
        cap = {}
        orig_sample, orig_ygl = model.sample, R.utils.y_grad_laplacian
        def sample_cap(shape, nframes=None):
            z, x = orig_sample(shape, nframes=nframes); cap["z"], cap["x"] = z.detach().clone(), x.detach().clone(); return z, x
        def ygl_cap(f, x):
            y, g, l = orig_ygl(f, x); cap["logp"], cap["grad"], cap["lap"] = y.detach().clone(), g.detach().clone(), l.detach().clone(); return y, g, l
        model.sample = sample_cap; R.utils.y_grad_laplacian = ygl_cap
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            gphi, gtheta = model(B)
        model.sample = orig_sample; R.utils.y_grad_laplacian = orig_ygl
        model.zero_grad()
        (gphi + gtheta).backward()
        for k_, v_ in cap.items():
            out[f"{tag}_{k_}"] = v_.numpy()
        kin = -0.25 * cap["lap"] - 0.125 * (cap["grad"] ** 2).sum(dim=(-2, -1))
        out[f"{tag}_Eloc"] = (kin + model.pair_potential.V(cap["x"]) + model.sp_potential.V(cap["x"])).numpy()
        keys = list(model.state_indices_collection.keys()); cnts = list(model.state_indices_collection.values())
        out[f"{tag}_cfg"] = np.array([nup, B, seed], dtype=np.int64)
        out[f"{tag}_beta"] = np.array(bta); out[f"{tag}_dE"] = np.array(dE)
        out[f"{tag}_keys"] = np.array(keys, dtype=np.int64); out[f"{tag}_counts"] = np.array(cnts, dtype=np.int64)
        out[f"{tag}_Es"] = model.Es_original.numpy()
        out[f"{tag}_logits"] = model.log_state_weights.detach().numpy().copy()
        out[f"{tag}_states"] = np.stack([orbital_indices(ho, s[0]) for s in model.states])
        for k_ in ("E", "E_std", "F", "F_std", "S", "S_analytical"):
            out[f"{tag}_{k_}"] = np.array(getattr(model, k_))
        out[f"{tag}_logp_states_all"] = model.logp_states_all.numpy()
        out[f"{tag}_gphi"] = np.array(gphi.item()); out[f"{tag}_gtheta"] = np.array(gtheta.item())
        for nme, p in model.named_parameters():
            out[f"{tag}_pg_{nme}"] = p.grad.numpy().reshape(-1).copy()
        print("betavmc", tag, "E", model.E, "F", model.F, "S", model.S, flush=True)
    for nm, m in (("eta", eta), ("mu", mu)):
        w1, b1, w2 = mlp_np(m)
        out[f"{nm}_w1"], out[f"{nm}_b1"], out[f"{nm}_w2"] = w1, b1, w2
    np.savez_compressed(os.path.join(HERE, "g6_betavmc.npz"), **out)

# ---------------------------------------------------------------------------------------------
def ho3d_closures(R):
    """3-D oscillator orbitals as Python closures, built HERE from the reference's own HO2D closures (src/orbitals.py:65-82):
    phi3D_{nx,ny,nz}(x, y, z) = phi2D_{nx,ny}(x, y) * [pi^(1/4) * phi2D_{nz,0}(z, 0)]  -- the bracket is the normalised 1-D
    oscillator function pi^(-1/4) e^(-z^2/2) h_nz(z), because phi2D_{n,0}(z, 0) = pi^(-1/2) e^(-z^2/2) h_n(z) h_0(0) and h_0 = 1.
    Every factor is evaluated by reference code; only the product is formed here.  List order = fermiflow_amd.orbitals.HO3D:
    for shell: for nx in 0..shell: for ny in 0..shell-nx: (nx, ny, shell-nx-ny); E = shell + 3/2."""
    torch = R.torch
    ho = R.orbitals.HO2D()
    k2 = lambda nx, ny: (nx + ny) * (nx + ny + 1) // 2 + nx          # index of (nx, ny) in HO2D().orbitals (src/orbitals.py:81)
    c = float(np.pi) ** 0.25

    def make(nx, ny, nz):
        fxy, fz = ho.orbitals[k2(nx, ny)], ho.orbitals[k2(nz, 0)]
        def phi(x):
            z = x[..., 2]
            return fxy(x[..., :2]) * (c * fz(torch.stack((z, torch.zeros_like(z)), dim=-1)))
        return phi
    orbs, Es, nxyz = [], [], []
    for shell in range(8):
        for nx in range(shell + 1):
            for ny in range(shell + 1 - nx):
                orbs.append(make(nx, ny, shell - nx - ny)); Es.append(shell + 1.5); nxyz.append((nx, ny, shell - nx - ny))
    return orbs, Es, nxyz


def replay_sample(R, bd, up, dn, B, dim, seed, steps, tau=0.1):
    """The body of FreeFermion.sample (src/base_dist.py:62-70) statement by statement with randn(B, n, dim) in place of the
    hard-coded randn(B, n, 2) of :62; log_prob is the reference's.  Returns noise, accept masks, walkers."""
    torch = R.torch
    torch.manual_seed(seed)
    n = len(up) + len(dn)
    g0 = torch.randn(B, n, dim)
    x = g0.clone()
    logp = bd.log_prob(up, dn, x)
    logp0 = logp.clone()
    gs, us, acc = [], [], []
    for _ in range(steps):
        g = torch.randn_like(x)
        new_x = x + tau * g
        new_logp = bd.log_prob(up, dn, new_x)
        p = torch.exp(new_logp - logp)
        u = torch.rand_like(p)
        a = u < p
        x[a] = new_x[a]; logp[a] = new_logp[a]
        gs.append(g.numpy().copy()); us.append(u.numpy().copy()); acc.append(a.numpy().copy())
    return g0.numpy(), np.stack(gs), np.stack(us), np.stack(acc), logp0.numpy(), x, logp


def g_3d(R_unused):
    """d = 3 (BASELINE.json configs[4]).  The reference has no 3-D orbital LIST (src/orbitals.py:56) and one
    hard-coded randn shape (src/base_dist.py:62) -- everything else on the path is dimension-generic and is run here AS IS on
    (B, n, 3) walkers: LogAbsSlaterDet.apply (src/slater.py:13-62), FreeFermion.log_prob (src/base_dist.py:49-56),
    y_grad_laplacian (src/utils.py:40-65), the Metropolis loop body (src/base_dist.py:63-70), Backflow, CNF.generate /
    delta_logp, GSVMC.logp (src/VMC.py:35-38) and the estimator lines of GSVMC.forward (src/VMC.py:46-58)."""
    out = {}
    R = ref_shim.load(1e-10, 1e-12)
    torch = R.torch
    orbs, Es, nxyz = ho3d_closures(R)
    out["nxyz"] = np.array(nxyz, dtype=np.int32); out["Es"] = np.array(Es)
    bd = R.base_dist.FreeFermion()
    rng = np.random.RandomState(321)
    # orbital values at random points (first 35 = shells 0..4)
    torch.manual_seed(6)
    pts = torch.randn(40, 3) * 1.3
    out["orb_pts"] = pts.numpy(); out["orb_vals"] = np.stack([o(pts).numpy() for o in orbs[:35]])
    # (i) one determinant: value, gradient, Laplacian
    for n in (1, 4, 10):
        idx = np.arange(10, dtype=np.int32) if n == 10 else np.sort(rng.choice(20, size=n, replace=False)).astype(np.int32)
        sel = tuple(orbs[i] for i in idx)
        torch.manual_seed(200 + n)
        x = torch.randn(16, n, 3, requires_grad=True)
        y, g, lap = R.utils.y_grad_laplacian(lambda x: R.slater.LogAbsSlaterDet.apply(sel, x), x)
        out[f"n{n}_orb"] = idx; out[f"n{n}_x"] = x.detach().numpy().copy()
        out[f"n{n}_logabsdet"] = y.detach().numpy(); out[f"n{n}_grad"] = g.detach().numpy(); out[f"n{n}_lap"] = lap.detach().numpy()
        E = sum(Es[i] for i in idx)
        eloc = -0.5 * lap - 0.5 * (g ** 2).sum(dim=(-2, -1)) + 0.5 * (x ** 2).sum(dim=(-2, -1))
        print("3d det n", n, "max |Eloc - E|", (eloc - E).abs().max().item())
        assert torch.allclose(eloc, E * torch.ones(16), rtol=1e-6), (n, eloc)
    # two spin species through FreeFermion.log_prob: random subsets (3, 6) and configs[4]'s closed shells (10, 10)
    for tag, iu, idn, B in (("lp36", np.sort(rng.choice(20, 3, replace=False)), np.sort(rng.choice(20, 6, replace=False)), 12),
                            ("lp1010", np.arange(10), np.arange(10), 8)):
        up, dn = tuple(orbs[i] for i in iu), tuple(orbs[i] for i in idn)
        torch.manual_seed(88 + len(iu))
        x = torch.randn(B, len(iu) + len(idn), 3, requires_grad=True)
        y, g, lap = R.utils.y_grad_laplacian(lambda x: bd.log_prob(up, dn, x), x)
        out[tag + "_up"], out[tag + "_dn"] = iu.astype(np.int32), idn.astype(np.int32)
        out[tag + "_x"] = x.detach().numpy().copy(); out[tag + "_logp"] = y.detach().numpy()
        out[tag + "_grad"] = g.detach().numpy(); out[tag + "_lap"] = lap.detach().numpy()
    # (ii) Metropolis replay
    for name, (nup, ndown, B, seed, steps) in {"m2d2": (2, 2, 64, 17, 100), "m10d10": (10, 10, 16, 19, 60), "m4d3": (4, 3, 32, 23, 100)}.items():
        up, dn = tuple(orbs[:nup]), tuple(orbs[:ndown])
        g0, gs, us, acc, logp0, x, logp = replay_sample(R, bd, up, dn, B, 3, seed, steps)
        out[name + "_cfg"] = np.array([nup, ndown, B, seed, steps], dtype=np.int64)
        out[name + "_noise_sha"] = np.array(sha(g0) + sha(gs) + sha(us))
        out[name + "_logp0"] = logp0
        out[name + "_accept"] = np.packbits(acc.astype(np.uint8), axis=None)
        out[name + "_x"] = x.numpy().copy(); out[name + "_logp"] = logp.numpy().copy()
        print("3d mcmc", name, "acc-rate", acc.mean(), "x sha", sha(x.numpy())[:16])
    # (iii) local energy, estimator and the six parameter gradients of a GSVMC iteration on 3-D walkers
    only = os.environ.get("FF_3D_CASES")
    for name, (nup, ndown, Z, B, seed) in {"e2d2": (2, 2, 2.0, 12, 31), "e5d4": (5, 4, 2.0, 4, 33), "e1d1": (1, 1, 0.5, 8, 35)}.items():
        if only and name not in only.split(","):
            continue
        t = time.time()
        eta, mu = nontrivial_weights(R)
        v = R.equivariant_funs.Backflow(eta, mu=mu)
        cnf = R.flow.CNF(v, (0.0, 1.0))
        import types
        model = R.VMC.GSVMC(nup, ndown, types.SimpleNamespace(orbitals=orbs), bd, cnf, R.potentials.CoulombPairPotential(Z),
                            sp_potential=R.potentials.HO())
        _, _, _, _, _, z, _ = replay_sample(R, bd, model.orbitals_up, model.orbitals_down, B, 3, seed, 100)
        x = cnf.generate(z)                                             # src/VMC.py:32
        x = x.detach().clone().requires_grad_(True)                     # :44
        logp_full = model.logp(x, params_require_grad=True)             # :46
        logp, grad_logp, lap_logp = R.utils.y_grad_laplacian(model.logp, x)   # :48
        kinetic = -1 / 4 * lap_logp - 1 / 8 * (grad_logp ** 2).sum(dim=(-2, -1))
        potential = model.pair_potential.V(x) + model.sp_potential.V(x)
        Eloc = (kinetic + potential).detach()
        E, E_std = Eloc.mean().item(), Eloc.std().item()
        gradE = (logp_full * (Eloc - E)).mean()
        model.zero_grad()
        gradE.backward()
        zb, dl = cnf.delta_logp(x.detach())
        o = dict(cfg=np.array([nup, ndown, B, seed], dtype=np.int64), Z=np.array(Z), z=z.detach().numpy(), x=x.detach().numpy(),
                 zback=zb.detach().numpy(), dlogp=dl.detach().numpy(),
                 logp=logp.detach().numpy(), grad=grad_logp.detach().numpy(), lap=lap_logp.detach().numpy(),
                 V=potential.detach().numpy(), Eloc=Eloc.numpy(), E=np.array(E), E_std=np.array(E_std), gradE=np.array(gradE.item()))
        for nme, p in model.named_parameters():
            o["pg_" + nme] = p.grad.numpy().reshape(-1).copy()
        for k_, v_ in o.items():
            out[f"{name}_{k_}"] = v_
        print("3d gsvmc", name, "E", E, "E_std", E_std, "%.1fs" % (time.time() - t), flush=True)
    for nm, m in (("eta", eta), ("mu", mu)):
        w1, b1, w2 = mlp_np(m)
        out[f"{nm}_w1"], out[f"{nm}_b1"], out[f"{nm}_w2"] = w1, b1, w2
    path = os.path.join(HERE, "g7_3d.npz")
    if only and os.path.exists(path):
        old = dict(np.load(path)); old.update(out); out = old
    np.savez_compressed(path, **out)


GROUPS = dict(mcmc=g_mcmc, slater=g_slater, backflow=g_backflow, cnf=g_cnf, gsvmc=g_gsvmc, betavmc=g_betavmc, d3=g_3d)

if __name__ == "__main__":
    which = sys.argv[1:] or list(GROUPS)
    R = ref_shim.load()
    for g in which:
        t0 = time.time()
        GROUPS[g](R)
        print(f"[{g}] {time.time() - t0:.1f}s", flush=True)
