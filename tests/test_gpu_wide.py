"""GPU parity of the one-walker-per-workgroup kernel family (csrc/ff_wide.hip, csrc/ff_adj_wide.h): BASELINE.json configs[4]
(nup = ndown = 10 in a 3-D trap, n d = 60) and every 2-D system beyond 12 particles -- shapes the reference serves through
its shape-generic PyTorch code (src/equivariant_funs.py:17-102, --nup/--ndown of src/FermionHO2D.py:18-19).

The oracle (oracle/ff_oracle.c) is the checker; the north-star bar for E_loc is 1e-5 relative (fp64).  d = 3: the reference has no
3-D orbital list, but everything else on its path is dimension-generic -- tests/golden/g7_3d.npz holds what the REFERENCE computes on
(B, n, 3) walkers with 3-D closures made of its own HO2D closures (make_golden.py group d3); the oracle is pinned to it
(tests/test_oracle_golden.py::test_3d_*) and the HIP path is compared with it directly below (test_3d_*_vs_reference)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.common import net_arrays, mcmc_noise_from_seed, gsvmc_param_grads, make_flow, T

pytestmark = pytest.mark.gpu
ELOC_RTOL = 1e-5     # BASELINE.json north_star: "E_loc within 1e-5 relative fp64"
# bounds of the configs[4] known-answer test (measured: tools/probes/c5_cond.py)
# measured (tools/probes/c5_cond.py, 131 072 Metropolis walkers): max |E_loc - 60| = 4.4e-12 -- walkers drawn from |psi|^2 stay away from
# its nodes, the largest condition number of a Slater matrix in the batch is 6e4 (random points, test_ho3d_base_distribution, do not)
C5_KAT_MAX, C5_KAT_MEAN = 1e-9, 1e-11


def N(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _onet(model):
    v = model.cnf.v_wrapper.v
    return O.Net(tuple(N(t) for t in (v.eta.fc1.weight, v.eta.fc1.bias, v.eta.fc2.weight)),
                 tuple(N(t) for t in (v.mu.fc1.weight, v.mu.fc1.bias, v.mu.fc2.weight)))


@pytest.fixture
def wide_family():
    """ff_set_kernel_family(1): every particle number takes the one-walker-per-workgroup kernels."""
    from fermiflow_amd import native
    prev = native.set_kernel_family(1)
    yield
    native.set_kernel_family(prev)


@pytest.mark.parametrize("nup,ndn", [(3, 3), (2, 1), (4, 3), (6, 6)])
def test_wide_kernels_agree_with_the_wave_per_group_kernels(dev, nup, ndn):
    """Same walkers, same weights, both kernel families (ff_set_kernel_family): flow, log-density, local energy (E_loc, grad,
    logp) and the adjoint's parameter gradient agree to solver tolerance -- the two families share nothing but the radial
    table and the step-size rules."""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    model = Gm._model(dev, nup, ndn, 2.0)
    v = model.cnf.v_wrapper.v
    B = 96
    torch.manual_seed(11 + nup)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    res = []
    for fam in (0, 1):
        prev = native.set_kernel_family(fam)
        try:
            net = v.net(refresh=True)
            x = native.cnf_generate(net, z, 0.0, 1.0, 1e-9, 1e-11)
            zb, dl = native.cnf_delta_logp(net, x, 0.0, 1.0, 1e-9, 1e-11)
            tu, td = model._tables(dev)
            r = native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-9, 1e-11, 2.0, True, want_stats=True)
            assert int(r["stats"][3]) == 0
            w = (r["eloc"] - r["eloc"].mean()) / B
            gx, gp = native.cnf_adjoint(net, r["z"].clone(), w[:, None, None] * r["glogp0"], -w, 0.0, 1.0, 1e-9, 1e-11)
            res.append(dict(x=N(x), zb=N(zb), dl=N(dl), eloc=N(r["eloc"]), grad=N(r["grad"]), logp=N(r["logp"]), gx=N(gx), gp=N(gp)))
        finally:
            native.set_kernel_family(prev)
    a, b = res
    np.testing.assert_allclose(b["x"], a["x"], atol=1e-8)
    np.testing.assert_allclose(b["zb"], a["zb"], atol=1e-8)
    np.testing.assert_allclose(b["dl"], a["dl"], atol=1e-8)
    assert (np.abs(b["eloc"] - a["eloc"]) / np.abs(a["eloc"])).max() < 1e-6
    np.testing.assert_allclose(b["grad"], a["grad"], atol=1e-6 * max(1.0, np.abs(a["grad"]).max()))
    np.testing.assert_allclose(b["logp"], a["logp"], atol=1e-7)
    np.testing.assert_allclose(b["gx"], a["gx"], atol=1e-7 * max(1.0, np.abs(a["gx"]).max()))
    np.testing.assert_allclose(b["gp"], a["gp"], atol=1e-7 * max(1.0, np.abs(a["gp"]).max()))


@pytest.mark.parametrize("nup,ndn", [(7, 6), (8, 8), (12, 12)])
def test_more_than_twelve_particles_vs_oracle(golden, dev, nup, ndn):
    """GSVMC(7, 6, HO2D(), ...) and beyond (VERDICT r02 missing #2): flow, local energy and the parameter gradient of a 13-,
    16- and 24-particle dot against the oracle."""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    G = golden["g5_gsvmc"]
    n = nup + ndn
    model = Gm._model(dev, nup, ndn, 1.0)
    net = O.Net(*net_arrays(G, "z2_nt_"))
    B = 6 if n <= 16 else 3
    torch.manual_seed(100 + n)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    v = model.cnf.v_wrapper.v
    x = native.cnf_generate(v.net(), z, 0.0, 1.0, 1e-8, 1e-10)
    xo, _ = O.cnf_generate(N(z), net, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(N(x), xo, atol=1e-7)
    r = model.local_energy(x, want_stats=True)
    assert int(r["stats"][3]) == 0
    ref = O.eloc(N(x), nup, ndn, net, 1.0, rtol=1e-9, atol=1e-11)
    assert (np.abs(N(r["eloc"]) - ref["eloc"]) / np.abs(ref["eloc"])).max() < ELOC_RTOL
    np.testing.assert_allclose(N(r["grad"]), ref["grad"], atol=2e-5 * max(1.0, np.abs(ref["grad"]).max()))
    w = (r["eloc"] - r["eloc"].mean()) / B
    _, gp = native.cnf_adjoint(v.net(), r["z"], w[:, None, None] * r["glogp0"], -w, 0.0, 1.0, 1e-8, 1e-10, need_gx=False)
    zo, dlo, _ = O.cnf_delta_logp(N(x), net, rtol=1e-10, atol=1e-12)
    _, g0o, _ = O.logprob(zo, nup, ndn)
    _, gpo, _ = O.cnf_adjoint(zo, dlo, N(w)[:, None, None] * g0o, -N(w), net, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(N(gp), gpo, atol=2e-5 * np.abs(gpo).max())


def test_thirteen_particle_training_iteration(dev):
    """the drop-in surface at 13 particles: GSVMC(7, 6, HO2D(), ...)(batch).backward() fills every .grad."""
    import __graft_entry__ as Gm
    model = Gm._model(dev, 7, 6, 2.0)
    torch.manual_seed(3)
    g = model(2048)
    g.backward()
    assert np.isfinite(model.E) and np.isfinite(model.E_std)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def _model3d(dev, nup, ndn, Z, zero):
    import fermiflow_amd as ff
    import __graft_entry__ as Gm
    gs = Gm._model(dev, 2, 2, Z)
    if zero:
        for p in gs.parameters():
            torch.nn.init.zeros_(p)
    return ff.GSVMC(nup, ndn, ff.HO3D(), ff.FreeFermion(device=dev), gs.cnf, ff.CoulombPairPotential(Z), sp_potential=ff.HO())


def test_config5_known_answer_at_full_size(dev):
    """BASELINE.json configs[4] at its per-GPU size: nup = ndown = 10 in the 3-D trap (closed shells 0..2), 131 072 walkers.
    Zero flow and Z = 0: every walker's E_loc is the sum of the occupied orbital energies, 2 (1.5 + 3 * 2.5 + 6 * 3.5) = 60
    (tests/test_basedist.py:5-60 one dimension up).  The assertion is a MAXIMUM over all 131 072 walkers plus the batch mean and E_std
    (the quantile criteria of rounds 3-4 left the worst walkers unbounded and the mean unchecked; they were
    inherited from the random-point test, where 10 x 10 determinants can be ill-conditioned -- Metropolis walkers are not)."""
    model = _model3d(dev, 10, 10, 0.0, True)
    torch.manual_seed(4)
    g = model(131072)
    g.backward()
    assert model.x.shape == (131072, 20, 3)
    err = (model.Eloc - 60.0).abs()
    assert err.max().item() < C5_KAT_MAX, err.max().item()          # EVERY walker
    assert abs(model.E - 60.0) < C5_KAT_MEAN, model.E
    assert model.E_std < C5_KAT_MAX
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.parametrize("bits", [64, 32])
def test_config5_heaviest_walkers_vs_oracle(dev, bits, capsys):
    """configs[4] with the benchmark's flow and Z = 2, a production sweep of 8 192 walkers: EVERY walker of the top cost classes (the 192
    most expensive by the flow pass's cost class -- close approaches to a vanishing radius, where the sensitivities take the most
    steps and carry the largest error) plus the first 64 walkers of the batch, 256 in all, against the oracle's generic jet
    arithmetic (O.eloc3d) at the north-star bar, for both precisions of the sensitivity matrices (config 2 had
    this test, configs 4 / 5 did not)."""
    from fermiflow_amd import native
    model = _model3d(dev, 10, 10, 2.0, False)
    prev = native.set_sens_precision(bits)
    try:
        torch.manual_seed(5)
        g = model(8192)
        g.backward()
    finally:
        native.set_sens_precision(prev)
    cost = model.walker_cost.to(torch.int64)
    order = torch.argsort(cost, descending=True, stable=True)
    heavy = order[:192]
    light = torch.arange(64, device=dev)
    light = light[~torch.isin(light, heavy)]
    idx = torch.cat([heavy, light])
    ref = O.eloc3d(N(model.x[idx]), 10, 10, _onet(model), 2.0, rtol=1e-9, atol=1e-11)
    rel = np.abs(N(model.Eloc[idx]) - ref["eloc"]) / np.abs(ref["eloc"])
    nh = len(heavy)
    with capsys.disabled():
        print(f"\n[config 5, {bits}-bit sensitivity matrices] cost classes of the batch {int(cost.min())}..{int(cost.max())}; the {nh} heaviest "
              f"(class >= {int(cost[heavy].min())}): max rel E_loc error vs oracle {rel[:nh].max():.2e}; {len(light)} ordinary walkers: {rel[nh:].max():.2e}")
    assert rel.max() < ELOC_RTOL, rel.max()
    np.testing.assert_allclose(N(model.Eloc[idx]).mean(), ref["eloc"].mean(), rtol=1e-6)


@pytest.mark.parametrize("bits", [64, 32])
def test_config5_local_energy_vs_oracle(dev, bits, capsys):
    """configs[4] with the benchmark's flow and Z = 2: 24 walkers of a training iteration against the oracle's generic jet
    arithmetic (O.eloc3d), E_loc within the north-star bar; flow and parameter gradient too.  bits = 32: the same sweep with the
    sensitivity matrices in fp32 on v_mfma_f32_16x16x4 (ff_set_sens_precision(32) -- what `bench.py --workload c5` and the
    driver's --sens_bits 32 run): the SAME bar against the SAME fp64 oracle."""
    from fermiflow_amd import native
    model = _model3d(dev, 10, 10, 2.0, False)
    prev = native.set_sens_precision(bits)
    try:
        torch.manual_seed(5)
        g = model(2048)
        g.backward()
    finally:
        native.set_sens_precision(prev)
    assert np.isfinite(model.E) and all(torch.isfinite(p.grad).all() for p in model.parameters())
    net = _onet(model)
    nb = 24
    ref = O.eloc3d(N(model.x[:nb]), 10, 10, net, 2.0, rtol=1e-9, atol=1e-11)
    rel = np.abs(N(model.Eloc[:nb]) - ref["eloc"]) / np.abs(ref["eloc"])
    with capsys.disabled():
        print(f"\n[config 5, {bits}-bit sensitivity matrices] 24 walkers: max rel E_loc error vs oracle {rel.max():.2e}")
    assert rel.max() < ELOC_RTOL, rel.max()
    # ff_ode.compact_finish: the same walkers finished in the sensitivity kernel's epilogue (compact workspace), same bar
    prev = native.set_sens_precision(bits)
    try:
        tu, td = model._tables(dev)
        rc = native.eloc(tu, td, 10, 10, model.cnf.v_wrapper.v.net(), model.x[:nb].contiguous(), 0.0, 1.0, 1e-6, 1e-8, 2.0, True, compact=True)
    finally:
        native.set_sens_precision(prev)
    relc = np.abs(N(rc["eloc"]) - ref["eloc"]) / np.abs(ref["eloc"])
    with capsys.disabled():
        print(f"[config 5, {bits}-bit, finish fused into the sensitivity kernel] max rel E_loc error vs oracle {relc.max():.2e}")
    assert relc.max() < ELOC_RTOL, relc.max()
    if bits == 32:
        return
    # stand-alone calls at a tight tolerance: flow, log-density and the adjoint's parameter gradient
    v = model.cnf.v_wrapper.v
    torch.manual_seed(6)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (4,))
    x = native.cnf_generate(v.net(), z, 0.0, 1.0, 1e-9, 1e-11)
    xo, _ = O.cnf_generate(N(z), net, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(N(x), xo, atol=1e-7)
    zb, dl = native.cnf_delta_logp(v.net(), x, 0.0, 1.0, 1e-9, 1e-11)
    zo, dlo, _ = O.cnf_delta_logp(N(x), net, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(N(zb), zo, atol=1e-7)
    np.testing.assert_allclose(N(dl), dlo, atol=1e-7)
    rng = np.random.default_rng(0)
    az, ad = rng.normal(size=zo.shape), rng.normal(size=4)
    gx, gp = native.cnf_adjoint(v.net(), torch.as_tensor(zo, device=dev), torch.as_tensor(az, device=dev), torch.as_tensor(ad, device=dev),
                                0.0, 1.0, 1e-9, 1e-11)
    gxo, gpo, _ = O.cnf_adjoint(zo, dlo, az, ad, net, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(N(gx), gxo, atol=1e-6 * max(1.0, np.abs(gxo).max()))
    np.testing.assert_allclose(N(gp), gpo, atol=1e-6 * np.abs(gpo).max())
    # ff_ode.walker_h_equal on the one-walker-per-workgroup adjoint (ff_open_steps_kernel in front of the launch): the same call as with
    # the rounded steps passed explicitly, and the oracle's gradient
    targs = (torch.as_tensor(zo, device=dev), torch.as_tensor(az, device=dev), torch.as_tensor(ad, device=dev), 0.0, 1.0, 1e-9, 1e-11)
    h = torch.tensor([0.21, 0.3, 0.46, 1.2], dtype=torch.float64, device=dev)
    hr = torch.where(h * 1.1 < 1.0, 1.0 / torch.ceil(1.0 / (h * 1.1) - 1e-9), h * 1.1)
    ge = native.cnf_adjoint(v.net(), *targs, want_stats=True, walker_h_init=h, walker_h_scale=1.1, walker_h_equal=True)
    gr = native.cnf_adjoint(v.net(), *targs, want_stats=True, walker_h_init=hr, walker_h_scale=1.0)
    assert torch.equal(ge[0], gr[0]) and torch.equal(ge[1], gr[1]) and torch.equal(ge[2][:4], gr[2][:4]) and int(ge[2][3]) == 0
    np.testing.assert_allclose(N(ge[1]), gpo, atol=1e-6 * np.abs(gpo).max())


@pytest.mark.parametrize("nup,ndn,dim", [(3, 3, 2), (2, 1, 2), (7, 6, 2), (12, 12, 2), (5, 4, 3), (10, 10, 3)])
def test_compact_finish_equals_the_finish_kernels(dev, nup, ndn, dim):
    """ff_ode.compact_finish: the one-walker-per-workgroup kernels finish their walkers in their epilogue (tr(H0 J J^T) on the matrix
    cores, Slater table built by the workgroup; beyond 24 coordinates nothing but z(t0) and Delta in the workspace) -- every output
    against the separate finish kernels on the same sensitivities."""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    model = Gm._model(dev, nup, ndn, 2.0) if dim == 2 else _model3d(dev, nup, ndn, 2.0, False)
    v = model.cnf.v_wrapper.v
    B = 200
    torch.manual_seed(31 + nup)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    prev = native.set_kernel_family(1)
    try:
        net = v.net(refresh=True)
        x = native.cnf_generate(net, z, 0.0, 1.0, 1e-8, 1e-10)
        tu, td = model._tables(dev)
        a = native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-8, 1e-10, 2.0, True, compact=False)
        b = native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-8, 1e-10, 2.0, True, compact=True)
    finally:
        native.set_kernel_family(prev)
    M = (nup + ndn) * dim
    assert b["z"].untyped_storage().nbytes() < (a["z"].untyped_storage().nbytes() if M > 24 else 1 << 62)
    if M > 24:
        assert b["z"].untyped_storage().nbytes() == 8 * (B * (M + 1) + 2)
    for k in ("z", "dlogp"):
        assert torch.equal(a[k], b[k]), k      # the same integration, bit for bit
    for k in ("logp", "V", "eloc", "lap", "grad", "glogp0"):
        sc = max(1.0, a[k].abs().max().item())
        assert (a[k] - b[k]).abs().max().item() < 1e-9 * sc, (k, (a[k] - b[k]).abs().max().item(), sc)


def test_multi_wave_kernels_repeat_bit_for_bit(dev):
    """Regression test of a cross-wave race found in round 4: in the kernels with more than one wave per walker the lanes
    that own no row of A = dv/dz parked a -0.0 on A[M][0..d-1] -- the (grad Delta)' entries of particle 0, stored by another wave in the
    same phase -- and now and then the zero won: an evaluation lost a component, the step was rejected, the result moved by 1e-11.
    Alternating different shapes (what exposes it: tools/probes/det_scan.py) every run must now repeat the first one bit for bit."""
    import __graft_entry__ as Gm
    import fermiflow_amd as ff
    from fermiflow_amd import native
    def setup(nup, ndn, dim, bits, compact=False):
        if dim == 2:
            model = Gm._model(dev, nup, ndn, 2.0)
        else:
            model = _model3d(dev, nup, ndn, 2.0, False)
        torch.manual_seed(31 + nup)
        z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (200,))
        net = model.cnf.v_wrapper.v.net(refresh=True)
        x = native.cnf_generate(net, z, 0.0, 1.0, 1e-8, 1e-10)
        tu, td = model._tables(dev)
        def run():
            prev = native.set_sens_precision(bits)
            try:
                return native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-8, 1e-10, 2.0, True, want_stats=True, compact=compact)
            finally:
                native.set_sens_precision(prev)
        return run
    # (the last two: the same kernels with the finish in their epilogue, ff_ode.compact_finish)
    shapes = [(7, 6, 2, 64), (7, 6, 2, 32), (12, 12, 2, 64), (10, 10, 3, 64), (10, 10, 3, 32), (12, 12, 2, 64, True), (10, 10, 3, 32, True)]
    runs = {s: setup(*s) for s in shapes}
    ref = {s: runs[s]() for s in shapes}
    for it in range(25):
        for s in shapes:
            r = runs[s]()
            assert int(r["stats"][0]) == int(ref[s]["stats"][0]), (s, it)
            for k in ("z", "dlogp", "eloc", "grad"):
                assert torch.equal(r[k], ref[s][k]), (s, it, k)


def test_wide_direct_evaluation_equals_the_table_path(dev):
    """Backflow.net(radial="exact") (what FERMIFLOW_RADIAL=exact selects; no radial table: every sigmoid evaluated, parameter gradient integrated per hidden unit) against the
    tabulated kernels, 13 particles: E_loc and the parameter gradient agree far below the solver tolerance."""
    import __graft_entry__ as Gm
    from fermiflow_amd import native
    out = []
    for mode in ("table", "exact"):
        model = Gm._model(dev, 7, 6, 2.0)
        v = model.cnf.v_wrapper.v
        torch.manual_seed(21)
        z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (64,))
        net = v.net(radial=mode, refresh=True)
        x = native.cnf_generate(net, z, 0.0, 1.0, 1e-9, 1e-11)
        tu, td = model._tables(dev)
        r = native.eloc(tu, td, 7, 6, net, x, 0.0, 1.0, 1e-9, 1e-11, 2.0, True)
        w = (r["eloc"] - r["eloc"].mean()) / 64
        _, gp = native.cnf_adjoint(net, r["z"].clone(), w[:, None, None] * r["glogp0"], -w, 0.0, 1.0, 1e-9, 1e-11, need_gx=False)
        out.append((N(x), N(r["eloc"]), N(gp)))
    np.testing.assert_allclose(out[1][0], out[0][0], atol=1e-9)
    assert (np.abs(out[1][1] - out[0][1]) / np.abs(out[0][1])).max() < 1e-8
    np.testing.assert_allclose(out[1][2], out[0][2], atol=1e-8 * np.abs(out[0][2]).max())


def test_config5_fp32_sensitivity_path_error_report(dev):
    """ff_set_sens_precision(32): J, A = dv/dz and S = J J^T in fp32 on v_mfma_f32_16x16x4 (BASELINE.json configs[4]: "fp32 MFMA
    path"), everything else fp64.  8 192 walkers against the fp64 kernels on the same walkers: the maximum E_loc difference stays
    inside the north-star bar (1e-5; the oracle comparison of the same path is test_config5_local_energy_vs_oracle[32])."""
    from fermiflow_amd import native
    model = _model3d(dev, 10, 10, 2.0, False)
    v = model.cnf.v_wrapper.v
    torch.manual_seed(8)
    B = 8192
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    net = v.net(refresh=True)
    x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8)
    tu, td = model._tables(dev)
    out = {}
    for bits in (64, 32):
        prev = native.set_sens_precision(bits)
        try:
            r = native.eloc(tu, td, 10, 10, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True)
            assert int(r["stats"][3]) == 0
            out[bits] = (r["eloc"].clone(), r["grad"].clone(), int(r["stats"][0]))
        finally:
            native.set_sens_precision(prev)
    rel = ((out[32][0] - out[64][0]) / out[64][0]).abs()
    gerr = (out[32][1] - out[64][1]).abs().max().item() / out[64][1].abs().max().item()
    q = torch.quantile(rel, torch.tensor([0.5, 0.999], dtype=torch.float64, device=dev))
    print(f"[fp32 sensitivities, config 5] E_loc rel. error vs fp64: median {q[0].item():.2e}, p99.9 {q[1].item():.2e}, max {rel.max().item():.2e}; "
          f"grad logp {gerr:.2e} of its largest entry; RHS evaluations {out[32][2] / B:.2f} (fp64: {out[64][2] / B:.2f}) per walker; "
          f"mean E_loc {out[32][0].mean().item():.6f} vs {out[64][0].mean().item():.6f}")
    assert q[0].item() < 1e-6 and rel.max().item() < ELOC_RTOL and gerr < 1e-5      # measured: 6.5e-9 / 3.5e-7 / 2.3e-7
    assert abs(out[32][0].mean().item() / out[64][0].mean().item() - 1) < 1e-7


def test_driver_runs_the_three_dimensional_trap(dev, capsys):
    """the drop-in driver with --dim 3 (HO3D orbitals) and --sens_bits 32: BASELINE.json configs[4]'s command line at a small batch;
    zero-initialised flow (the driver's default, src/FermionHO2D.py:40-43) -> the first iteration's E is the free-fermion value plus
    the Coulomb energy, finite, and training moves it."""
    from fermiflow_amd import FermionHO2D, native
    try:
        torch.manual_seed(2)
        FermionHO2D.main(["--dim", "3", "--nup", "10", "--ndown", "10", "--Z", "0.5", "--batch", "1024", "--iternum", "2", "--sens_bits", "32"])
    finally:
        native.set_sens_precision(64)
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("iter:")]
    assert len(lines) == 2
    E = [float(ln.split("E:")[1].split()[0]) for ln in lines]
    assert all(np.isfinite(E)) and 60.0 < E[0] < 120.0 and E[0] != E[1]


# ------------------------------------------------------------------------------------------------------------------------------
# d = 3 against the REFERENCE (tests/golden/g7_3d.npz): log-density with gradient and Laplacian, bit-exact
# Metropolis, the local energy and a whole GSVMC iteration with its six parameter gradients.

def test_3d_log_density_vs_reference(golden, dev):
    """FreeFermion.log_prob / y_grad_laplacian on (B, n, 3) walkers: LogAbsSlaterDet.apply (src/slater.py:13-62) of 1, 4 and 10 3-D
    orbitals and the two-species log-density (src/base_dist.py:49-56) incl. configs[4]'s closed shells (10, 10)."""
    import fermiflow_amd as ff
    G = golden["g7_3d"]
    h = ff.HO3D()
    assert [(o.nx, o.ny, o.nz) for o in h.orbitals] == [tuple(r) for r in G["nxyz"].tolist()]
    bd = ff.FreeFermion(device=dev)
    for n in (1, 4, 10):
        up = tuple(h.orbitals[k] for k in G[f"n{n}_orb"])
        x = T(G[f"n{n}_x"], dev)
        np.testing.assert_allclose(N(bd.log_prob(up, (), x)) / 2, G[f"n{n}_logabsdet"], atol=1e-12)
        lp, g, lap = ff.y_grad_laplacian(ff.utils.freefermion_logp(bd, up, ()), x)
        np.testing.assert_allclose(N(lp) / 2, G[f"n{n}_logabsdet"], atol=1e-12)
        np.testing.assert_allclose(N(g) / 2, G[f"n{n}_grad"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(N(lap) / 2, G[f"n{n}_lap"], rtol=1e-9, atol=1e-6)
    for tag in ("lp36", "lp1010"):
        up, dn = tuple(h.orbitals[k] for k in G[tag + "_up"]), tuple(h.orbitals[k] for k in G[tag + "_dn"])
        lp, g, lap = ff.y_grad_laplacian(ff.utils.freefermion_logp(bd, up, dn), T(G[tag + "_x"], dev))
        np.testing.assert_allclose(N(lp), G[tag + "_logp"], atol=1e-11)
        np.testing.assert_allclose(N(g), G[tag + "_grad"], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(N(lap), G[tag + "_lap"], rtol=1e-8, atol=1e-5)


@pytest.mark.parametrize("name", ["m2d2", "m10d10", "m4d3"])
def test_3d_mcmc_bit_exact_vs_reference(golden, dev, name):
    """The Metropolis loop body of the reference (src/base_dist.py:63-70) on (B, n, 3) walkers from the torch-CPU noise stream:
    acceptance indices and final walkers bit-identical -- the sixteen-lane sampler (m10d10: configs[4]'s determinants) and the
    one-lane-per-walker kernels alike."""
    import fermiflow_amd as ff
    G = golden["g7_3d"]
    nup, ndn, g0, g, u, accept = mcmc_noise_from_seed(G, name, dim=3)
    h = ff.HO3D()
    bd = ff.FreeFermion(device=dev)
    x, logp, acc = bd.sample_with_noise(h.orbitals[:nup], h.orbitals[:ndn], T(g0, dev), T(g, dev), T(u, dev))
    assert (N(acc) == accept).all(), "acceptance indices differ from the reference"
    assert (N(x) == G[name + "_x"]).all(), "final walkers are not bit-identical"
    np.testing.assert_allclose(N(logp), G[name + "_logp"], atol=1e-11)


@pytest.mark.parametrize("name", ["e2d2", "e5d4", "e1d1"])
def test_3d_local_energy_vs_reference(golden, dev, name):
    """logp, grad, Laplacian, V, E_loc per walker of y_grad_laplacian(GSVMC.logp, x) on 3-D walkers (src/utils.py:40-65 through the
    nested adjoints; src/VMC.py:46-55) at the reference's default tolerance -- e2d2 on the wave-per-walker-group kernels (M = 12),
    e5d4 on the one-walker-per-workgroup family (M = 27), e1d1 the column sweep."""
    import fermiflow_amd as ff
    G = golden["g7_3d"]
    nup, ndn, B, seed = (int(v) for v in G[name + "_cfg"])
    cnf = make_flow(*net_arrays(G, ""), dev)
    model = ff.GSVMC(nup, ndn, ff.HO3D(), ff.FreeFermion(device=dev), cnf, ff.CoulombPairPotential(float(G[name + "_Z"])), sp_potential=ff.HO())
    x = T(G[name + "_x"], dev)
    np.testing.assert_allclose(N(cnf.generate(T(G[name + "_z"], dev))), G[name + "_x"], atol=2e-6)
    zb, dl = cnf.delta_logp(x)
    np.testing.assert_allclose(N(zb), G[name + "_zback"], atol=2e-6)
    np.testing.assert_allclose(N(dl), G[name + "_dlogp"], atol=2e-6)
    r = model.local_energy(x, want_stats=True)
    assert int(r["stats"][3]) == 0
    el = G[name + "_Eloc"]
    assert (np.abs(N(r["eloc"]) - el) / np.abs(el)).max() < ELOC_RTOL
    np.testing.assert_allclose(N(r["eloc"]), el, rtol=2e-7)         # what is actually achieved
    np.testing.assert_allclose(N(r["logp"]), G[name + "_logp"], atol=1e-6)
    np.testing.assert_allclose(N(r["grad"]), G[name + "_grad"], atol=1e-6)
    np.testing.assert_allclose(N(r["lap"]), G[name + "_lap"], rtol=1e-7, atol=1e-5)
    np.testing.assert_allclose(N(r["V"]), G[name + "_V"], rtol=1e-12)
    if nup + ndn > 4:       # the same walkers with the sensitivity matrices in fp32 (configs[4]'s "fp32 MFMA path"), same bar
        from fermiflow_amd import native
        prev = native.set_sens_precision(32)
        try:
            r32 = model.local_energy(x)
        finally:
            native.set_sens_precision(prev)
        assert (np.abs(N(r32["eloc"]) - el) / np.abs(el)).max() < ELOC_RTOL


@pytest.mark.parametrize("name,rt,at,vtol,gtol", [("e2d2", 1e-10, 1e-12, 1e-7, 1e-6), ("e2d2", 1e-6, 1e-8, 1e-5, 1e-5),
                                                  ("e5d4", 1e-10, 1e-12, 1e-7, 1e-6), ("e5d4", 1e-6, 1e-8, 1e-5, 1e-5),
                                                  ("e1d1", 1e-10, 1e-12, 1e-7, 1e-6)])
def test_3d_gsvmc_forward_backward_vs_reference(golden, dev, name, rt, at, vtol, gtol, capsys):
    """GSVMC.forward -> .backward() end to end (src/VMC.py:40-59, src/FermionHO2D.py:69-72) on the reference's 3-D base walkers through
    the production sweep: x, E, E_std, gradE and every parameter's .grad against the reference's own numbers.  (e5d4 at the default
    tolerance: the gradient of FOUR walkers, nothing averages -- 4e-6 of its largest entry measured.)"""
    import fermiflow_amd as ff
    G = golden["g7_3d"]
    nup, ndn, B, seed = (int(v) for v in G[name + "_cfg"])
    cnf = make_flow(*net_arrays(G, ""), dev)
    cnf.rtol, cnf.atol = rt, at
    model = ff.GSVMC(nup, ndn, ff.HO3D(), ff.FreeFermion(device=dev), cnf, ff.CoulombPairPotential(float(G[name + "_Z"])), sp_potential=ff.HO())
    for sweep in range(2):
        gradE = model.forward_from(T(G[name + "_z"], dev))
        model.zero_grad()
        gradE.backward()
        np.testing.assert_allclose(N(model.x), G[name + "_x"], atol=100 * rt)
        np.testing.assert_allclose(model.E, float(G[name + "_E"]), rtol=vtol)
        np.testing.assert_allclose(model.E_std, float(G[name + "_E_std"]), rtol=10 * vtol)
        np.testing.assert_allclose(gradE.item(), float(G[name + "_gradE"]), rtol=100 * vtol, atol=1e-9)
        ref = gsvmc_param_grads(G, name)
        got = np.concatenate([N(p.grad).reshape(-1) for p in model.parameters()])
        with capsys.disabled():
            print(f"\n[3-D sweep vs reference, {name}, rtol {rt:g}, sweep {sweep}] E rel. error {abs(model.E / float(G[name + '_E']) - 1):.1e}, "
                  f"gradient error {np.abs(got - ref).max() / np.abs(ref).max():.1e} of its largest entry")
        np.testing.assert_allclose(got, ref, atol=gtol * np.abs(ref).max())
