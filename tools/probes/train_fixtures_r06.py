"""Probe (round 6): weight sets TRAINED AT THEIR OWN SHAPE with the reference's loop -- init_zeros() flow,
Adam (src/FermionHO2D.py:40-43,61-72) -- for the shapes whose sweeps ran a looser sensitivity tolerance in rounds 2-5, and a config-2
set from a 3000-iteration run.  Then, per set, the sweep's largest relative E_loc error against a 1e-11 solve for sens_tol in
(1, 3, 5, 10), by cost class, with evaluations and pass time.  Writes gpurun_out/trained_r06.npz (-> tests/golden/trained_weights.npz).

usage: python tools/probes/train_fixtures_r06.py [which=soak,n12,c5] [nseeds=3]
The c5 run at lr 1e-2 diverged in round 5 (tools/probes/c5_driver_train.py); the loop here restarts a shape at lr / 3 when E leaves
[0, 10 x its first value] or a parameter turns non-finite, and records the rate it ended with."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
import fermiflow_amd as ff
from fermiflow_amd import native
from fermiflow_amd.utils import make_adam

dev = torch.device("cuda:0")
which = (sys.argv[1] if len(sys.argv) > 1 else "soak,n12,c5").split(",")
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
SHAPES = {  # tag: (nup, ndn, dim, training batch, iterations, evaluation batch)
    "soak3000": (3, 3, 2, 65536, 3000, 65536),
    "n12_1000": (6, 6, 2, 16384, 1000, 32768),
    "c5_1000": (10, 10, 3, 8192, 1000, 16384),
}
KEY = {"soak": "soak3000", "n12": "n12_1000", "c5": "c5_1000"}


def build(nup, ndn, dim):
    if dim == 2:
        return G._model(dev, nup, ndn, 2.0)
    gs = G._model(dev, 2, 2, 2.0)
    m = ff.GSVMC(nup, ndn, ff.HO3D(), ff.FreeFermion(device=dev), gs.cnf, ff.CoulombPairPotential(2.0), sp_potential=ff.HO())
    m.to(dev)
    return m


def train(tag):
    nup, ndn, dim, B, iters, _ = SHAPES[tag]
    lr = 1e-2
    while True:
        model = build(nup, ndn, dim)
        v = model.cnf.v_wrapper.v
        v.eta.init_zeros(); v.mu.init_zeros(); model.to(dev)
        opt = make_adam(model.parameters(), lr=lr)
        torch.manual_seed(1234)
        E0, ok, t0 = None, True, time.time()
        for it in range(1, iters + 1):
            g = model(B); opt.zero_grad(); g.backward(); opt.step()
            if it % 25 == 0 or it <= 3:
                E = model.E
                E0 = E if E0 is None else E0
                fin = all(torch.isfinite(p).all().item() for p in model.parameters())
                if not (fin and np.isfinite(E) and 0.0 < E < 10.0 * E0):
                    print(f"  [{tag}] lr {lr:g}: left the basin at iteration {it} (E {E}, finite {fin}) -> lr / 3", flush=True)
                    ok = False
                    break
                if it % 250 == 0:
                    print(f"  [{tag}] lr {lr:g} it {it}: E {E:.4f} E_std {model.E_std:.3f} max|w1| {v.eta.fc1.weight.abs().max().item():.3f} "
                          f"{v.mu.fc1.weight.abs().max().item():.3f} max|w2| {v.eta.fc2.weight.abs().max().item():.4f} {v.mu.fc2.weight.abs().max().item():.4f} "
                          f"({time.time() - t0:.0f} s)", flush=True)
        if ok:
            return model, lr
        lr /= 3.0


def sweep(tag, model):
    nup, ndn, dim, _, _, B = SHAPES[tag]
    tu, td = model._tables(dev)
    zs, tights = [], []
    buckets = [("<=4", 0, 4), ("5-6", 5, 6), ("7-8", 7, 8), ("9-11", 9, 11), ("12-15", 12, 15), (">=16", 16, 99)]
    for st in (1.0, 3.0, 5.0, 10.0):
        model.sens_tol, model.sens_tol_class = st, (8 if st > 1 else 6)
        model._h_scale_loose = 1.0
        model._h_flow = None
        mx, evs, ms, pl = [], [], [], []
        bmax = {b[0]: 0.0 for b in buckets}
        for k in range(nseeds):
            if len(zs) <= k:
                torch.manual_seed(500 + k)
                with torch.no_grad():
                    zs.append(model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,)))
            model.forward_from(zs[k])
            model.profile = {"stages": False}
            for rep in range(3):
                model.forward_from(zs[k])
            torch.cuda.synchronize()
            pr, model.profile = model.profile, None
            net = model.cnf.v_wrapper.v.net()
            if len(tights) <= k:
                tights.append(native.eloc(tu, td, nup, ndn, net, model.x, 0.0, 1.0, 1e-11, 1e-13, 2.0, True)["eloc"])
            rel = (model.Eloc - tights[k]).abs() / tights[k].abs()
            mx.append(rel.max().item())
            if st == 1.0:
                one = native.eloc(tu, td, nup, ndn, net, model.x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True)["eloc"]
                pl.append(((one - tights[k]).abs() / tights[k].abs()).max().item())
            for nm, lo, hi in buckets:
                sel = (model.walker_cost >= lo) & (model.walker_cost <= hi)
                if sel.any():
                    bmax[nm] = max(bmax[nm], rel[sel].max().item())
            evs.append(sum(int(s[0].item()) for s in pr["eloc_stats"]) / 3 / B)
            ms.append(sum(a.elapsed_time(b) for a, b in pr["pass1"]) / 3)
        print(f"  [{tag}] sens_tol {st:g}: max rel E_loc err by seed " + " ".join(f"{m:.2e}" for m in mx) +
              (" | plain one-tolerance call " + " ".join(f"{m:.2e}" for m in pl) if pl else "") +
              f" | evals {np.mean(evs):.2f} pass {np.mean(ms):.3f} ms | by class: " + "  ".join(f"{k}: {v:.1e}" for k, v in bmax.items()), flush=True)


out = {}
for w in which:
    tag = KEY[w]
    model, lr = train(tag)
    v = model.cnf.v_wrapper.v
    for nm, m in (("eta", v.eta), ("mu", v.mu)):
        out[f"{tag}_{nm}_w1"] = m.fc1.weight.detach().cpu().numpy().reshape(-1)
        out[f"{tag}_{nm}_b1"] = m.fc1.bias.detach().cpu().numpy().reshape(-1)
        out[f"{tag}_{nm}_w2"] = m.fc2.weight.detach().cpu().numpy().reshape(-1)
    out[f"{tag}_lr"] = np.array(lr)
    print(f"== {tag}: trained at lr {lr:g}; E {model.E:.5f} E_std {model.E_std:.3f}  max|w1| {max(v.eta.fc1.weight.abs().max().item(), v.mu.fc1.weight.abs().max().item()):.3f}", flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez_compressed("gpurun_out/trained_r06.npz", **out)
    sweep(tag, model)
