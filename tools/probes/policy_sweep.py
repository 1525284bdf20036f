"""Probe: E_loc error (vs a 1e-11 solve), RHS evaluations and pass time of the sweep's tolerance policy for several settings of
(sens_tol, sens_tol_class, sum_weight) on the four weight sets of tests/golden/trained_weights.npz + the benchmark's synthetic ones.
usage: policy_sweep.py [B] [nseeds]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
from fermiflow_amd import native
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "trained_weights.npz"))
NUP, NDN, DIM = (int(t) for t in os.environ.get("FF_POLICY_SHAPE", "3,3,2").split(","))
NTRAIN = int(os.environ.get("FF_POLICY_TRAIN_ITERS", "300"))


def build():
    if DIM == 2:
        return G._model(dev, NUP, NDN, 2.0)
    import fermiflow_amd as ff
    gs = G._model(dev, 2, 2, 2.0)
    return ff.GSVMC(NUP, NDN, ff.HO3D(), ff.FreeFermion(device=dev), gs.cnf, ff.CoulombPairPotential(2.0), sp_potential=ff.HO())


def load(model, tag):
    v = model.cnf.v_wrapper.v
    with torch.no_grad():
        for nm, m in (("eta", v.eta), ("mu", v.mu)):
            m.fc1.weight.copy_(torch.as_tensor(W[f"{tag}_{nm}_w1"]).reshape(-1, 1))
            m.fc1.bias.copy_(torch.as_tensor(W[f"{tag}_{nm}_b1"]))
            m.fc2.weight.copy_(torch.as_tensor(W[f"{tag}_{nm}_w2"]).reshape(1, -1))


configs = [(10, 8, 4), (1, 8, 4), (5, 8, 4), (3, 8, 4), (10, 7, 4), (10, 6, 4), (5, 7, 4), (10, 8, 16), (10, 8, 40), (5, 8, 16), (3, 8, 16), (30, 8, 40)]
if os.environ.get("FF_POLICY_CONFIGS"):
    configs = [tuple(float(t) for t in c.split(",")) for c in os.environ["FF_POLICY_CONFIGS"].split(";")]
tags = os.environ.get("FF_POLICY_SETS", "head,trained,driver,driver1000").split(",")
for tag in tags:
    model = build()
    if tag == "bench_trained":      # what bench.py's trained leg measures: 300 iterations at lr 1e-4 from the synthetic weights
        from fermiflow_amd.utils import make_adam
        opt = make_adam(model.parameters(), lr=1e-4)
        torch.manual_seed(1234)
        for i in range(NTRAIN):
            g = model(B); opt.zero_grad(); g.backward(); opt.step()
    elif tag.startswith("driver_train:"):      # init_zeros() + N iterations of the reference's loop at lr 1e-2 (src/FermionHO2D.py:40-43,61-72)
        from fermiflow_amd.utils import make_adam
        v = model.cnf.v_wrapper.v
        v.eta.init_zeros(); v.mu.init_zeros(); model.to(dev)
        opt = make_adam(model.parameters(), lr=1e-2)
        torch.manual_seed(1234)
        for i in range(int(tag.split(":")[1])):
            g = model(B); opt.zero_grad(); g.backward(); opt.step()
        print("   E", model.E, "max|w1|", v.eta.fc1.weight.abs().max().item(), v.mu.fc1.weight.abs().max().item(), flush=True)
    elif tag != "head":
        load(model, tag)
    tu, td = model._tables(dev)
    zs, tights = [], []
    print("==", tag, flush=True)
    for cfg in configs:
        st, sc, sw = cfg[:3]
        model._h_scale_loose = float(cfg[3]) if len(cfg) > 3 else 1.0
        model.sens_tol, model.sens_tol_class, model.sum_weight = float(st), int(sc), float(sw)
        rej = []
        model._h_flow = None
        mx, p9999, evs, ms, dE = [], [], [], [], []
        buckets = [("<=4", 0, 4), ("5", 5, 5), ("6", 6, 6), ("7", 7, 7), ("8", 8, 8), ("9-11", 9, 11), ("12-13", 12, 13), ("14-15", 14, 15), (">=16", 16, 99)]
        bmax = {b[0]: 0.0 for b in buckets}; bcnt = {b[0]: 0 for b in buckets}
        for k in range(nseeds):
            if len(zs) <= k:
                torch.manual_seed(int(os.environ.get('FF_POLICY_SEED0', '500')) + k)
                with torch.no_grad():
                    zs.append(model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,)))
            model.forward_from(zs[k])
            model.profile = {"stages": False}
            for rep in range(3):
                model.forward_from(zs[k])
            torch.cuda.synchronize()
            pr, model.profile = model.profile, None
            if len(tights) <= k:
                tights.append(native.eloc(tu, td, NUP, NDN, model.cnf.v_wrapper.v.net(), model.x, 0.0, 1.0, 1e-11, 1e-13, 2.0, True)["eloc"])
            rel = (model.Eloc - tights[k]).abs() / tights[k].abs()
            mx.append(rel.max().item()); p9999.append(rel.quantile(0.9999).item())
            cost = model.walker_cost
            for nm, lo, hi in buckets:
                sel = (cost >= lo) & (cost <= hi)
                if sel.any():
                    bmax[nm] = max(bmax[nm], rel[sel].max().item()); bcnt[nm] += int(sel.sum())
            evs.append(sum(int(s[0].item()) for s in pr["eloc_stats"]) / 3 / B)
            rej.append(sum(int(s[2].item()) for s in pr["eloc_stats"]) / 3 / B)
            ms.append(sum(a.elapsed_time(b) for a, b in pr["pass1"]) / 3)
            dE.append(abs(model.Eloc.mean().item() / tights[k].mean().item() - 1))
        print(f"  sens_tol {st:g} class<={sc:g} sum_w {sw:g}: max err {max(mx):.2e} p99.99 {max(p9999):.2e} mean-E {max(dE):.1e} | h_scale_loose {model._h_scale_loose:g} evals {np.mean(evs):.2f} rejected steps/walker {np.mean(rej):.3f} pass {np.mean(ms):.3f} ms\n      by class: " +
              "  ".join(f"{k}: {bmax[k]:.1e} ({bcnt[k] // nseeds})" for k in bmax) +
              ("\n      first-step factors, classes 2..14: " + " ".join(f"{v:.2f}" for v in model._h_tab[model._h_tab_cur][2:15].tolist()) if model._h_tab is not None else ""), flush=True)
