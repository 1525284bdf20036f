"""Probe: tolerance of the sensitivity components in the local-energy pass (ff_ode.sens_tol = f for the walkers whose
flow-pass cost class, ff_ode.walker_class, is <= thr; 1 for the others).  Evaluations, time and E_loc error against a 1e-11 solve.
usage: sens_tol.py [nup ndown B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
from fermiflow_amd import native
dev = torch.device("cuda:0")
nup, ndn, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (3, 3, 65536)
n = nup + ndn
model = G._model(dev, nup, ndn, 2.0)
net = model.cnf.v_wrapper.v.net()
tu, td = model._tables(dev)
z, _, _ = native.mcmc_sample(tu, td, nup, ndn, B, 100, 0.1, 1, dev)
f64 = dict(dtype=torch.float64, device=dev)
hg = torch.zeros(B, **f64); cost = torch.zeros(B, dtype=torch.int32, device=dev)
x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, walker_cost=cost, walker_h_out=hg)
# the sweep's flow pass is warm-started (2-3 steps instead of 4): its classes are what the policy sees
cost2 = torch.zeros_like(cost); hg2 = torch.zeros_like(hg)
native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, walker_cost=cost2, walker_h_init=hg.mean().reshape(1), walker_h_scale=0.75,
                    walker_h_out=hg2, walker_h_uniform=True)
order = native.walker_order(cost2)
ref = native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-11, 1e-13, 2.0, True)["eloc"]
base = model._h_scale_eloc
print("cost classes (warm flow pass): " + " ".join("%d:%d" % (c, int((cost2 == c).sum())) for c in range(int(cost2.min()), int(cost2.max()) + 1)))
for fac, thr, sc in [(1, 0, base), (10, 7, 0.9), (10, 8, 0.9), (10, 9, 0.9), (10, 8, 0.8), (10, 8, 1.0), (10, 8, 1.1), (30, 8, 0.9), (30, 8, 1.0), (10, 99, 0.9)]:
    loose = cost2 <= thr
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r = native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True, pass1_events=(e0, e1), walker_order=order,
                        walker_h_init=hg2, walker_h_scale=base, walker_class=cost2, sens_tol=float(fac), sens_tol_class=thr,
                        walker_h_scale_loose=sc)
        torch.cuda.synchronize()
        if rep: ts.append(e0.elapsed_time(e1))
    rel = (r["eloc"] - ref).abs() / ref.abs()
    print("f=%3d class<=%2d (%.1f%% of walkers) warm x%.2f: %.3f ms evals %.2f rej %.3f | E_loc rel err max %.2e p99.9 %.2e p99 %.2e | loose max %.2e | mean E rel diff %.2e" % (
        fac, thr, 100.0 * loose.double().mean(), sc, sum(ts) / len(ts), r["stats"][0].item() / B, r["stats"][2].item() / B, rel.max(), rel.quantile(0.999),
        rel.quantile(0.99), rel[loose].max() if loose.any() else 0.0, abs(r["eloc"].mean() - ref.mean()) / abs(ref.mean())), flush=True)
