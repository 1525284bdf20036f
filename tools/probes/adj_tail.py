"""Probe: is the tabulated adjoint bound by its heaviest walkers?  Times ff_cnf_adjoint on the heavy walkers alone, on the
others alone and on everything (65536 walkers, config 2, warm-started and cost-ordered as in the sweep)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
from fermiflow_amd import native
dev = torch.device("cuda:0")
nup, ndn, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (3, 3, 65536)
model = G._model(dev, nup, ndn, 2.0)
net = model.cnf.v_wrapper.v.net()
tu, td = model._tables(dev)
z, _, _ = native.mcmc_sample(tu, td, nup, ndn, B, 100, 0.1, 1, dev)
f64 = dict(dtype=torch.float64, device=dev)
hg = torch.zeros(B, **f64); he = torch.zeros(B, **f64); cost = torch.zeros(B, dtype=torch.int32, device=dev)
x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, walker_cost=cost, walker_h_out=hg)
r = native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, walker_h_init=hg, walker_h_scale=0.6, walker_h_out=he)
w = (r["eloc"] - r["eloc"].mean()) / B

def run(sel, tag):
    zs, az, ad, hs = r["z"][sel].contiguous(), (w[:, None, None] * r["glogp0"])[sel].contiguous(), (-w)[sel].contiguous(), he[sel].contiguous()
    n = zs.shape[0]
    order = native.walker_order(cost[sel].contiguous())
    ts = []
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _, gp, st = native.cnf_adjoint(net, zs, az, ad, 0.0, 1.0, 1e-6, 1e-8, need_gx=False, want_stats=True, walker_order=order,
                                       walker_h_init=hs, walker_h_scale=1.25)
        e1.record(); torch.cuda.synchronize()
        if rep: ts.append(e0.elapsed_time(e1))
    print("%-22s %6d walkers: %.3f ms, evals/walker %.1f, max accepted %d, rejected %d" % (tag, n, sum(ts) / len(ts), st[0].item() / n, st[1].item(), st[2].item()), flush=True)

run(cost >= 0, "all")
for thr in (8, 12):
    run(cost > thr, "class > %d" % thr)
    run(cost <= thr, "class <= %d" % thr)
run(torch.argsort(cost, descending=True)[:40], "top 40")
