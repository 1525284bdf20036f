"""Probe: is the local-energy sensitivity kernel bound by its heaviest walkers?  Times the kernel on the heavy walkers
(flow-pass cost class > thr) alone and on the others alone, and prints the step counts of the heavy ones."""
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
hg = torch.zeros(B, dtype=torch.float64, device=dev); cost = torch.zeros(B, dtype=torch.int32, device=dev)
x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, walker_cost=cost, walker_h_out=hg)
base = model._h_scale_eloc

def run(sel, sens, tag, sc):
    xs, hs = x[sel].contiguous(), hg[sel].contiguous()
    n = xs.shape[0]
    c2 = torch.zeros(n, dtype=torch.int32, device=dev)
    order = native.walker_order(cost[sel].contiguous())
    ts = []
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r = native.eloc(tu, td, nup, ndn, net, xs, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True, pass1_events=(e0, e1), walker_order=order,
                        walker_cost=c2, walker_h_init=hs, walker_h_scale=sc, **({} if sens is None else dict(walker_class=torch.zeros(n, dtype=torch.int32, device=dev), sens_tol=sens, sens_tol_class=0)))
        torch.cuda.synchronize()
        if rep: ts.append(e0.elapsed_time(e1))
    st = r["stats"]
    print("%-28s %6d walkers: %.3f ms, evals/walker %.1f, stats %s" % (tag, n, sum(ts) / len(ts), st[0].item() / n, [round(v, 1) for v in st.tolist()]), flush=True)
    return c2

for thr in (8, 10, 12):
    heavy = cost > thr
    run(heavy, None, "class > %d strict" % thr, base)
    run(heavy, 10.0, "class > %d f=10" % thr, base)
run(cost <= 8, 10.0, "class <= 8 f=10", 0.75)
run(cost <= 8, None, "class <= 8 strict", base)
run(cost >= 0, None, "all strict", base)
idx = torch.argsort(cost, descending=True)[:40]
c2 = run(idx, None, "top 40 strict", base)
print("flow class:", cost[idx].tolist()); print("eloc pass cost (steps + rc):", c2.tolist())
c2 = run(idx, 10.0, "top 40 f=10", base)
print("eloc pass cost (steps + rc):", c2.tolist())
