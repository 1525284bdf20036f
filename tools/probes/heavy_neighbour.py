"""Probe: what does the one-walker-per-wave kernel of the heavy route cost the throughput kernel it runs beside (config 2)?
Run once with the product library and once with a `make variant NAME=noheavy EXTRA=-DFF_DIAG_NO_HEAVY` build (FERMIFLOW_LIB=...): the
second leaves the heavy walkers' outputs undefined -- a timing diagnostic only.    python tools/probes/heavy_neighbour.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as G
from fermiflow_amd import native
dev = torch.device("cuda:0")
B = 65536
model = G._model(dev, 3, 3, 2.0)
net = model.cnf.v_wrapper.v.net()
tu, td = model._tables(dev)
z, _, _ = native.mcmc_sample(tu, td, 3, 3, B, 100, 0.1, 1, dev)
cost = torch.empty(B, dtype=torch.int32, device=dev)
hg = torch.zeros(B, dtype=torch.float64, device=dev)
x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, walker_cost=cost, walker_h_out=hg)
tab = torch.full((2, 32), 0.6, dtype=torch.float64, device=dev); tab[:, :9] = 0.9
order, hm, hs = native.walker_schedule(cost, hg, tab[0], tab[1], None, interval=1.0)
ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
res = {"lib": os.environ.get("FERMIFLOW_LIB", "product"), "heavy_walkers": int((cost >= 12).sum())}
for hc in (0, -1):
    ts = []
    for rep in range(12):
        r = native.eloc(tu, td, 3, 3, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True, pass1_events=ev, walker_order=order,
                        walker_h_init=hs, walker_h_scale=1.0, walker_h_scale_loose=1.0, walker_class=cost, sens_tol=1.0, sens_tol_class=6,
                        heavy_class=hc)
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    ts = sorted(ts[2:])
    res["routed" if hc == 0 else "unrouted"] = {"pass_ms_median": ts[len(ts) // 2], "min": ts[0], "evals": r["stats"][0].item() / B}
print(json.dumps(res))
