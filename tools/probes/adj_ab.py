"""A/B of the adjoint (ff_cnf_adjoint) at config 2: time of the whole call on the walkers of a fresh sweep."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as G
from fermiflow_amd import native
dev = torch.device("cuda:0")
B = 65536
model = G._model(dev, 3, 3, 2.0)
torch.manual_seed(1234)
g = model(B); g.backward()
net = model.cnf.v_wrapper.v.net()
tu, td = model._tables(dev)
hg = torch.empty(B, dtype=torch.float64, device=dev); he = torch.empty_like(hg)
x = model.x
r = native.eloc(tu, td, 3, 3, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True)
native.cnf_generate(net, r["z"], 0.0, 1.0, 1e-6, 1e-8, walker_h_out=hg)
w = (r["eloc"] - r["eloc"].mean()) / B
az, ad = w[:, None, None] * r["glogp0"], -w
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    _, gp, st = native.cnf_adjoint(net, r["z"], az, ad, 0.0, 1.0, 1e-6, 1e-8, need_gx=False, want_stats=True, walker_h_init=hg, walker_h_scale=0.75)
    e1.record(); torch.cuda.synchronize()
print(os.environ.get("FERMIFLOW_LIB", "default")[-20:], "adjoint ms %.3f" % e0.elapsed_time(e1), "evals/walker %.2f" % (st[0].item() / B), "|gp| %.10e" % gp.norm().item())
