"""Probe (round 6): time of ONE local-energy call of the matrix-core kernel with few / many first-step rejections (the retry queue's load).
usage: FERMIFLOW_LIB=... python tools/probes/retry_timing.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
from fermiflow_amd import native
dev = torch.device("cuda:0")
B = 65536
W = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "trained_weights.npz"))
for tag, hfac in (("head", 0.6), ("head", 1.2), ("trained", 0.6), ("trained", 1.0)):
    model = G._model(dev, 3, 3, 2.0)
    if tag != "head":
        v = model.cnf.v_wrapper.v
        with torch.no_grad():
            for nm, m in (("eta", v.eta), ("mu", v.mu)):
                m.fc1.weight.copy_(torch.as_tensor(W[f"{tag}_{nm}_w1"]).reshape(-1, 1))
                m.fc1.bias.copy_(torch.as_tensor(W[f"{tag}_{nm}_b1"]))
                m.fc2.weight.copy_(torch.as_tensor(W[f"{tag}_{nm}_w2"]).reshape(1, -1))
    net = model.cnf.v_wrapper.v.net()
    tu, td = model._tables(dev)
    torch.manual_seed(3)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    hg = torch.zeros(B, dtype=torch.float64, device=dev); cost = torch.zeros(B, dtype=torch.int32, device=dev)
    x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, walker_cost=cost, walker_h_out=hg)
    order = native.walker_order(cost)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ts = []
    for rep in range(4):
        r = native.eloc(tu, td, 3, 3, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True, walker_h_init=hg, walker_h_scale=hfac, walker_order=order,
                        walker_class=cost, pass1_events=ev)
        torch.cuda.synchronize(); ts.append(ev[0].elapsed_time(ev[1]))
    st = r["stats"]
    if int(st[12]) > 0:
        wait = st[8:10].view(torch.int64)[0].item() / 100.0; rd = st[10:12].view(torch.int64)[0].item() / 100.0
        print(f"    queue: {int(st[14])} retry groups ({int(st[13])} taken mid-stream), {int(st[12])} tickets; waiting {wait / 2048:.1f} us per wave, entry reads {rd / 2048:.1f} us per wave; "
              f"last wave left the main phase at {(int(st[16]) - int(st[18])) / 100.0:.0f} us, the kernel at {(int(st[17]) - int(st[18])) / 100.0:.0f} us")
    print(f"{tag} first step {hfac} x flow step: pass {min(ts):.3f} ms (runs {' '.join(f'{t:.3f}' for t in ts)})  evals/walker {st[0].item() / B:.2f}  rejected/walker {st[2].item() / B:.3f}  E {r['eloc'].mean().item():.6f}", flush=True)
