"""Probe: bench.py's trained leg (300 iterations at lr 1e-4 from the synthetic weights) -- E_loc error of the sweep policy by cost class,
first and second forward_from on the same walkers."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
from fermiflow_amd import native
from fermiflow_amd.utils import make_adam
dev = torch.device("cuda:0")
B = 65536
model = G._model(dev, 3, 3, 2.0)
opt = make_adam(model.parameters(), lr=1e-4)
torch.manual_seed(1234)
for i in range(300):
    g = model(B); opt.zero_grad(); g.backward(); opt.step()
print("policy", model.sens_tol, model.sens_tol_class, model._h_scale_loose, "E", model.E)
tu, td = model._tables(dev)
for sd in (11, 12, 13):
    torch.manual_seed(sd)
    with torch.no_grad():
        z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    for rep in range(2):
        model.forward_from(z)
        x, e, cost = model.x, model.Eloc.clone(), model.walker_cost.clone()
        tight = native.eloc(tu, td, 3, 3, model.cnf.v_wrapper.v.net(), x, 0.0, 1.0, 1e-11, 1e-13, 2.0, True)["eloc"]
        rel = (e - tight).abs() / tight.abs()
        i = int(rel.argmax())
        by = {c: rel[cost == c].max().item() for c in sorted(set(cost.tolist()))}
        print(f"seed {sd} rep {rep}: max {rel.max().item():.2e} walker {i} class {int(cost[i])} E_loc {e[i].item():.4f} tight {tight[i].item():.4f} | " +
              " ".join(f"{c}:{v:.1e}" for c, v in by.items() if v > 5e-7), flush=True)
        # the same walkers, stand-alone call at one tolerance
    a = native.eloc(tu, td, 3, 3, model.cnf.v_wrapper.v.net(), x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True)["eloc"]
    rel = (a - tight).abs() / tight.abs()
    print(f"   stand-alone one-tolerance call: max {rel.max().item():.2e} walker {int(rel.argmax())} class {int(cost[int(rel.argmax())])}")
