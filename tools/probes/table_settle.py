"""Probe: wave-level evaluations per walker of the local-energy pass on fixed weights once the learned first-step table has settled
(40 sweeps of fresh walkers; mean of the last 20) -- for A/B builds of the table's update rule.
usage: python tools/probes/table_settle.py [head|trained|driver|driver1000]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as Gm
dev = torch.device("cuda:0")
tag = sys.argv[1] if len(sys.argv) > 1 else "trained"
model = Gm._model(dev, 3, 3, 2.0)
if tag != "head":
    W = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "trained_weights.npz"))
    v = model.cnf.v_wrapper.v
    with torch.no_grad():
        for nm, m in (("eta", v.eta), ("mu", v.mu)):
            m.fc1.weight.copy_(torch.as_tensor(W[f"{tag}_{nm}_w1"]).reshape(-1, 1))
            m.fc1.bias.copy_(torch.as_tensor(W[f"{tag}_{nm}_b1"]))
            m.fc2.weight.copy_(torch.as_tensor(W[f"{tag}_{nm}_w2"]).reshape(1, -1))
torch.manual_seed(77)
ev, ms = [], []
for it in range(40):
    model.profile = {"stages": False}
    with torch.no_grad():
        model(65536)
    pr, model.profile = model.profile, None
    torch.cuda.synchronize()
    ev.append(int(pr["eloc_stats"][0][0].item()) / 65536)
    ms.append(pr["pass1"][0][0].elapsed_time(pr["pass1"][0][1]))
print(f"{tag}: evaluations per walker (wave level) {np.mean(ev[20:]):.2f}, pass {np.mean(ms[20:]):.3f} ms, rejected/walker {int(pr['eloc_stats'][0][2].item()) / 65536:.3f}; factors 2..12: "
      + " ".join(f"{x:.2f}" for x in model._h_tab[model._h_tab_cur][2:13].tolist()))
