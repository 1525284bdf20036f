"""Probe: attempted steps per walker of the sweep's local-energy pass (by cost class), and what the wave-level lockstep of the four-walkers-per-wave
kernel costs: evaluations if every walker ran alone vs by waves of four in schedule order."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
from fermiflow_amd import native
dev = torch.device("cuda:0")
B = 65536
model = G._model(dev, 3, 3, 2.0)
if len(sys.argv) > 1 and sys.argv[1] != "head":      # a weight set of tests/golden/trained_weights.npz: trained | driver | driver1000
    W = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "trained_weights.npz"))
    v = model.cnf.v_wrapper.v
    with torch.no_grad():
        for nm, m in (("eta", v.eta), ("mu", v.mu)):
            m.fc1.weight.copy_(torch.as_tensor(W[f"{sys.argv[1]}_{nm}_w1"]).reshape(-1, 1))
            m.fc1.bias.copy_(torch.as_tensor(W[f"{sys.argv[1]}_{nm}_b1"]))
            m.fc2.weight.copy_(torch.as_tensor(W[f"{sys.argv[1]}_{nm}_w2"]).reshape(1, -1))
cap = {}
orig = native.eloc
def eloc(*a, **k):
    cap["steps"] = torch.zeros(B, dtype=torch.int32, device=dev)
    cap["order"] = k.get("walker_order")
    cap["hs"] = k.get("walker_h_init")
    k["walker_cost"] = cap["steps"]
    k["want_stats"] = True
    r = orig(*a, **k)
    cap["stats"] = r["stats"]
    return r
native.eloc = eloc
torch.manual_seed(5)
for it in range(12):
    with torch.no_grad():
        model(B)
steps, order, cost = cap["steps"].cpu().numpy(), cap["order"].cpu().numpy(), model.walker_cost.cpu().numpy()
hs = cap["hs"].cpu().numpy()
print("wave-level evals/walker (kernel statistic):", cap["stats"][0].item() / B, "rejected/walker", cap["stats"][2].item() / B)
print("attempted steps per walker: " + " ".join(f"{s}:{(steps == s).sum()}" for s in range(1, 12)))
light = cost < 16
so = steps[order]
lo = light[order]
w = so[lo][: (lo.sum() // 4) * 4].reshape(-1, 4)
print("light walkers: mean attempted steps", steps[light].mean(), "-> evals alone", 6 * steps[light].mean() + 1, "| by waves of four in schedule order: mean of wave max", w.max(1).mean(), "-> evals", 6 * w.max(1).mean() + 1)
print("waves by (min, max) steps:", {f"{a}-{b}": int(((w.min(1) == a) & (w.max(1) == b)).sum()) for a in range(1, 6) for b in range(a, 7) if ((w.min(1) == a) & (w.max(1) == b)).any()})
for c in range(2, 13):
    m = cost == c
    if m.any():
        print(f"class {c}: n {m.sum()} steps mean {steps[m].mean():.2f}  hist " + " ".join(f"{s}:{(steps[m] == s).sum()}" for s in range(1, 8)) + f"  first step {hs[m].mean():.3f}")
