"""Probe: is a walker's first-step rejection in the local-energy pass predicted by how little the equal-step rounding took off its
step?  margin = (planned step) / (flow step x class factor) in (k / (k + 1), 1]: 1 = nothing taken off.
usage: python tools/probes/reject_margin.py [weight set of tests/golden/trained_weights.npz | head]"""
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
torch.manual_seed(5)
for it in range(12):
    with torch.no_grad():
        model(65536)
torch.cuda.synchronize()
cost, hs, he = (t.cpu() for t in model._h_prev)
hval = model._hg_last.cpu()
tab = model._h_tab[1 - model._h_tab_cur].cpu()      # the table the last pass applied
cls = cost.clamp(0, 31).long()
hq = hval * tab[cls]
margin = hs / hq
rej = he < 0.999 * hs
k = torch.round(1.0 / hs)
print(f"{tag}: rejected first steps {rej.double().mean():.3f}")
for c in range(2, 12):
    sel = cls == c
    if sel.sum() < 200:
        continue
    line = f"class {c:2d} n={int(sel.sum()):6d} k={k[sel].mean():.2f} rej={rej[sel].double().mean():.3f} | by margin quartile:"
    q = torch.quantile(margin[sel].double(), torch.tensor([0.25, 0.5, 0.75], dtype=torch.float64))
    edges = [0.0] + q.tolist() + [2.0]
    for a, b in zip(edges[:-1], edges[1:]):
        s2 = sel & (margin > a) & (margin <= b)
        line += f" ({a:.2f},{b:.2f}]: {rej[s2].double().mean():.3f}"
    print(line)
# what grouping by margin would buy: waves of four in schedule order (class, then margin) against (class, index)
for name, keyf in (("class only", lambda: cls.double()), ("class, then margin", lambda: cls.double() + 0.5 * (margin > margin.median()).double())):
    order = torch.argsort(keyf(), stable=True)
    r4 = rej[order][: (len(order) // 4) * 4].view(-1, 4).any(dim=1).double().mean()
    print(f"waves of four with a rejection, ordered by {name}: {r4:.3f}")
