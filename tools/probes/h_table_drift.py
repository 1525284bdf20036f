"""Probe: the learned first-step table of the local-energy pass over a long run (config 2) -- evaluations per walker, the factors by cost
class, the classes' populations and first-step rejection rates.    python tools/probes/h_table_drift.py [lr] [iterations]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
from fermiflow_amd.utils import make_adam
dev = torch.device("cuda:0")
B = 65536
m = G._model(dev, 3, 3, 2.0)
opt = make_adam(m.parameters(), lr=float(sys.argv[1]) if len(sys.argv) > 1 else 2e-5)
torch.manual_seed(1234)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
every = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for it in range(N):
    probe = it < 3 or it % every == 0 or it == N - 1
    m.profile = {"stages": False} if probe else None
    g = m(B); opt.zero_grad(); g.backward(); opt.step()
    if probe:
        st = m.profile["eloc_stats"][0]
        torch.cuda.synchronize()
        sa = m.profile["adjoint_stats"][0]
        line = "it %4d evals %.2f max-acc %d E %.4f hflow %.4f adjoint evals %.2f rej/walker %.3f" % (
            it, st[0].item() / B, st[1].item(), m.E, float(m._h_flow.mean()), sa[0].item() / B, sa[2].item() / B)
        if m._h_tab is not None and m._h_prev is not None:
            tab = m._h_tab[m._h_tab_cur].cpu()
            c, hs, he = (t.cpu() for t in m._h_prev)
            ok = (hs > 0) & (he > 0)
            cls = c.clamp(0, 31)
            parts = []
            for k in range(0, 14):
                sel = ok & (cls == k)
                nk = int(sel.sum())
                if nk >= 64:
                    rej = float((he[sel] < 0.999 * hs[sel]).double().mean())
                    steps = float((1.0 / hs[sel]).mean())
                    kw = torch.round(1.0 / hs[sel])
                    ev = float(((kw < 1.5) | (he[sel] >= 0.999 / (kw - 1).clamp(min=1))).double().mean())      # accepted the step of a plan one shorter
                    parts.append("c%d n=%d f=%.3f rej=%.2f k=%.2f ev=%.2f" % (k, nk, tab[k], rej, steps, ev))
            line += " | " + "; ".join(parts)
        print(line, flush=True)
