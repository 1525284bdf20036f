"""Probe: RHS evaluations per walker of the local-energy pass and the flow pass's cost classes over a long run."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
from fermiflow_amd.utils import make_adam
dev = torch.device("cuda:0")
B = 65536
m = G._model(dev, 3, 3, 2.0)
opt = make_adam(m.parameters(), lr=float(sys.argv[1]) if len(sys.argv) > 1 else 2e-5)
torch.manual_seed(1234)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
for it in range(N):
    probe = it in (0, 1, 2, 5, 10, 20, 50, 100, 150, 200, 300, 399, 600, 999, 1499)
    m.profile = {} if probe else None
    g = m(B); opt.zero_grad(); g.backward(); opt.step()
    if probe:
        st = m.profile["eloc_stats"][0]
        torch.cuda.synchronize()
        print("it %4d  eloc evals/walker %.2f  max accepted %d  E %.4f  h_flow %.4f" % (it, st[0].item() / B, st[1].item(), m.E, float(m._h_flow.mean())), flush=True)
