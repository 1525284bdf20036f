"""Probe (soak): the reference's training run at config 2 -- init_zeros() flow, Adam lr 1e-2 (src/FermionHO2D.py:40-43,61-72) -- for many
iterations: E, evaluations per walker of both passes, failed integrations, ms per iteration by window.  Run under `timeout`.
usage: python tools/probes/driver_soak.py [iterations] [window]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
from fermiflow_amd.utils import make_adam
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
win = int(sys.argv[2]) if len(sys.argv) > 2 else 250
B = 65536
m = G._model(dev, 3, 3, 2.0)
v = m.cnf.v_wrapper.v
v.eta.init_zeros(); v.mu.init_zeros(); m.to(dev)
opt = make_adam(m.parameters(), lr=1e-2)
torch.manual_seed(1234)
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(1, N + 1):
    probe = it % win == 0
    m.profile = {"stages": False} if probe else None
    g = m(B); opt.zero_grad(); g.backward(); opt.step()
    if probe:
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / win * 1e3
        se, sa = m.profile["eloc_stats"][0], m.profile["adjoint_stats"][0]
        ok = all(torch.isfinite(p).all().item() for p in m.parameters())
        print(f"it {it:5d}: {dt:.3f} ms/iteration  E {m.E:.4f} +- {m.E_std / B ** 0.5:.4f}  eloc evals {se[0].item() / B:.2f} (failed {int(se[3])})  "
              f"adjoint evals {sa[0].item() / B:.2f} (rejected/walker {sa[2].item() / B:.3f}, failed {int(sa[3])})  parameters finite {ok}  max|w1| {max(v.eta.fc1.weight.abs().max().item(), v.mu.fc1.weight.abs().max().item()):.2f}", flush=True)
        t0 = time.perf_counter()
