"""Probe: the first iterations of the driver's run (init_zeros, Adam lr 1e-2, src/FermionHO2D.py:40-43,61-72) one by one: time, evaluations, which kernels."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
from fermiflow_amd.utils import make_adam
dev = torch.device("cuda:0")
B = 65536
warm = G._model(dev, 3, 3, 2.0); warm(B).backward(); torch.cuda.synchronize()      # kernels loaded, allocator warm
m = G._model(dev, 3, 3, 2.0)
v = m.cnf.v_wrapper.v
v.eta.init_zeros(); v.mu.init_zeros(); m.to(dev)
opt = make_adam(m.parameters(), lr=1e-2)
torch.manual_seed(4321)
for i in range(1, 16):
    m.profile = {"stages": False}
    torch.cuda.synchronize(); t = time.perf_counter()
    g = m(B); opt.zero_grad(); g.backward(); opt.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    st = m.profile["eloc_stats"][-1]; p1 = m.profile["pass1"][-1]
    tab = m.cnf.v_wrapper.v.net().t[-1][:6].tolist()
    print(f"iter {i}: {dt * 1e3:.2f} ms  eloc pass {p1[0].elapsed_time(p1[1]):.3f} ms evals {st[0].item() / B:.1f} max steps {st[1].item()} rej {st[2].item() / B:.2f}  E {m.E:.4f}  "
          f"max|w1| {v.eta.fc1.weight.abs().max().item():.4f} max|w2| {v.eta.fc2.weight.abs().max().item():.4f}  table header {['%.3g' % x for x in tab]}", flush=True)
