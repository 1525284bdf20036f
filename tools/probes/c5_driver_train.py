"""Probe: configs[4] shape (10 + 10 particles, d = 3) trained from init_zeros() at lr 1e-2 as the reference's driver does -- time per iteration, E,
weights; (a run of this took > 25 minutes for 600 iterations of 16 384 walkers in round 5: what happens?)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
import fermiflow_amd as ff
from fermiflow_amd.utils import make_adam
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
gs = G._model(dev, 2, 2, 2.0)
model = ff.GSVMC(10, 10, ff.HO3D(), ff.FreeFermion(device=dev), gs.cnf, ff.CoulombPairPotential(2.0), sp_potential=ff.HO())
v = model.cnf.v_wrapper.v
v.eta.init_zeros(); v.mu.init_zeros(); model.to(dev)
opt = make_adam(model.parameters(), lr=1e-2)
torch.manual_seed(1234)
model.profile = {"stages": False}
for i in range(1, 601):
    t = time.time()
    g = model(B); opt.zero_grad(); g.backward(); opt.step()
    torch.cuda.synchronize()
    st = model.profile["eloc_stats"][-1]
    if i <= 5 or i % 10 == 0 or time.time() - t > 0.5:
        print(f"iter {i}: {1e3 * (time.time() - t):.1f} ms  E {model.E:.4f} E_std {model.E_std:.3f}  evals/walker {st[0].item() / B:.1f} max steps {st[1].item()} rej {st[2].item() / B:.2f} fail {st[3].item()} "
              f"max|w1| {v.eta.fc1.weight.abs().max().item():.3f} {v.mu.fc1.weight.abs().max().item():.3f} max|w2| {v.eta.fc2.weight.abs().max().item():.3f} {v.mu.fc2.weight.abs().max().item():.3f}", flush=True)
    model.profile = {"stages": False}
