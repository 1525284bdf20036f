"""Probe: RHS evaluations and rejected steps of the adjoint pass (and its stage time) for a given opening-step rule.
usage: python tools/probes/adjoint_open.py [nup ndown d batch iters]      (rule: see GSVMC._sweep; FERMIFLOW_ADJ_* in the environment)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fermiflow_amd as ff
import __graft_entry__ as Gm
nup, ndn, d, B, iters = (int(a) for a in (sys.argv[1:6] + ["6", "6", "2", "32768", "6"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
gs = Gm._model(dev, 2, 2, 2.0)
orb = ff.HO3D() if d == 3 else ff.HO2D()
model = ff.GSVMC(nup, ndn, orb, ff.FreeFermion(device=dev), gs.cnf, ff.CoulombPairPotential(2.0), sp_potential=ff.HO())
model.prefetch_walkers = False
if os.environ.get("FF_SENS_BITS"):
    from fermiflow_amd import native
    native.set_sens_precision(int(os.environ["FF_SENS_BITS"]))
torch.manual_seed(0)
gps = []
for it in range(iters):
    model.profile = {}
    g = model(B); g.backward()
    torch.cuda.synchronize()
    ev = model.profile["events"][0]
    sa, se = model.profile["adjoint_stats"][0], model.profile["eloc_stats"][0]
    gp = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
    gps.append(gp.clone())
    for p in model.parameters():
        p.grad = None
    print(f"iter {it}: eloc evals {se[0].item() / B:.2f}  adjoint evals {sa[0].item() / B:.2f} rejected/walker {sa[2].item() / B:.4f} max steps {int(sa[1])} fail {int(sa[3])}  "
          f"adjoint stage {ev['estimator'].elapsed_time(ev['adjoint']):.3f} ms  |grad| {gp.norm().item():.9e}", flush=True)
