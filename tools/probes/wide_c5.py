"""Probe (not part of the product): stage times of one training iteration with the one-walker-per-workgroup kernels.
usage: python tools/probes/wide_c5.py [nup ndown d batch iters]   (default: 10 10 3 16384 3 = BASELINE configs[4] shape)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fermiflow_amd as ff
import __graft_entry__ as Gm

nup, ndn, d, B, iters = (int(a) for a in (sys.argv[1:6] + ["10", "10", "3", "16384", "3"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
gs = Gm._model(dev, 2, 2, 2.0)
orb = ff.HO3D() if d == 3 else ff.HO2D()
model = ff.GSVMC(nup, ndn, orb, ff.FreeFermion(device=dev), gs.cnf, ff.CoulombPairPotential(2.0), sp_potential=ff.HO())
model.prefetch_walkers = False
if os.environ.get("FF_SENS_BITS"):
    from fermiflow_amd import native
    native.set_sens_precision(int(os.environ["FF_SENS_BITS"]))
torch.manual_seed(0)
for it in range(iters):
    model.profile = {}
    torch.cuda.synchronize()
    t0 = time.time()
    g = model(B)
    g.backward()
    torch.cuda.synchronize()
    t1 = time.time()
    ev = model.profile["events"][0]
    names = ["t0", "mcmc", "generate", "eloc", "estimator", "adjoint"]
    st = {names[k + 1]: ev[names[k]].elapsed_time(ev[names[k + 1]]) for k in range(len(names) - 1)}
    p1 = model.profile["pass1"][0]
    stats = model.profile["eloc_stats"][0].tolist()
    print(f"iter {it}: {1e3 * (t1 - t0):8.2f} ms  E = {model.E:.6f} +- {model.E_std:.4f}  stages(ms) " +
          " ".join(f"{k}={v:.2f}" for k, v in st.items()) + f"  sens-kernel={p1[0].elapsed_time(p1[1]):.2f} ms  evals/walker={stats[0] / B:.2f} "
          f"max steps={stats[1]} rejected={stats[2]} fail={stats[3]}", flush=True)
    if os.environ.get("FERMIFLOW_LIB", "").endswith("stamps.so"):
        st8 = model.profile["eloc_stats"][0][8:26].view(torch.int64)[:9].double()      # (stamp 8: behind ff_dp5_consume2)
        names8 = ["publish", "radii + S product", "heads + records", "S to LDS + row lanes", "J product", "R2", "2nd-order sums", "stage coefficients (+ walker prologue)", "consume"]
        print("   ticks per evaluation: " + "  ".join(f"{nm} {v / stats[0]:.0f}" for nm, v in zip(names8, st8.tolist())) + f"   total {st8.sum().item() / stats[0]:.0f}")
