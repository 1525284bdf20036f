"""Development probe: step-size warm start (ff_ode.walker_h_init/_scale/_out) -- evaluations, rejections, time and
accuracy against a tight-tolerance solve, for a range of scale factors.  python tools/probes/warm_start.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
from fermiflow_amd import native
dev = torch.device("cuda:0")
nup = int(os.environ.get("NUP", 3)); ndn = int(os.environ.get("NDN", 3)); B = int(os.environ.get("B", 65536))
model = G._model(dev, nup, ndn, 2.0)
net = model.cnf.v_wrapper.v.net()
tu, td = model._tables(dev)
z, _, _ = native.mcmc_sample(tu, td, nup, ndn, B, 100, 0.1, 1, dev)
f = dict(dtype=torch.float64, device=dev)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


hg = torch.zeros(B, **f); cost = torch.zeros(B, dtype=torch.int32, device=dev)
tg, (x, st) = timed(lambda: native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, want_stats=True, walker_cost=cost, walker_h_out=hg))
order = native.walker_order(cost)
print("generate cold: %.3f ms evals %.2f  h_out median %.3f min %.3f max %.3f" % (tg, st[0].item() / B, hg.median(), hg.min(), hg.max()))
xt = native.cnf_generate(net, z, 0.0, 1.0, 1e-11, 1e-13)
print("   x err vs tight: %.2e" % (x - xt).abs().max())
for sc in (0.5, 0.75, 1.0):
    hprev = torch.full((B,), float(hg.median()), **f)
    tw, (xw, st) = timed(lambda: native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, want_stats=True, walker_h_init=hprev, walker_h_scale=sc))
    print("generate warm (median of previous x %.2f): %.3f ms evals %.2f rej %.3f  x err %.2e" % (sc, tw, st[0].item() / B, st[2].item() / B, (xw - xt).abs().max()))

ref = native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-11, 1e-13, 2.0, True)
he = torch.zeros(B, **f)
tc, rc = timed(lambda: native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True, walker_order=order, walker_h_out=he))
rel = lambda r: ((r["eloc"] - ref["eloc"]).abs() / ref["eloc"].abs())
print("eloc cold: %.3f ms evals %.2f rej %.3f  E_loc rel err max %.2e  p99.9 %.2e   h_out median %.3f" % (tc, rc["stats"][0].item() / B, rc["stats"][2].item() / B, rel(rc).max(), rel(rc).quantile(0.999), he.median()))
for sc in (0.3, 0.4, 0.5, 0.6, 0.8, 1.0):
    he2 = torch.zeros(B, **f)
    tw, rw = timed(lambda: native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True, walker_order=order,
                                        walker_h_init=hg, walker_h_scale=sc, walker_h_out=he2))
    print("eloc warm x%.1f: %.3f ms evals %.2f rej %.3f  E_loc rel err max %.2e p99.9 %.2e mean E diff %.2e" % (
        sc, tw, rw["stats"][0].item() / B, rw["stats"][2].item() / B, rel(rw).max(), rel(rw).quantile(0.999),
        abs(rw["eloc"].mean() - ref["eloc"].mean()) / abs(ref["eloc"].mean())))

w = (ref["eloc"] - ref["eloc"].mean()) / B
args = (ref["z"], w[:, None, None] * ref["glogp0"], -w, 0.0, 1.0)
_, gref = native.cnf_adjoint(net, *args, 1e-11, 1e-13, need_gx=False)
tc, (_, gc, st) = timed(lambda: native.cnf_adjoint(net, *args, 1e-6, 1e-8, need_gx=False, want_stats=True))
print("adjoint cold: %.3f ms evals %.2f rej %.3f  grad rel err %.2e" % (tc, st[0].item() / B, st[2].item() / B, (gc - gref).norm() / gref.norm()))
for sc in (0.5, 0.75, 1.0, 1.5, 2.0):
    tw, (_, gw, st) = timed(lambda: native.cnf_adjoint(net, *args, 1e-6, 1e-8, need_gx=False, want_stats=True, walker_h_init=he, walker_h_scale=sc))
    print("adjoint warm x%.2f: %.3f ms evals %.2f rej %.3f  grad rel err %.2e" % (sc, tw, st[0].item() / B, st[2].item() / B, (gw - gref).norm() / gref.norm()))
