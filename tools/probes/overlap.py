"""Probe: does the Metropolis kernel of the NEXT sweep run beside the adjoint of this one?  Times ff_mcmc_sample and
ff_cnf_adjoint alone and launched together on two streams (config 2, 65536 walkers)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
from fermiflow_amd import native
dev = torch.device("cuda:0")
B = 65536
model = G._model(dev, 3, 3, 2.0)
net = model.cnf.v_wrapper.v.net()
tu, td = model._tables(dev)
z, _, _ = native.mcmc_sample(tu, td, 3, 3, B, 100, 0.1, 1, dev)
hg = torch.zeros(B, dtype=torch.float64, device=dev); he = torch.zeros_like(hg)
x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, walker_h_out=hg)
r = native.eloc(tu, td, 3, 3, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, walker_h_init=hg, walker_h_scale=0.6, walker_h_out=he)
E = r["eloc"].mean().reshape(1)
s2 = torch.cuda.Stream()

def adj():
    return native.cnf_adjoint(net, r["z"], r["glogp0"], None, 0.0, 1.0, 1e-6, 1e-8, need_gx=False, energy=(r["eloc"], E, 1.0 / B),
                              walker_h_init=he, walker_h_scale=1.25)

def mc():
    return native.mcmc_sample(tu, td, 3, 3, B, 100, 0.1, 7, dev)

def timed(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)

def both():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        mc()
    adj()

def both_adj_first():
    ev = torch.cuda.Event()
    adj()
    with torch.cuda.stream(s2):
        mc()

lo, hi = -1, 0
try:
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so"); a, b = ctypes.c_int(), ctypes.c_int(); hip.hipDeviceGetStreamPriorityRange(ctypes.byref(a), ctypes.byref(b)); lo, hi = a.value, b.value
except Exception as e:
    print("priority range: n/a", e)
s_low = torch.cuda.Stream(priority=max(lo, hi))     # numerically largest = least priority

def both_low_prio():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s_low):
        s_low.wait_event(ev)
        mc()
    adj()

print(os.environ.get("FERMIFLOW_LIB", "default")[-8:], "adjoint %.3f | mcmc %.3f | mcmc first %.3f | adjoint first %.3f | mcmc on low-priority stream (range %d..%d) %.3f ms" % (
    timed(adj), timed(mc), timed(both), timed(both_adj_first), lo, hi, timed(both_low_prio)), flush=True)
