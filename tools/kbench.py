#!/usr/bin/env python3
"""Per-kernel timing harness (development tool): times the three fused ODE kernels and the MCMC kernel at the
BASELINE configuration with HIP events.  python tools/kbench.py [--B 65536] [--reps 5]"""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as G
from fermiflow_amd import native

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=65536)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--nup", type=int, default=3)
ap.add_argument("--ndown", type=int, default=3)
ap.add_argument("--tag", default="")
a = ap.parse_args()
dev = torch.device("cuda:0")
model = G._model(dev, a.nup, a.ndown, 2.0)
net = model.cnf.v_wrapper.v.net()
tu, td = model._tables(dev)
n = a.nup + a.ndown
z, _, _ = native.mcmc_sample(tu, td, a.nup, a.ndown, a.B, 100, 0.1, 1, dev)


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps, out


res = {"tag": a.tag, "B": a.B}
res["mcmc_ms"], _ = timeit(lambda: native.mcmc_sample(tu, td, a.nup, a.ndown, a.B, 100, 0.1, 1, dev))
steps = torch.empty(a.B, dtype=torch.int32, device=dev)
sched = not os.environ.get("FF_NO_SCHED")
warm = not os.environ.get("FF_NO_WARM")
hg = torch.zeros(a.B, dtype=torch.float64, device=dev); he = torch.zeros_like(hg)
x0 = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, walker_h_out=hg)
hprev = hg.clone() if warm else None
res["generate_ms"], (x, st) = timeit(lambda: native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, want_stats=True, walker_cost=steps,
                                                                 walker_h_init=hprev, walker_h_scale=0.75, walker_h_out=hg))
wk = dict(walker_h_init=hg, walker_h_scale=0.6, walker_h_out=he) if warm else {}
wa = dict(walker_h_init=he, walker_h_scale=1.25) if warm else {}
res["order_ms"], order = timeit(lambda: native.walker_order(steps))
if not sched:
    order = None
res["generate_evals"] = st[0].item() / a.B
res["logp_ms"], (_, _, st) = timeit(lambda: native.cnf_delta_logp(net, x, 0.0, 1.0, 1e-6, 1e-8, want_stats=True))
res["logp_evals"] = st[0].item() / a.B
ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
steps2 = torch.empty_like(steps)
r = native.eloc(tu, td, a.nup, a.ndown, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True, walker_cost=steps2, walker_order=order, **wk)
order2 = native.walker_order(steps2) if sched else None
res["steps_corr"] = float(torch.corrcoef(torch.stack([steps.double(), steps2.double()]))[0, 1])
res["steps_max"] = [int(steps.max()), int(steps2.max())]
tot = 0.0
for _ in range(a.reps):
    r = native.eloc(tu, td, a.nup, a.ndown, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True, pass1_events=ev, walker_order=order, **wk)
    torch.cuda.synchronize(); tot += ev[0].elapsed_time(ev[1])
res["eloc_pass1_ms"] = tot / a.reps
res["eloc_evals"] = r["stats"][0].item() / a.B
res["eloc_rej"] = r["stats"][2].item() / a.B
w = (r["eloc"] - r["eloc"].mean()) / a.B
res["adjoint_ms"], (_, gp, st) = timeit(lambda: native.cnf_adjoint(net, r["z"], w[:, None, None] * r["glogp0"], -w, 0.0, 1.0, 1e-6, 1e-8,
                                                                   need_gx=False, want_stats=True, walker_order=order2, **wa))
res["adjoint_evals"] = st[0].item() / a.B
sa = st[8:26].view(torch.int64)[:9].double()
if sa.sum() > 0:   # 0 stage input, 1 publish, 2 radius, 3 component, 4 consume, 5 consume of stage 6 (deposit), per wave-eval
    res["adjoint_stamps_ticks"] = [round(v) for v in (sa / (st[0].item() / 5.0)).tolist()]
st8 = r["stats"][8:26].view(torch.int64)[:9].double()
GW = 4.0 if 4 <= n <= 6 else float(max(1, 64 // (2 * n)))      # walkers per wave of the local-energy kernel (matrix-core kernel: four)
if st8.sum() > 0:
    res["stamps_pct"] = [round(v, 1) for v in (100 * st8 / st8.sum()).tolist()]
    res["stamps_ticks"] = [round(v) for v in (st8 / (r["stats"][0].item() / GW)).tolist()]
    res["core_clock_GHz"] = r["stats"][26].item() / max(1, r["stats"][27].item()) * 0.1
    res["stamp_ticks_per_wave_eval"] = st8.sum().item() / (r["stats"][0].item() / GW)   # 0 coef, 1 publish, 2 radius, 3 transpose(after sweep), 4 -, 5 consume, 6 form in[], 7 sweep
if r["stats"].numel() > 32:   # per-workgroup trace of the local-energy kernel (FF_STAMPS_TRACE build)
    import numpy as np
    nb = min(16384, (r["stats"].numel() - 32) // 4)
    t = r["stats"][32:32 + 4 * nb].view(nb, 4).cpu().numpy().astype(np.int64)
    t = t[t[:, 1] != 0]
if r["stats"].numel() >= 65600 + 36:      # consume-phase ticks by Dormand-Prince stage (-2 .. 6), per wave-evaluation in that stage
    sa = r["stats"][65600:65636].view(torch.int64).double()
    res["consume_ticks_by_stage(-2..6)"] = [round(float(a / max(1.0, float(c)))) for a, c in zip(sa[:9], sa[9:])]
    res["consume_evals_by_stage(-2..6)"] = [int(c) for c in sa[9:]]
if r["stats"].numel() > 32 and len(t):
    t0 = t[:, 0].min()
    st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0      # microseconds
    dur = en - st
    res["trace"] = {"blocks": int(len(t)), "span_us": float(en.max()), "dur_us_mean": float(dur.mean()), "dur_us_p95": float(np.percentile(dur, 95)),
                    "core_GHz_mean": float((t[:, 2] / (dur * 1e3 + 1e-9)).mean()),
                    "concurrency_mean": float(dur.sum() / en.max()),
                    "start_us_percentiles": [float(np.percentile(st, q)) for q in (10, 50, 90, 99)]}
    hw = t[:, 3]
    xcc = (hw >> 16) & 0xf
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7 if False else (hw >> 12) & 0xf
    res["trace"]["per_xcc_blocks"] = np.bincount(xcc, minlength=8).tolist()
    res["trace"]["per_xcc_mean_dur"] = [round(float(dur[xcc == k].mean()), 1) if (xcc == k).any() else 0 for k in range(8)]
    np.save(os.path.join(ROOT, "gpurun_out", "trace.npy"), t)
res["E"] = r["eloc"].mean().item()
res["gp_norm"] = gp.norm().item()
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}))
