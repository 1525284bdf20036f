"""Development probe: achieved HBM rate of the stand-alone one-lane-per-walker kernels (API entry points outside the
sweep): ff_logprob, ff_slater_logabsdet_fwd/_bwd, ff_potential, ff_backflow_v_div."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
from fermiflow_amd import native
dev = torch.device("cuda:0")
model = G._model(dev, 3, 3, 2.0)
tu, td = model._tables(dev)
B = 1 << 20
x = torch.randn(B, 6, 2, dtype=torch.float64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(name, fn, nbytes, reps=5):
    fn(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-28s %.3f ms  %.2f TB/s" % (name, ms, nbytes / ms / 1e9))


timed("potential", lambda: native.potential(x, 2.0, True), B * 104)
timed("logprob (value)", lambda: native.logprob(tu, td, 3, 3, x, need_grad=False, need_lap=False) if "need_grad" in native.logprob.__code__.co_varnames else native.logprob(tu, td, 3, 3, x), B * 104)
xs = x[:, :3].contiguous()
timed("slater fwd (3x3)", lambda: native.slater_fwd(tu, xs), B * 56)
go = torch.ones(B, dtype=torch.float64, device=dev)
timed("slater bwd (3x3)", lambda: native.slater_bwd(tu, xs, go), B * (48 + 8 + 48))
