"""Development probe: does sampling the NEXT sweep's walkers on a second stream (the base distribution does not depend
on the parameters) hide the Metropolis kernel behind the current sweep's kernels?
Measured (65 536 walkers): baseline 3.41 ms/iter, prefetch 3.38-3.49 -- no: the sweep's kernels leave no idle issue slots
that a co-resident Metropolis wave could use (register files are full), so the idea was dropped."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 65536))


def run(prefetch, where):
    model = G._model(dev, 3, 3, 2.0)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    side = torch.cuda.Stream()
    state = {"z": None, "evt": None}
    orig = model.basedist.sample

    def launch_next():
        z_evt = torch.cuda.Event()
        side.wait_stream(torch.cuda.current_stream()) if where == "after" else None
        with torch.cuda.stream(side):
            z = orig(model.orbitals_up, model.orbitals_down, (B,))
            z_evt.record(side)
        state["z"], state["evt"] = z, z_evt

    def sample(up, down, shape, **kw):
        if prefetch and state["z"] is not None:
            torch.cuda.current_stream().wait_event(state["evt"])
            z = state["z"]; z.record_stream(torch.cuda.current_stream())
            state["z"] = None
            if where == "start": launch_next()
            return z
        z = orig(up, down, shape, **kw)
        if prefetch and where == "start": launch_next()
        return z
    model.basedist.sample = sample

    def step():
        g = model(B); opt.zero_grad(); g.backward(); opt.step()
        if prefetch and where in ("end", "after"): launch_next()
    for _ in range(5): step()
    torch.cuda.synchronize()
    N = 30
    t0 = time.perf_counter()
    for _ in range(N): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3, model.E


print("baseline            %.3f ms/iter  E=%.6f" % run(False, ""))
for w in ("start", "end", "after"):
    print("prefetch (%-5s)    %.3f ms/iter  E=%.6f" % ((w,) + run(True, w)))
