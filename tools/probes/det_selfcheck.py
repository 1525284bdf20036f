"""-DFF_WIDE_SELFCHECK build (every right-hand side of the wide local-energy kernels evaluated twice and compared): which component of
which lane differs between two evaluations of the same inputs?   FERMIFLOW_LIB=.../libfermiflow_hip_sc.so python tools/probes/det_selfcheck.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as Gm
import fermiflow_amd as ff
from fermiflow_amd import native
dev = torch.device("cuda:0")
NRUN = int(os.environ.get("NRUN", "100"))
def setup(nup, ndn, dim, bits):
    if dim == 2:
        model = Gm._model(dev, nup, ndn, 2.0)
    else:
        gs = Gm._model(dev, 2, 2, 2.0)
        model = ff.GSVMC(nup, ndn, ff.HO3D(), ff.FreeFermion(device=dev), gs.cnf, ff.CoulombPairPotential(2.0), sp_potential=ff.HO())
    v = model.cnf.v_wrapper.v
    torch.manual_seed(31 + nup)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (200,))
    net = v.net(refresh=True)
    x = native.cnf_generate(net, z, 0.0, 1.0, 1e-8, 1e-10)
    tu, td = model._tables(dev)
    def run():
        pb = native.set_sens_precision(bits)
        try:
            return native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-8, 1e-10, 2.0, True, want_stats=True)
        finally:
            native.set_sens_precision(pb)
    return run
shapes = [(7, 6, 2, 64), (7, 6, 2, 32), (12, 12, 2, 64), (10, 10, 3, 64), (10, 10, 3, 32)]
runs = {s: setup(*s) for s in shapes}
events = collections.defaultdict(list)
evs = collections.defaultdict(list)
for it in range(NRUN):
    for s in shapes:
        r = runs[s]()
        st = r["stats"].tolist()
        evs[s].append(st[0])
        if st[8] > 0:
            import struct
            f = lambda i: struct.unpack("f", struct.pack("i", i))[0]
            events[s].append((st[8], [tuple(st[9 + 9 * k: 14 + 9 * k]) + tuple(f(v) for v in st[14 + 9 * k: 18 + 9 * k]) for k in range(min(st[8], 2))]))
for s in shapes:
    m = max(set(evs[s]), key=evs[s].count)
    print(s, ": runs off the modal evaluation count", sum(e != m for e in evs[s]), "of", NRUN, "; runs with a self-check mismatch", len(events[s]))
    for e in events[s][:6]:
        print("     mismatches in the launch:", e[0], " first (lane, component [100+: J element], stage, walker, agree 1: first=second 2: second=third 4: first=third, the three values, rel. diff first-third):", e[1])
