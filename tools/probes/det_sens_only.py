"""Reproducibility of the sensitivity pass ALONE (ff_eloc_sensitivities, no finish kernels behind it) against the whole ff_eloc_nd."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
import __graft_entry__ as Gm
import fermiflow_amd as ff
from fermiflow_amd import native, _lib as L
dev = torch.device("cuda:0")
NRUN = int(os.environ.get("NRUN", "100"))
def setup(nup, ndn, dim):
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
    return net, x, tu, td
for (nup, ndn, dim) in ((12, 12, 2), (10, 10, 3), (7, 6, 2)):
    net, x, tu, td = setup(nup, ndn, dim)
    B, n = x.shape[0], nup + ndn
    M = n * dim
    nb = L.lib().ff_eloc_workspace_bytes(L.i64(B), n, dim)
    for mode in ("sensitivities only", "whole ff_eloc_nd"):
        evs, zs = [], []
        for it in range(NRUN):
            if mode == "sensitivities only":
                ws = torch.empty(nb // 8, dtype=torch.float64, device=dev)
                st = torch.zeros(32, dtype=torch.int32, device=dev)
                o = L.ode(0.0, 1.0, 1e-8, 1e-10)
                L.check(L.lib().ff_eloc_sensitivities(L.stream(), L.i64(B), n, dim, net.ref(), C.byref(o), L.ptr(x), L.ptr(ws), L.ptr(st)), "sens")
                z = ws[:B * M].clone(); ev = int(st[0])
            else:
                r = native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-8, 1e-10, 2.0, True, want_stats=True)
                z = r["z"].reshape(-1).clone(); ev = int(r["stats"][0])
            evs.append(ev); zs.append(z)
        mode_ev = max(set(evs), key=evs.count)
        print((nup, ndn, dim), mode, ": runs off the modal evaluation count", sum(e != mode_ev for e in evs), "of", NRUN, "; evaluations", min(evs), "...", max(evs))
