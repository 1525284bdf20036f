"""The local-energy kernels behind a polluter that leaves NaN patterns in all of LDS and in the vector registers of every SIMD:
a read before the first write shows as NaN / fault, not as an extra rejected step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
import __graft_entry__ as Gm
import fermiflow_amd as ff
from fermiflow_amd import native, _lib as L
P = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "polluter.so"))
dev = torch.device("cuda:0")
NRUN = int(os.environ.get("NRUN", "60"))
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
for (nup, ndn, dim, bits) in ((3, 3, 2, 64), (7, 6, 2, 64), (7, 6, 2, 32), (12, 12, 2, 64), (10, 10, 3, 64), (10, 10, 3, 32)):
    net, x, tu, td = setup(nup, ndn, dim)
    for polluted in (False, True):
        evs, nans = [], 0
        pb = native.set_sens_precision(bits)
        for it in range(NRUN):
            if polluted:
                assert P.ff_pollute(L.stream()) == 0
            r = native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-8, 1e-10, 2.0, True, want_stats=True)
            evs.append(int(r["stats"][0]))
            nans += int(torch.isnan(r["eloc"]).sum()) + int(torch.isnan(r["z"]).sum())
        native.set_sens_precision(pb)
        m = max(set(evs), key=evs.count)
        print((nup, ndn, dim, bits), "polluted" if polluted else "clean   ", ": runs off the modal evaluation count", sum(e != m for e in evs), "of", NRUN, "; NaNs", nans, "; evaluations", min(evs), "...", max(evs), flush=True)
