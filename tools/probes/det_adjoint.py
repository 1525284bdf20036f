"""Run-to-run reproducibility of the multi-wave ADJOINT kernels (ff_adj_wide.h) and of the flow kernels, shapes alternating."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as Gm
import fermiflow_amd as ff
from fermiflow_amd import native
dev = torch.device("cuda:0")
NRUN = int(os.environ.get("NRUN", "60"))
def setup(nup, ndn, dim):
    if dim == 2:
        model = Gm._model(dev, nup, ndn, 2.0)
    else:
        gs = Gm._model(dev, 2, 2, 2.0)
        model = ff.GSVMC(nup, ndn, ff.HO3D(), ff.FreeFermion(device=dev), gs.cnf, ff.CoulombPairPotential(2.0), sp_potential=ff.HO())
    torch.manual_seed(31 + nup)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (200,))
    net = model.cnf.v_wrapper.v.net(refresh=True)
    x = native.cnf_generate(net, z, 0.0, 1.0, 1e-8, 1e-10)
    az = torch.randn_like(x); ad = torch.randn(200, dtype=torch.float64, device=dev)
    def run():
        xs = native.cnf_generate(net, z, 0.0, 1.0, 1e-8, 1e-10)
        zb, dl = native.cnf_delta_logp(net, x, 0.0, 1.0, 1e-8, 1e-10)
        gx, gp = native.cnf_adjoint(net, zb, az, ad, 0.0, 1.0, 1e-8, 1e-10)
        return dict(x=xs, zb=zb, dl=dl, gx=gx, gp=gp)
    return run
shapes = [(3, 3, 2), (7, 6, 2), (12, 12, 2), (5, 5, 3), (10, 10, 3)]
runs = {s: setup(*s) for s in shapes}
ref = {s: runs[s]() for s in shapes}
cnt = {s: {k: 0 for k in ref[s]} for s in shapes}
for it in range(NRUN):
    for s in shapes:
        r = runs[s]()
        for k in r:
            cnt[s][k] += int(not torch.equal(r[k], ref[s][k]))
for s in shapes:
    print(s, "runs differing from the first, of", NRUN, ":", cnt[s])
