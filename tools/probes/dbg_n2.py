import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as Gm
from fermiflow_amd import native
dev = torch.device("cuda:0")
nup, ndn, B, wc = (int(v) for v in sys.argv[1:5])
model = Gm._model(dev, nup, ndn, 1.0)
torch.manual_seed(102)
z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
v = model.cnf.v_wrapper.v
net = v.net(radial="table")
x = native.cnf_generate(net, z, 0.0, 1.0, 1e-8, 1e-10)
tu, td = model._tables(dev)
cost = torch.zeros(B, dtype=torch.int32, device=dev) if wc else None
torch.cuda.synchronize()
r = native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-6, 1e-8, 1.0, True, want_stats=True, walker_cost=cost)
torch.cuda.synchronize()
print(os.environ.get("FERMIFLOW_LIB", "default")[-14:], nup, ndn, B, wc, r["stats"][:4].tolist(), flush=True)
