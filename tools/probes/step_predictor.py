import os, sys, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as G
from fermiflow_amd import native
dev = torch.device("cuda:0")
model = G._model(dev, 3, 3, 2.0)
net = model.cnf.v_wrapper.v.net()
tu, td = model._tables(dev)
B = 65536
z, _, _ = native.mcmc_sample(tu, td, 3, 3, B, 100, 0.1, 1, dev)
cg = torch.empty(B, dtype=torch.int32, device=dev)
x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, walker_cost=cg)
st = torch.empty(B, dtype=torch.int32, device=dev)
r = native.eloc(tu, td, 3, 3, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, walker_cost=st)
st2 = torch.empty(B, dtype=torch.int32, device=dev)
w = (r["eloc"] - r["eloc"].mean()) / B
native.cnf_adjoint(net, r["z"], w[:, None, None] * r["glogp0"], -w, 0.0, 1.0, 1e-6, 1e-8, need_gx=False, walker_cost=st2)
def rmin(y):
    d = (y[:, :, None, :] - y[:, None, :, :]).norm(dim=-1) + 1e3 * torch.eye(6, device=dev)
    return d.flatten(1).min(dim=1).values
rx, rz = x.norm(dim=-1).min(dim=1).values, r["z"].norm(dim=-1).min(dim=1).values   # one-body radii
rm = torch.minimum(rx, rz)
s = st.double()
print("eloc steps hist", torch.bincount(st.clamp(max=40)).tolist())
print("adj steps hist", torch.bincount(st2.clamp(max=40)).tolist())
print("corr eloc-adj steps", float(torch.corrcoef(torch.stack([s, st2.double()]))[0, 1]))
for name, rr in (("rmin_x", rx), ("rmin_z", rz), ("min", rm)):
    print(name, "corr(steps, 1/r)", float(torch.corrcoef(torch.stack([s, 1 / rr]))[0, 1]), "corr(steps, -log r)", float(torch.corrcoef(torch.stack([s, -rr.log()]))[0, 1]))
    edges = [0, 0.002, 0.005, 0.01, 0.02, 0.05, 0.1, 0.2, 0.3, 0.5, 1, 10]
    for a, b in zip(edges[:-1], edges[1:]):
        m = (rr >= a) & (rr < b)
        if m.any(): print("   r in [%.2f,%.2f): n=%d mean steps %.1f max %d | adj mean %.1f max %d" % (a, b, int(m.sum()), float(s[m].mean()), int(st[m].max()), float(st2[m].double().mean()), int(st2[m].max())))

print("generate cost class hist", torch.bincount(cg).tolist())
for c in range(int(cg.max()) + 1):
    m = cg == c
    if m.any(): print("   class %d: n=%d eloc steps mean %.1f max %d" % (c, int(m.sum()), float(s[m].mean()), int(st[m].max())))
