"""A/B of the local-energy kernels at config 2 (65536 walkers): time of ff_eloc_sensitivities, evaluations, agreement."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as G
from fermiflow_amd import native
dev = torch.device("cuda:0")
nup = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ndn = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
model = G._model(dev, nup, ndn, 2.0)
torch.manual_seed(1234)
z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
net = model.cnf.v_wrapper.v.net()
cost = torch.empty(B, dtype=torch.int32, device=dev); hg = torch.empty(B, dtype=torch.float64, device=dev)
x = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, walker_cost=cost, walker_h_out=hg)
if os.environ.get("FF_AB_LIGHT"):      # throughput only: the walkers of the lowest cost classes, repeated to fill the batch
    keep = (cost <= int(os.environ["FF_AB_LIGHT"])).nonzero().squeeze(1)
    idx = keep[torch.arange(B, device=dev) % keep.numel()]
    x, z, cost, hg = x[idx].contiguous(), z[idx].contiguous(), cost[idx].contiguous(), hg[idx].contiguous()
    print("light walkers:", keep.numel(), "of", B)
if os.environ.get("FF_AB_HEAVY"):      # latency only: the walkers of the highest cost classes alone
    idx = torch.argsort(cost, descending=True)[: int(os.environ["FF_AB_HEAVY"])]
    x, z, cost, hg = x[idx].contiguous(), z[idx].contiguous(), cost[idx].contiguous(), hg[idx].contiguous()
    B = x.shape[0]
order = native.walker_order(cost)
tu, td = model._tables(dev)
scale = model._h_scale_eloc
wc = torch.zeros(x.shape[0], dtype=torch.int32, device=dev)
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r = native.eloc(tu, td, nup, ndn, net, x, 0.0, 1.0, 1e-6, 1e-8, 2.0, True, want_stats=True, pass1_events=(e0, e1),
                    walker_order=order, walker_h_init=hg, walker_h_scale=scale, walker_cost=wc)
    torch.cuda.synchronize()
print(os.environ.get("FF_ELOC_KERNEL", "rows"), nup, ndn, B, "pass1 ms %.3f" % e0.elapsed_time(e1), "evals/walker %.2f" % (r["stats"][0].item() / B),
      "max acc", r["stats"][1].item(), "steps hist", torch.bincount(wc.clamp(max=40)).tolist(), "rej", r["stats"][2].item(), "fail", r["stats"][3].item(), "E %.6f" % r["eloc"].mean().item(), flush=True)
if os.environ.get("FERMIFLOW_LIB", "").endswith("stamps.so"):
    import numpy as np
    st = r["stats"].cpu().numpy()[8:8 + 18].view(np.uint64)
    tot = st.sum()
    kind = os.environ.get("FF_ELOC_KERNEL", "mfma")
    names = ["form", "publish", "R1", "sweep/mfma", "R2", "sums", "consume", "epilogue", "-"]
    if kind == "mfma":
        names = ["form", "publish", "R1", "rows", "products", "R2", "sums+consume", "epilogue", "-"]
    nwe = r["stats"][0].item() / (4.0 if kind == "mfma" else 5.0)
    print("phase cycles per wave-eval:", {n: int(v / nwe) for n, v in zip(names, st)}, "total/eval", int(tot / nwe))
torch.save(r["eloc"].cpu(), f"/tmp/eloc_{os.environ.get('FF_ELOC_KERNEL', 'rows')}.pt")
if os.path.exists("/tmp/eloc_rows.pt") and os.path.exists("/tmp/eloc_columns.pt"):
    a, b = torch.load("/tmp/eloc_rows.pt"), torch.load("/tmp/eloc_columns.pt")
    if a.shape == b.shape:
        print("max rel diff rows vs columns: %.2e" % ((a - b).abs() / b.abs()).max().item())
