"""Is the host ahead of the GPU, and what do the stage markers cost?  (config 2; run on the GPU box)
  python tools/probes/host_ahead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as G
import fermiflow_amd as ff
from fermiflow_amd.utils import make_adam

dev = torch.device("cuda:0")
model = G._model(dev, 3, 3, 2.0)
opt = make_adam(model.parameters(), lr=2e-5)
torch.manual_seed(1234)
B = 65536

def step():
    g = model(B); opt.zero_grad(); g.backward(); opt.step()

for _ in range(15):
    step()
torch.cuda.synchronize()
import collections
acc = collections.defaultdict(list)
for rep in range(12):
    for tag in ("no markers", "markers"):
        model.profile = None if tag == "no markers" else {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        acc[tag].append(((time.perf_counter() - t0) / 10 * 1e3, th / 10 * 1e3))
for tag, v in acc.items():
    print(f"{tag:12s}: GPU ms/iter per block " + " ".join(f"{a:.3f}" for a, _ in v) + f" | host {sum(b for _, b in v) / len(v):.3f}")
model.profile = None
# host time of one iteration's enqueue when the GPU is kept busy by a long kernel in front (pure host cost)
torch.cuda.synchronize()
x = torch.randn(8192, 8192, device=dev)
for _ in range(30):
    x = x @ x * 1e-4
t0 = time.perf_counter()
for _ in range(10):
    step()
print(f"host cost of enqueuing one iteration: {(time.perf_counter() - t0) / 10 * 1e3:.4f} ms")
torch.cuda.synchronize()
