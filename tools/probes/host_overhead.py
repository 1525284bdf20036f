"""Development probe: host time to enqueue one training iteration vs GPU time per iteration."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
dev = torch.device("cuda:0")
model = G._model(dev, 3, 3, 2.0)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
B = int(os.environ.get("B", 65536))
def step():
    g = model(B); opt.zero_grad(); g.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/iter, total %.3f ms/iter" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
