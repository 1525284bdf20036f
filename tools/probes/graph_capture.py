"""Development probe: how much would capturing one whole training iteration in a HIP graph save?  (The replayed graph
re-uses the captured Philox key, so this measures time only.)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
B = int(os.environ.get("B", 65536))
model = G._model(dev, 3, 3, 2.0)
opt = torch.optim.Adam(model.parameters(), lr=2e-5, capturable=True)


def step():
    g = model(B); opt.zero_grad(set_to_none=False); g.backward(); opt.step()


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(5): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize()
print("eager   %.3f ms/iter  E=%.6f" % ((time.perf_counter() - t0) / N * 1e3, model.E))
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step()
torch.cuda.synchronize()
for _ in range(3): graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): graph.replay()
torch.cuda.synchronize()
print("graphed %.3f ms/iter  E=%.6f" % ((time.perf_counter() - t0) / N * 1e3, model.E))
