"""configs[4] known-answer test (zero flow, Z = 0, E_loc == 60): how the E_loc error of a walker relates to the condition number of
its two 10 x 10 Slater matrices -- sets the bound of tests/test_gpu_wide.py::test_config5_known_answer_at_full_size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import fermiflow_amd as ff
from tests.test_gpu_wide import _model3d
dev = torch.device("cuda:0")
model = _model3d(dev, 10, 10, 0.0, True)
torch.manual_seed(4)
g = model(131072); g.backward()
x = model.x
h = ff.HO3D()
def cond(xs):
    D = torch.stack([o(xs) for o in h.orbitals[:10]], dim=-1)      # (B, 10 particles, 10 orbitals)
    return torch.linalg.cond(D)
c = torch.maximum(cond(x[:, :10]), cond(x[:, 10:]))
err = (model.Eloc - 60.0).abs()
print("E", model.E, "max err", err.max().item(), "n>1e-6", (err > 1e-6).sum().item(), "n>1e-3", (err > 1e-3).sum().item())
for lo, hi in ((0, 1e3), (1e3, 1e4), (1e4, 1e5), (1e5, 1e6), (1e6, 1e7), (1e7, 1e9), (1e9, 1e30)):
    m = (c >= lo) & (c < hi)
    if m.any():
        print(f"cond [{lo:.0e},{hi:.0e}): n={m.sum().item():7d} max err {err[m].max().item():.3e}  max err/cond^2 {(err[m]/c[m]**2).max().item():.3e} max err/cond {(err[m]/c[m]).max().item():.3e}")
