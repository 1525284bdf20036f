"""Probe: RHS evaluations of the flow pass (ff_cnf_generate) by the rule its walkers open with -- the mean step the previous pass
accepted x scale, optionally rounded down to equal steps (ff_ode.walker_h_equal).  Weight sets: synthetic + tests/golden/trained_weights.npz.
usage: python tools/probes/flow_open.py [nup ndown]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
from fermiflow_amd import native
dev = torch.device("cuda:0")
nup, ndn = (int(a) for a in (sys.argv[1:3] + ["3", "3"][len(sys.argv) - 1:]))
B = 65536
W = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "trained_weights.npz"))
for tag in ("head", "trained", "driver", "driver1000"):
    model = G._model(dev, nup, ndn, 2.0)
    if tag != "head":
        if nup + ndn != 6:
            continue
        v = model.cnf.v_wrapper.v
        with torch.no_grad():
            for nm, m in (("eta", v.eta), ("mu", v.mu)):
                m.fc1.weight.copy_(torch.as_tensor(W[f"{tag}_{nm}_w1"]).reshape(-1, 1))
                m.fc1.bias.copy_(torch.as_tensor(W[f"{tag}_{nm}_b1"]))
                m.fc2.weight.copy_(torch.as_tensor(W[f"{tag}_{nm}_w2"]).reshape(1, -1))
    net = model.cnf.v_wrapper.v.net(refresh=True)
    torch.manual_seed(3)
    z = model.basedist.sample(model.orbitals_up, model.orbitals_down, (B,))
    hg = torch.zeros(B, dtype=torch.float64, device=dev)
    x0, st = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, want_stats=True, walker_h_out=hg)
    hm = hg.mean().reshape(1)
    print(f"== {tag}: cold {st[0].item() / B:.2f} evaluations per walker, mean largest accepted step {hm.item():.4f} (min {hg.min().item():.3f})")
    for scale, eq in ((0.75, False), (0.75, True), (0.85, True), (0.95, True), (1.0, True), (1.1, True)):
        ts = []
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            x, st = native.cnf_generate(net, z, 0.0, 1.0, 1e-6, 1e-8, want_stats=True, walker_h_init=hm, walker_h_scale=scale, walker_h_uniform=True,
                                        walker_h_equal=eq)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        err = (x - x0).abs().max().item()
        print(f"   scale {scale:4.2f} equal {int(eq)}: {st[0].item() / B:6.2f} evaluations, rejected steps per walker {st[2].item() / B:.3f}, max steps {int(st[1])}, "
              f"{min(ts) * 1e3:.0f} us, |x - x_cold| {err:.1e}")
