"""Helpers shared by the oracle, hostsim and GPU parity tests."""
import hashlib

import numpy as np
import torch


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def mcmc_noise_from_seed(G, name, dim=2):
    """Regenerate the torch-CPU noise stream of a g1 (dim = 2) or g7 (dim = 3) case in the reference's draw order
    (src/base_dist.py:62,65,68) and check it against the committed SHA-256 (detects RNG drift)."""
    nup, ndn, B, seed, steps = (int(v) for v in G[name + "_cfg"])
    torch.manual_seed(seed)
    n = nup + ndn
    g0 = torch.randn(B, n, dim, dtype=torch.float64)
    gs, us = [], []
    for _ in range(steps):
        gs.append(torch.randn(B, n, dim, dtype=torch.float64))
        us.append(torch.rand(B, dtype=torch.float64))
    g0, g, u = g0.numpy(), torch.stack(gs).numpy(), torch.stack(us).numpy()
    if sha(g0) + sha(g) + sha(u) != str(G[name + "_noise_sha"]):
        raise RuntimeError("torch CPU RNG stream differs from the one the golden vectors were made with")
    accept = np.unpackbits(G[name + "_accept"])[:steps * B].reshape(steps, B)
    return nup, ndn, g0, g, u, accept


def net_arrays(G, prefix, use_mu=True):
    eta = (G[prefix + "eta_w1"], G[prefix + "eta_b1"], G[prefix + "eta_w2"])
    mu = (G[prefix + "mu_w1"], G[prefix + "mu_b1"], G[prefix + "mu_w2"]) if use_mu else None
    return eta, mu


def cnf_param_grads(G, tag):
    names = [str(s) for s in G[tag + "_pnames"]]
    return np.concatenate([G[f"{tag}_g_{nm}"] for nm in names])


GSVMC_PG = ["cnf.v_wrapper.v.eta.fc1.weight", "cnf.v_wrapper.v.eta.fc1.bias", "cnf.v_wrapper.v.eta.fc2.weight",
            "cnf.v_wrapper.v.mu.fc1.weight", "cnf.v_wrapper.v.mu.fc1.bias", "cnf.v_wrapper.v.mu.fc2.weight"]


def gsvmc_param_grads(G, name, use_mu=True):
    keys = GSVMC_PG if use_mu else GSVMC_PG[:3]
    return np.concatenate([G[f"{name}_pg_{k}"] for k in keys])


# ---- GPU-side helpers (fermiflow_amd is imported lazily: the CPU suite imports this module too)
def T(a, dev, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def make_mlp(w, dev):
    import fermiflow_amd as ff
    m = ff.MLP(1, len(w[1]))
    with torch.no_grad():
        m.fc1.weight.copy_(torch.as_tensor(w[0]).reshape(-1, 1))
        m.fc1.bias.copy_(torch.as_tensor(w[1]))
        m.fc2.weight.copy_(torch.as_tensor(w[2]).reshape(1, -1))
    return m.to(dev)


def make_flow(eta, mu, dev):
    import fermiflow_amd as ff
    v = ff.Backflow(make_mlp(eta, dev), mu=make_mlp(mu, dev) if mu is not None else None)
    return ff.CNF(v, (0.0, 1.0))
