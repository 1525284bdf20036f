"""Helpers shared by the oracle, hostsim and GPU parity tests."""
import hashlib

import numpy as np
import torch


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def mcmc_noise_from_seed(G, name):
    """Regenerate the torch-CPU noise stream of a g1 case in the reference's draw order
    (src/base_dist.py:62,65,68) and check it against the committed SHA-256 (detects RNG drift)."""
    nup, ndn, B, seed, steps = (int(v) for v in G[name + "_cfg"])
    torch.manual_seed(seed)
    n = nup + ndn
    g0 = torch.randn(B, n, 2, dtype=torch.float64)
    gs, us = [], []
    for _ in range(steps):
        gs.append(torch.randn(B, n, 2, dtype=torch.float64))
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
