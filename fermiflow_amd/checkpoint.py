"""state_dict checkpoints for the two drivers (absent upstream; SURVEY 8(f).3, pattern of tests/demos/checkpoint.py:44-72
in the reference).  A checkpoint carries everything the NEXT iteration depends on: parameters, optimizer moments, the
sweep state of the estimator (previous mean energies used as reduction shifts, step-size warm start, persistent walkers:
`_extra_state` of the model's state_dict) and the torch RNG streams the sweep draws from (CPU generator -> Philox key of
the Metropolis kernel; device generator -> BetaVMC's state sampling).

Multi-rank runs write ONE checkpoint (rank 0's).  Parameters, optimizer and the RNG streams are identical on every rank; of the
sweep state, the prefetched walkers are rank-stamped (other ranks re-draw their own shard from the same Philox key: bit-identical
to the uninterrupted run) and `h_flow`, the mean flow step size that opens the next sweep's first integration, is rank 0's local
mean: on ranks above 0 the resumed run therefore repeats the uninterrupted one to solver tolerance (the error test of every step is
unchanged), not bit for bit."""
import os

import torch


def save(path, model, optimizer, it, device=None):
    ck = {"model": model.state_dict(), "optimizer": optimizer.state_dict(), "iter": int(it),
          "rng_cpu": torch.get_rng_state()}
    if device is not None and torch.device(device).type == "cuda":
        ck["rng_cuda"] = torch.cuda.get_rng_state(device)
    tmp = path + ".tmp"
    torch.save(ck, tmp)
    os.replace(tmp, path)          # a crash while writing never leaves a truncated checkpoint behind


def load(path, model, optimizer, device=None):
    """Restores model / optimizer / RNG state in place; returns the iteration the checkpoint was written after."""
    ck = torch.load(path, map_location=device, weights_only=True)      # tensors, dicts, ints only: nothing to unpickle
    model.load_state_dict(ck["model"])
    optimizer.load_state_dict(ck["optimizer"])
    if "rng_cpu" in ck:
        torch.set_rng_state(ck["rng_cpu"].cpu())
    if "rng_cuda" in ck and device is not None and torch.device(device).type == "cuda":
        torch.cuda.set_rng_state(ck["rng_cuda"].cpu(), device)
    return int(ck["iter"])
