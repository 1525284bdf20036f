"""Data parallelism over walkers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Walkers are independent in the MCMC, the flow and the local energy; ranks couple only through two all-reduces per sweep
  GSVMC:   [sum (e - c), sum (e - c)^2, sum logp, sum logp (e - c)]   (ff_reduce_energy -> here -> ff_energy_finish: E, E_std, surrogate;
           c = the previous sweep's E, src/VMC.py:56-59), then the 3(He+Hm)-double parameter gradient (src/FermionHO2D.py:71)
  BetaVMC: [moments of E_loc | partial per-state sums]                 (ff_beta_state_partials -> here -> ff_beta_finish,
           src/VMC.py:146-171), then the parameter gradient
all a few hundred bytes to 2.4 KB, i.e. latency-bound: one fused buffer per phase, no bucketing.
The functions work on any device so the same code runs under gloo on CPU in the tests.
"""
import os

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _active(force=False):
    # FERMIFLOW_DIST_FORCE=1: run the collectives even in a single-rank group (exercises the RCCL path on a 1-GPU box)
    force = force or os.environ.get("FERMIFLOW_DIST_FORCE") == "1"
    return dist.is_available() and dist.is_initialized() and (force or dist.get_world_size() > 1)


def all_reduce_sum_(t, force=False):
    """In-place sum over ranks.  A single-rank group skips the collective unless force=True (the tests run the RCCL
    branch once that way on a 1-GPU box)."""
    if _active(force):
        if t.is_cuda and dist.get_backend() == "gloo":      # test set-up: several ranks sharing one GPU over gloo
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_(t, src=0, force=False):
    """In-place broadcast from rank `src`."""
    if _active(force):
        if t.is_cuda and dist.get_backend() == "gloo":
            c = t.cpu()
            dist.broadcast(c, src=src)
            t.copy_(c)
        else:
            dist.broadcast(t, src=src)
    return t


def sync_parameters(module):
    """Every rank takes rank 0's parameters and buffers (one flat broadcast).  The estimators call this once, on their
    first sweep: ranks that built their model from differently seeded generators (e.g. BetaVMC's random state logits,
    src/VMC.py:83) would otherwise all-reduce gradients of different functions without any error."""
    if not _active():
        return
    ts = [p.data for p in module.parameters()] + [b for b in module.buffers()]
    ts = [t for t in ts if t.is_floating_point()]
    if not ts:
        return
    flat = torch.cat([t.reshape(-1).to(torch.float64) for t in ts])
    broadcast_(flat)
    off = 0
    for t in ts:
        t.copy_(flat[off:off + t.numel()].reshape(t.shape))
        off += t.numel()


def shard(batch, rank=None, world_size=None):
    """Contiguous block of the global walker batch owned by `rank`: (offset, count)."""
    if rank is None:
        rank, world_size = world()
    base, rem = divmod(int(batch), world_size)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count
