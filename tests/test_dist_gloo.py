"""world_size-2 CPU test (gloo) of the data-parallel estimator reductions in fermiflow_amd/dist.py:
two ranks holding halves of a walker batch must produce the single-process E, E_std and gradient."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, eloc_all, logp_all, g_all, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fermiflow_amd import dist as D
    B = eloc_all.numel()
    off, cnt = D.shard(B)
    e = eloc_all[off:off + cnt]
    E, E_std, n = D.global_mean_std(e.sum(), cnt, lambda m: ((e - m) ** 2).sum())
    # the sweep's variant: tensors in, tensors out, global count known on the host (no .item() anywhere)
    Et, St = D.global_mean_std_dev(e.sum(), B, lambda tot, scale: ((e - tot[0] * scale) ** 2).sum())
    assert isinstance(Et, torch.Tensor) and abs(Et.item() - E) < 1e-13 * abs(E) and abs(St.item() - E_std) < 1e-12 * E_std
    w = (e - E) / n
    buf = torch.cat([(logp_all[off:off + cnt] * w).sum().reshape(1), (w[:, None] * g_all[off:off + cnt]).sum(0)])
    D.all_reduce_sum_(buf)
    out[rank] = (E, E_std, n, buf.clone())
    dist.destroy_process_group()


def test_two_rank_estimator_matches_single_process():
    torch.manual_seed(0)
    B, P = 1001, 300
    eloc = 30 + 4 * torch.randn(B, dtype=torch.float64)
    logp = torch.randn(B, dtype=torch.float64)
    g = torch.randn(B, P, dtype=torch.float64)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), eloc, logp, g, out), nprocs=2, join=True)
    E, E_std = eloc.mean().item(), eloc.std().item()
    w = (eloc - E) / B
    ref = torch.cat([(logp * w).sum().reshape(1), (w[:, None] * g).sum(0)])
    for r in (0, 1):
        e, s, n, buf = out[r]
        assert n == B
        assert abs(e - E) < 1e-12 * abs(E) and abs(s - E_std) < 1e-12 * E_std
        np.testing.assert_allclose(buf.numpy(), ref.numpy(), rtol=1e-10, atol=1e-13)


def _sync_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fermiflow_amd import dist as D
    torch.manual_seed(100 + rank)                      # ranks build DIFFERENT parameters (e.g. BetaVMC's random state logits)
    m = torch.nn.Sequential(torch.nn.Linear(1, 5), torch.nn.Linear(5, 1)).double()
    m.register_buffer("shift", torch.randn(3, dtype=torch.float64))
    before = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
    D.sync_parameters(m)
    after = torch.cat([p.detach().reshape(-1) for p in m.parameters()] + [m.shift])
    idx = torch.arange(10) * (rank + 1)
    D.broadcast_(idx)
    out[rank] = (before, after, idx)
    dist.destroy_process_group()


def test_ranks_adopt_rank0_parameters_and_state_list():
    """ranks with differently seeded models must not all-reduce gradients of different functions --
    sync_parameters (called by the estimators' first sweep) and broadcast_ (BetaVMC's state list) make rank 0 authoritative."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sync_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert not torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][1][:out[0][0].numel()], out[0][0])
    assert torch.equal(out[0][2], out[1][2]) and out[1][2].tolist() == list(range(10))
