"""Walker data parallelism on the GPU: two ranks (sharing the one GPU of the test box, gloo for the tiny all-reduces;
production uses nccl = RCCL, one GPU per rank) must reproduce the single-process iteration on the same global batch:
identical walkers (Philox counters are global walker indices), same E / E_std, same parameter gradient."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(rank, world, port, B, out):
    import torch.distributed as dist
    import __graft_entry__ as G
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    model = G._model(dev, 3, 3, 2.0)
    torch.manual_seed(123)
    g = model(B)
    g.backward()
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu().numpy()
    out[rank] = (model.E, model.E_std, g.item(), grads, model.x[:4].cpu().numpy(), model.x.shape[0])
    if world > 1:
        dist.destroy_process_group()


def test_two_ranks_equal_one_rank():
    B = 4096
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    one, two = mgr.dict(), mgr.dict()
    mp.spawn(_run, args=(1, 0, B, one), nprocs=1, join=True)
    mp.spawn(_run, args=(2, _free_port(), B, two), nprocs=2, join=True)
    E, Es, g, grads, x0, n = one[0]
    assert n == B
    for r in (0, 1):
        E2, Es2, g2, grads2, x02, n2 = two[r]
        assert n2 == B // 2
        assert abs(E2 - E) < 1e-12 * abs(E) and abs(Es2 - Es) < 1e-11 * Es
        assert abs(g2 - g) < 1e-10 * max(1.0, abs(g))
        np.testing.assert_allclose(grads2, grads, rtol=0, atol=1e-10 * np.abs(grads).max())
    assert (two[0][4] == x0).all()          # rank 0's first walkers are the global first walkers, bit for bit


def _run_beta(rank, world, port, B, out):
    import torch.distributed as dist
    import fermiflow_amd as ff
    import __graft_entry__ as G
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    gs = G._model(dev, 3, 0, 2.0)
    model = ff.BetaVMC(2.0, 3, 0, 2.0, True, ff.HO2D(), ff.FreeFermion(device=dev), gs.cnf,
                       ff.CoulombPairPotential(2.0), sp_potential=ff.HO())
    model.to(dev)
    torch.manual_seed(77)
    gphi, gtheta = model(B)
    (gphi + gtheta).backward()
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu().numpy()
    out[rank] = (model.E, model.E_std, model.F, model.F_std, model.S, gphi.item(), gtheta.item(), grads)
    if world > 1:
        dist.destroy_process_group()


def test_betavmc_two_ranks_equal_one_rank():
    B = 2048
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    one, two = mgr.dict(), mgr.dict()
    mp.spawn(_run_beta, args=(1, 0, B, one), nprocs=1, join=True)
    mp.spawn(_run_beta, args=(2, _free_port(), B, two), nprocs=2, join=True)
    ref = one[0]
    for r in (0, 1):
        got = two[r]
        for a, b in zip(got[:7], ref[:7]):
            assert abs(a - b) <= 1e-10 * max(1.0, abs(b)), (a, b)
        np.testing.assert_allclose(got[7], ref[7], rtol=0, atol=1e-10 * np.abs(ref[7]).max())


def _run_nccl_single(rank, port, B, out):
    """One rank, backend nccl (= RCCL): every collective of the sweep really runs (FERMIFLOW_DIST_FORCE)."""
    import torch.distributed as dist
    import __graft_entry__ as G
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["FERMIFLOW_DIST_FORCE"] = "1"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from fermiflow_amd import dist as D
    t = torch.arange(4, dtype=torch.float64, device=dev)
    D.all_reduce_sum_(t); D.broadcast_(t)
    assert t.tolist() == [0.0, 1.0, 2.0, 3.0]
    model = G._model(dev, 3, 3, 2.0)
    torch.manual_seed(123)
    g = model(B)
    g.backward()
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu().numpy()
    out[0] = (model.E, model.E_std, g.item(), grads, dist.get_backend())
    dist.destroy_process_group()


def test_rccl_single_rank_sweep_equals_plain_sweep():
    """the nccl (RCCL) branch of dist.all_reduce_sum_ / broadcast_ executes, inside a full sweep,
    and changes nothing."""
    B = 4096
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    one, rc = mgr.dict(), mgr.dict()
    mp.spawn(_run, args=(1, 0, B, one), nprocs=1, join=True)
    mp.spawn(_run_nccl_single, args=(_free_port(), B, rc), nprocs=1, join=True)
    E, Es, g, grads, _, _ = one[0]
    E2, Es2, g2, grads2, backend = rc[0]
    assert backend == "nccl"
    assert E2 == E and Es2 == Es and g2 == g and (grads2 == grads).all()


def _run_rccl(rank, world, port, B, out):
    """One rank per GPU over nccl (= RCCL): the production collective path of fermiflow_amd/dist.py."""
    import torch.distributed as dist
    import __graft_entry__ as G
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["FERMIFLOW_DIST_FORCE"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device(f"cuda:{rank}")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    model = G._model(dev, 3, 3, 2.0)
    torch.manual_seed(123)
    g = model(B)
    g.backward()
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu().numpy()
    out[rank] = (model.E, model.E_std, g.item(), grads, model.x[:4].cpu().numpy(), model.x.shape[0], dist.get_backend())
    dist.destroy_process_group()


def test_two_ranks_over_rccl_equal_one_rank():
    """test_two_ranks_equal_one_rank over RCCL, one GPU per rank -- the collective the 8-GPU bench uses, not
    the gloo / CPU detour the one-GPU test box is limited to.  Skipped unless two devices are visible (RCCL refuses two ranks on one
    GPU), so that the first multi-GPU box exercises it."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL: one device per rank)")
    B = 4096
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    one, two = mgr.dict(), mgr.dict()
    mp.spawn(_run, args=(1, 0, B, one), nprocs=1, join=True)
    mp.spawn(_run_rccl, args=(2, _free_port(), B, two), nprocs=2, join=True)
    E, Es, g, grads, x0, n = one[0]
    for r in (0, 1):
        E2, Es2, g2, grads2, x02, n2, backend = two[r]
        assert backend == "nccl" and n2 == B // 2
        assert abs(E2 - E) < 1e-12 * abs(E) and abs(Es2 - Es) < 1e-11 * Es
        assert abs(g2 - g) < 1e-10 * max(1.0, abs(g))
        np.testing.assert_allclose(grads2, grads, rtol=0, atol=1e-10 * np.abs(grads).max())
    assert (two[0][4] == x0).all()


def _bench(args, env_extra=None, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = json.loads(ln)
    return p.returncode, line, p.stderr


def test_bench_gpus_2_refuses_a_one_gpu_box_and_runs_two_ranks_over_gloo():
    """bench.py --gpus 2: with the production backend (nccl = RCCL, one GPU per rank) it must exit 2 on a
    box with one device instead of reporting n_gpus it did not use; with FF_BENCH_BACKEND=gloo the self-launch path runs two
    ranks on the one GPU -- weak scaling doubles the global batch, strong scaling (--scaling strong) splits the 1-rank batch,
    and then E must equal the 1-rank E to reduction-order noise (same Philox streams by global walker index)."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    common = ["--steps", "2", "--warmup", "1", "--no-extras", "--walkers-per-gpu", "4096"]
    rc, line, err = _bench(["--gpus", "2"] + common)
    assert rc == 2 and line is None and "device(s) visible" in err
    rc1, one, _ = _bench(["--gpus", "1"] + common)
    assert rc1 == 0 and one["n_gpus"] == 1 and one["scaling"] == "weak"
    rc2, two, err2 = _bench(["--gpus", "2", "--scaling", "strong"] + common, {"FF_BENCH_BACKEND": "gloo"})
    assert rc2 == 0, err2[-2000:]
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["global_walkers"] == 4096
    assert abs(two["E"] - one["E"]) < 1e-12 * abs(one["E"]) and abs(two["E_std"] - one["E_std"]) < 1e-10 * one["E_std"]
    rc3, weak, err3 = _bench(["--gpus", "2"] + common, {"FF_BENCH_BACKEND": "gloo"})
    assert rc3 == 0, err3[-2000:]
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["config"]["global_walkers"] == 8192


def test_eight_ranks_strong_scaling_dry_run():
    """the first 8-GPU run should be boring.  bench.py --gpus 8 --scaling strong at the headline's global
    batch (65 536 walkers, 8 192 per rank) with eight gloo ranks sharing the one GPU of the test box: the self-launch path, the
    shard arithmetic, both estimator all-reduces and the rank-0 parameter sync at the real width -- E and E_std must equal the
    one-rank values to reduction-order noise (the walkers are the same: Philox counters are global walker indices)."""
    if torch.cuda.device_count() >= 8:
        pytest.skip("written for a one-GPU box (the driver measures the real curve)")
    common = ["--steps", "2", "--warmup", "1", "--no-extras", "--walkers-per-gpu", "65536"]
    rc1, one, err1 = _bench(["--gpus", "1"] + common)
    assert rc1 == 0, err1[-2000:]
    rc8, eight, err8 = _bench(["--gpus", "8", "--scaling", "strong"] + common, {"FF_BENCH_BACKEND": "gloo"}, timeout=900)
    assert rc8 == 0, err8[-2000:]
    assert eight["n_gpus"] == 8 and eight["scaling"] == "strong" and eight["config"]["global_walkers"] == 65536
    assert abs(eight["E"] - one["E"]) < 1e-12 * abs(one["E"]), (eight["E"], one["E"])
    assert abs(eight["E_std"] - one["E_std"]) < 1e-10 * one["E_std"]


def test_ff_comm_single_rank_all_reduce():
    """ff_comm_* (SURVEY 8(b)'s minimum symbol set; include/fermiflow.h): the RCCL wrappers a torch-free caller sums the estimator's
    buffers with.  One rank on the one GPU of the test box: id, init, an in-place all-reduce of the 4-double and the 300-double
    buffer of a sweep on torch's current stream (a one-rank sum is the identity), destroy.  More ranks need more devices (RCCL refuses
    two ranks on one GPU); the driver's 8-GPU bench goes through torch.distributed, whose "nccl" backend is the same RCCL."""
    import ctypes as C
    from fermiflow_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    uid = (C.c_char * 128)()
    L.check(lib.ff_comm_unique_id(uid), "ff_comm_unique_id")
    comm = C.c_void_p()
    L.check(lib.ff_comm_init(C.byref(comm), 1, 0, uid), "ff_comm_init")
    try:
        for n in (4, 300):
            buf = torch.arange(1, n + 1, dtype=torch.float64, device=dev) / 7
            want = buf.clone()
            L.check(lib.ff_comm_allreduce(comm, L.stream(), L.ptr(buf), L.i64(n)), "ff_comm_allreduce")
            torch.cuda.synchronize()
            assert torch.equal(buf, want)
        assert lib.ff_comm_allreduce(comm, L.stream(), None, L.i64(3)) != 0            # null buffer: FF_EINVAL, no crash
    finally:
        L.check(lib.ff_comm_destroy(comm), "ff_comm_destroy")
    assert lib.ff_comm_destroy(None) == 0
    assert lib.ff_comm_init(C.byref(comm), 2, 5, uid) != 0                            # rank outside the world
