#!/usr/bin/env python3
"""bench.py -- headline benchmark: walker-steps/s of one full VMC training iteration.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one complete iteration of the reference training loop (src/FermionHO2D.py:66-72):
    gradE = model(batch); optimizer.zero_grad(); gradE.backward(); optimizer.step()
i.e. 100 Metropolis walker-steps per walker + CNF.generate + local energy + E/E_std + parameter gradient
(adjoint) + Adam.  Workload = BASELINE.json configs[1]: nup = ndown = 3, 2-D, Z = 2.0, H = 50, fp64,
65 536 walkers PER GPU (weak scaling: walkers shard with no data-path collective, only the three tiny
estimator all-reduces of fermiflow_amd/dist.py).  value = n_gpus * 65536 * 100 * K / time.

Extra objects on the JSON line: `roofline` for the dominant kernel (the fused local-energy integration,
fp64-VALU bound; its algorithmic flops are stated in DESIGN.md), `roofline_hbm` for the one HBM-bound
kernel of the path (parity-mode Metropolis sweep), `stages` (ms per stage), `cpu_baseline` (the C oracle
timed on the host cores on a bounded sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per RHS evaluation of one walker in the local-energy kernel ("Kernels"):
FLOP_PER_SIGMOID_UNIT = 30      # 1 exp + 1 rcp + ~8 FMA-class ops      (FERMIFLOW_RADIAL=exact)
FLOP_PER_TABLE_RADIUS = 60      # 4 heads x 5 Horner FMAs + index/offset arithmetic     (FERMIFLOW_RADIAL=table, default)
FLOP_PER_RADIUS_RECORD = 55     # r, 1/r, the direction-independent record and the own-row contributions of one radius
FLOP_PER_PAIR_TERM = 44         # counted from the jet sweep: one (direction, pair) term (FMA = 2); was 55 before the
FLOP_PER_ONEBODY_TERM = 40      # radius records took the direction-independent part out of the sweep (profiles r01_a..g)
FLOP_PER_LANE_GATHER = 30       # own rows (18 adds) + transposition sum (12 adds)
PEAK_FP64_TFLOPS = 78.6         # MI355X fp64 vector = fp64 matrix peak (vendor)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def self_launch(n):
    """Start n ranks of this script under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) as a
    child process and pass its output through.  The parent never initialises a GPU: device_count() only counts."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if os.environ.get("FF_BENCH_BACKEND", "nccl") == "nccl" and have < n:
        print(f"bench.py: --gpus {n} requested but {have} device(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--walkers-per-gpu", type=int, default=65536)
    ap.add_argument("--nup", type=int, default=3)
    ap.add_argument("--ndown", type=int, default=3)
    ap.add_argument("--Z", type=float, default=2.0)
    ap.add_argument("--lr", type=float, default=2e-5,
                    help="Adam step; small so the synthetic weights (hence the ODE step counts) stay put over the run")
    ap.add_argument("--cpu-walkers", type=int, default=4096, help="sample size of the CPU baseline (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the HBM-kernel and CPU-baseline legs")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Launched bare (`python bench.py --gpus N`): this parent starts the N ranks itself -- before anything here has
        # touched a GPU -- and relays rank 0's JSON line.
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    import __graft_entry__ as G
    from fermiflow_amd import native

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FF_BENCH_BACKEND=gloo (debug): lets several ranks share one GPU, to exercise the N > 1 code path on a 1-GPU box
    backend = os.environ.get("FF_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=backend)
    if args.gpus != world:
        # never report n_gpus != --gpus: a launcher that started the wrong number of ranks is an error
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
        sys.exit(2)
    if backend == "nccl" and torch.cuda.device_count() < (local_rank + 1):
        print(f"bench.py: rank {rank} needs cuda:{local_rank} but only {torch.cuda.device_count()} device(s) are visible",
              file=sys.stderr)
        sys.exit(2)
    n_gpus = world
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    model = G._model(dev, args.nup, args.ndown, args.Z)
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    B_glob = args.walkers_per_gpu * n_gpus
    torch.manual_seed(1234)      # same Philox key on every rank; streams are separated by the global walker index

    def step():
        gradE = model(B_glob)
        opt.zero_grad()
        gradE.backward()
        opt.step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    model.profile = {}
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    prof, model.profile = model.profile, None

    # ---- per-stage times and the dominant kernel's roofline (HIP events recorded on the launch stream)
    names = ["mcmc", "generate", "eloc", "estimator", "adjoint"]
    stages = {k: 0.0 for k in names}
    for ev in prof["events"]:
        prev = ev["t0"]
        for k in names:
            stages[k] += prev.elapsed_time(ev[k]); prev = ev[k]
    stages = {k: v / args.steps for k, v in stages.items()}
    k_ms = sum(a.elapsed_time(b) for a, b in prof["pass1"]) / args.steps
    evals = sum(int(s[0].item()) for s in prof["eloc_stats"]) / args.steps          # RHS evaluations summed over walkers
    n = args.nup + args.ndown
    M, R, H = 2 * n, n * (n - 1) // 2 + n, 50
    from fermiflow_amd import _lib as L
    radial = L.RADIAL_MODE
    P = n * (n - 1) // 2
    flop_per_eval = (R * ((H * FLOP_PER_SIGMOID_UNIT if radial == "exact" else FLOP_PER_TABLE_RADIUS) + FLOP_PER_RADIUS_RECORD)
                     + M * (P * FLOP_PER_PAIR_TERM + n * FLOP_PER_ONEBODY_TERM) + M * FLOP_PER_LANE_GATHER)
    achieved = evals * flop_per_eval / (k_ms * 1e-3) / 1e12
    roofline = {"kernel": "ff_ode_fwd_kernel<%d,2,2,%s> (local-energy sensitivities)" % (n, "true" if radial == "table" else "false"), "bound": "mfma",
                "note": "fp64 VALU-bound; on MI355X the fp64 vector and fp64 MFMA peaks coincide (78.6 TFLOP/s); "
                        "MFMA is not used: the MLP is 1->50->1 (no dense GEMM)",
                "achieved": achieved, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_TFLOPS,
                "traffic": None, "avg_launch_ms": k_ms, "rhs_evals_per_walker": evals / args.walkers_per_gpu,
                "flop_per_walker_eval": flop_per_eval, "radial_functions": radial}

    # HBM bytes of the dominant kernel: not measurable live; taken from the committed rocprofv3 PMC pass
    # (FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE), profiles/r01_k_hbm_traffic.json
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_k_hbm_traffic.json")))
        key = [k for k in tj if "ff_ode_fwd_kernel<%d, 2, 2, true>" % n in k]
        if key and radial == "table" and args.walkers_per_gpu == 65536:
            roofline["traffic"] = tj[key[0]]["hbm_bytes_fetchx2_plus_write"]
            roofline["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/r01_k_hbm_traffic.json), bytes per launch"
            roofline["algorithmic_bytes"] = args.walkers_per_gpu * 8 * (M + M * M + 4 * M + 1)
    except Exception:
        pass

    out = {"metric": "walker-steps/sec (full VMC iteration: 100 MCMC steps + generate + E_loc + grad + Adam)",
           "value": B_glob * 100 * args.steps / dt, "unit": "walker-steps/s", "n_gpus": n_gpus, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"GSVMC nup={args.nup} ndown={args.ndown} 2D Z={args.Z} H=50 t_span=(0,1) rtol=1e-6 atol=1e-8, "
                                  f"{args.walkers_per_gpu} walkers/GPU, 100 Metropolis steps/iter, seeded gaussian weights x(30,300)",
                      "global_walkers": B_glob, "parallelism": f"walker-dp{n_gpus}"},
           "E": model.E, "E_std": model.E_std, "stages_ms": stages, "roofline": roofline}

    if rank == 0 and not args.no_extras:
        # ---- the HBM-bound kernel of the path: parity-mode Metropolis sweep (noise streamed from HBM)
        Bm, S = args.walkers_per_gpu, 100
        g0, g, u = native.rng_fill(Bm, n, S, 7, dev)
        tu, td = model._tables(dev)
        native.mcmc_sample_noise(tu, td, args.nup, args.ndown, g0, g, u)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            native.mcmc_sample_noise(tu, td, args.nup, args.ndown, g0, g, u)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nbytes = Bm * S * (8 * M + 8 + 1) + Bm * (2 * 8 * M + 8)     # noise + uniforms + accept mask; init + final x, logp
        del g0, g, u
        # ---- the stand-alone pairwise kernels (potentials.py / equivariant_funs.py entry points; inside the sweep these
        #      terms are fused into the ODE kernels): Coulomb + trap energy is HBM-bound, backflow v + div is fp64-bound
        Bp = 16 * args.walkers_per_gpu
        xp = torch.randn(Bp, n, 2, dtype=torch.float64, device=dev)

        def timed(fn, reps=5):
            fn(); e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        ms_p = timed(lambda: native.potential(xp, args.Z, True))
        netp = model.cnf.v_wrapper.v.net(radial="exact")
        xb = xp[: args.walkers_per_gpu]
        ms_b = timed(lambda: native.backflow_v_div(netp, xb))
        bytes_p = Bp * (8 * M + 8)
        flop_b = args.walkers_per_gpu * R * (H * FLOP_PER_SIGMOID_UNIT + 20)
        out["roofline_pairwise"] = {
            "potential": {"kernel": "ff_potential_stream_kernel (HO + Coulomb pairs)", "bound": "hbm", "walkers": Bp, "avg_launch_ms": ms_p,
                          "achieved": bytes_p / (ms_p * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                          "frac": bytes_p / (ms_p * 1e-3) / 1e9 / PEAK_HBM_GBS},
            "backflow": {"kernel": "ff_backflow_kernel (v and div v, direct sigmoids)", "bound": "mfma", "walkers": args.walkers_per_gpu,
                         "avg_launch_ms": ms_b, "achieved": flop_b / (ms_b * 1e-3) / 1e12, "peak": PEAK_FP64_TFLOPS,
                         "unit": "TFLOP/s", "frac": flop_b / (ms_b * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
                         "note": "fp64 VALU (exp/rcp chains); no MFMA: 1->H->1 layers"}}
        del xp
        out["roofline_hbm"] = {"kernel": (f"ff_mcmc_spin_kernel<{args.nup},noise>" if args.nup == args.ndown and 1 <= args.nup <= 6 else f"ff_mcmc_kernel<{args.nup},{args.ndown},noise>"), "bound": "hbm",
                               "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None, "avg_launch_ms": ms,
                               "walker_steps_per_s": Bm * S / (ms * 1e-3)}
        # ---- CPU baseline: the oracle's full sweep on the host cores, bounded sample
        if n_gpus == 1 and args.cpu_walkers > 0:
            from oracle import oracle as O
            v = model.cnf.v_wrapper.v
            net = O.Net(tuple(t.detach().cpu().numpy() for t in (v.eta.fc1.weight, v.eta.fc1.bias, v.eta.fc2.weight)),
                        tuple(t.detach().cpu().numpy() for t in (v.mu.fc1.weight, v.mu.fc1.bias, v.mu.fc2.weight)))
            O.gsvmc_sweep(64, args.nup, args.ndown, net, args.Z, seed=1)      # thread-pool warm-up
            t1 = time.perf_counter()
            r = O.gsvmc_sweep(args.cpu_walkers, args.nup, args.ndown, net, args.Z, seed=2)
            ct = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": args.cpu_walkers * 100 / ct, "unit": "walker-steps/s", "cores": O.num_threads(),
                                   "kind": "port",
                                   "sample": f"one full iteration (same stages, minus Adam) of {args.cpu_walkers} walkers, "
                                             f"oracle/ff_oracle.c with OpenMP, {ct:.1f} s", "E": r["E"], "E_std": r["E_std"],
                                   "stage_seconds": r["seconds"]}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
