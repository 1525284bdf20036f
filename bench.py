#!/usr/bin/env python3
"""bench.py -- headline benchmark: walker-steps/s of one full VMC training iteration.

  python bench.py --gpus N --steps K --warmup W            (N > 1: this script starts the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (what the driver runs)

A "step" is one complete iteration of the reference training loop (src/FermionHO2D.py:66-72):
    gradE = model(batch); optimizer.zero_grad(); gradE.backward(); optimizer.step()
i.e. 100 Metropolis walker-steps per walker + CNF.generate + local energy + E/E_std + parameter gradient
(adjoint) + Adam.  Default workload = BASELINE.json configs[1]: nup = ndown = 3, 2-D, Z = 2.0, H = 50, fp64,
65 536 walkers PER GPU (weak scaling: walkers shard with no data-path collective, only the two tiny
estimator all-reduces of fermiflow_amd/dist.py).  value = n_gpus * walkers * 100 * K / time.
  --workload beta   BASELINE.json configs[2]: BetaFermionHO2D beta = 10, nup = 3, boltzmann, 65 536 walkers
  --workload n12    BASELINE.json configs[3]: nup = ndown = 6, 32 768 walkers per GPU
  --workload c5     BASELINE.json configs[4]: nup = ndown = 10 in a 3-D trap, 131 072 walkers per GPU (fp64; matrix-core kernel)
(parity-test cases; their lines are kept under profiles/, the driver's line is the default workload).
  --scaling strong  the global batch stays at --walkers-per-gpu (default 65 536) however many GPUs share it
                    (BASELINE.json words its metric "65536 walkers, 1/2/4/8 GPUs"); default: weak, that many walkers PER GPU.
A second, shorter timed leg (`trained_leg`) repeats the measurement after --trained-iters untimed training iterations at
lr 1e-4: as the flow strengthens the ODE solves take more steps (tools/probes/long_run.py), so the headline -- synthetic
weights held in place by a tiny learning rate -- is the best case; both numbers are on the line.

Extra objects on the JSON line (rank 0, N = 1): `roofline` for the dominant kernel (the fused local-energy
integration) with its live HIP-event time, its algorithmic flops and -- from four rocprofv3 --pmc child
passes of this very script -- its HBM traffic, VALU-active / waiting shares of the wave-cycles, LDS bank-conflict share and
matrix-core operation count; `roofline_adjoint`, `roofline_mcmc` for the other two kernels above
10 % of a step; `roofline_hbm` (parity-mode Metropolis sweep: the HBM-bound kernel of the path), `roofline_pairwise`
(stand-alone potential / backflow kernels); `stages_ms`; `cpu_baseline` (the C oracle on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per RHS evaluation of one walker (counted from the sources, FMA = 2 flop):
FLOP_PER_SIGMOID_UNIT = 30      # 1 exp + 1 rcp + ~8 FMA-class ops      (FERMIFLOW_RADIAL=exact)
FLOP_PER_TABLE_RADIUS = 60      # 4 heads x 5 Horner FMAs + index/offset arithmetic     (FERMIFLOW_RADIAL=table, default)
FLOP_PER_RADIUS_RECORD = 55     # r, 1/r, the direction-independent record and the own-row contributions of one radius
FLOP_PER_PAIR_TERM = 44         # column sweep: one (direction, pair) term of the second-order jet
FLOP_PER_ONEBODY_TERM = 40
FLOP_PER_LANE_GATHER = 30       # own rows (18 adds) + transposition sum (12 adds)
FLOP_PER_RADIUS_S = 70          # row-layout / matrix-core kernels: W from S, second-order sources of one radius
FLOP_ADJ_RADIUS = 85            # tabulated adjoint: 45 heads + 40 record and own rows per radius
FLOP_ADJ_LANE = 18              # own-row gather per coordinate
FLOP_MCMC_STEP_PER_PARTICLE = 50    # proposal, Hermite recurrences, one row of the determinant update, accept (per particle of a walker-step)
PEAK_FP64_TFLOPS = 78.6         # MI355X fp64 vector = fp64 matrix peak (vendor)
PEAK_FP32_TFLOPS = 157.3        # MI355X fp32 matrix = fp32 vector peak (MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 at 64 FLOP/clk/SIMD)
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


def eloc_kernel_name(n, d=2):
    """The local-energy kernel the dispatcher picks (csrc/ff_cnf_fwd.hip, dispatch_fwd; csrc/ff_wide.hip beyond 12 / 4 particles)."""
    kind = os.environ.get("FF_ELOC_KERNEL", "auto")
    if os.environ.get("FF_WIDE") == "1" or kind == "wide" or (d == 2 and n >= 11 and kind == "auto") \
            or (d == 2 and n > 12) or (d == 3 and n > 4):
        return "wide", f"ff_wide_eloc_kernel<{d}, {(n * d + 4 + 15) // 16}, true>"
    if d == 2 and 2 <= n <= 6 and (kind == "mfma" or (kind == "auto" and n >= 4)):
        return "mfma", f"ff_eloc_mfma_kernel<{n}, 2, true, 2>"
    split = {7: 2, 8: 2, 9: 3, 10: 3, 11: 2, 12: 2}.get(n, 1)
    if kind == "rows" or n in (1, 7, 9, 11) or (kind in ("auto", "mfma") and n >= 9):
        return "rows", f"ff_eloc_rows_kernel<{n}, 2, {split}, true, 1>"
    if n in (8, 10, 12):
        return "columns", f"ff_eloc_split_kernel<{n}, 2, true>"
    return "columns", f"ff_ode_fwd_kernel<{n}, 2, 2, true>"


def eloc_flop_per_eval(kind, n, H, radial, d=2):
    M, P = d * n, n * (n - 1) // 2
    R = P + n
    heads = H * FLOP_PER_SIGMOID_UNIT if radial == "exact" else FLOP_PER_TABLE_RADIUS
    if kind == "columns":
        return R * (heads + FLOP_PER_RADIUS_RECORD) + M * (P * FLOP_PER_PAIR_TERM + n * FLOP_PER_ONEBODY_TERM) + M * FLOP_PER_LANE_GATHER
    # rows / mfma: two dense M x M x M products (J' = A J and S = J J^T) + per-radius work
    return 2 * 2 * M * M * M + R * (heads + FLOP_PER_RADIUS_RECORD + FLOP_PER_RADIUS_S) + M * FLOP_PER_LANE_GATHER


PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",),
              ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_INSTS_VALU"),
              ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES"),
              ("SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_MFMA", "SQ_BUSY_CYCLES"))


def pmc_counters(kernel_substr, argv, passes=PMC_PASSES, child_args=("--steps", "2", "--warmup", "1", "--no-extras")):
    """Per-launch averages of hardware counters of one kernel, from `rocprofv3 --pmc` child passes of this script (counters
    in their own runs, no tracing; --steps 2 --no-extras).  Returns (dict counter -> value per launch, None) or (None, reason).
    kernel_substr may be a tuple: the counters of all those kernels are added up and divided by the launches of the FIRST one
    (a pass that runs as several kernels side by side: the routed local-energy pass)."""
    names = (kernel_substr,) if isinstance(kernel_substr, str) else tuple(kernel_substr)
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not found"
    out = {}
    for ctrs in passes:
        d = tempfile.mkdtemp(prefix="ffpmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        cmd = [exe, "--pmc"] + list(ctrs) + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__)] + argv + \
              list(child_args)
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
        except Exception as e:      # noqa: BLE001
            shutil.rmtree(d, ignore_errors=True)
            if ctrs[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                return None, f"rocprofv3 pass {ctrs} failed: {type(e).__name__}"
            continue
        tot, cnt = {c: 0.0 for c in ctrs}, {c: 0 for c in ctrs}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                c = row["Counter_Name"]
                if c in tot and any(nm in row["Kernel_Name"] for nm in names):
                    tot[c] += float(row["Counter_Value"])
                    cnt[c] += 1 if names[0] in row["Kernel_Name"] else 0
        shutil.rmtree(d, ignore_errors=True)
        for c in ctrs:
            if cnt[c]:
                out[c] = tot[c] / cnt[c]
        if ctrs[0] in ("FETCH_SIZE", "WRITE_SIZE") and ctrs[0] not in out:
            return None, f"kernel {names[0]} not in the {ctrs[0]} pass"
    return out, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the GPU reaches its steady clocks only after some ten iterations (tools/probes/warmup_sweep.sh: 1.93 ms per iteration
    # at --steps 5 --warmup 2, 1.89 at 10/2, 1.84 at 20/2, 1.81 at 20/5 and 20/10: every stage shrinks by the same factor)
    # (a 20-step window is 31 ms, two scheduler hiccups wide -- a bare run now times 200 steps; when the caller asks
    # for fewer, a `long_window_leg` of 200 steps rides on the same line)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["gsvmc", "beta", "n12", "c5"], default="gsvmc")
    ap.add_argument("--walkers-per-gpu", type=int, default=0, help="default: 65536 (gsvmc, beta), 32768 (n12), 131072 (c5)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --walkers-per-gpu walkers on EVERY GPU; strong: that many walkers in total, split over the GPUs")
    ap.add_argument("--sens-bits", type=int, default=0, choices=[0, 32, 64],
                    help="precision of the sensitivity matrices J, A, S in the matrix-core local-energy kernel (11 particles and more): "
                         "64, or 32 = the fp32 MFMA path BASELINE.json configs[4] names; default: 32 for --workload c5, 64 otherwise")
    ap.add_argument("--trained-iters", type=int, default=300,
                    help="untimed training iterations at lr 1e-4 in front of the second timed leg (0 = no second leg)")
    ap.add_argument("--nup", type=int, default=0)
    ap.add_argument("--ndown", type=int, default=-1)
    ap.add_argument("--Z", type=float, default=2.0)
    ap.add_argument("--lr", type=float, default=2e-5,
                    help="Adam step; small so the synthetic weights (hence the ODE step counts) stay put over the run "
                         "(the reference's 1e-2 from these weights changes eta by 250 %% per step; --lr 1e-2 runs it)")
    ap.add_argument("--cpu-walkers", type=int, default=4096, help="sample size of the CPU baseline (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the stand-alone kernel legs, the PMC passes and the CPU baseline")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline.traffic = null)")
    ap.add_argument("--standalone-leg", action="store_true",
                    help="(child passes) run only the stand-alone HBM-bound kernels -- parity-mode Metropolis sweep, potential -- three times and exit")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Launched bare (`python bench.py --gpus N`): this parent starts the N ranks itself -- before anything here has
        # touched a GPU -- and relays rank 0's JSON line.
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    import __graft_entry__ as G
    import fermiflow_amd as ff
    from fermiflow_amd import native
    from fermiflow_amd import _lib as L

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

    wl = args.workload
    nup = args.nup or {"gsvmc": 3, "beta": 3, "n12": 6, "c5": 10}[wl]
    ndown = args.ndown if args.ndown >= 0 else {"gsvmc": 3, "beta": 0, "n12": 6, "c5": 10}[wl]
    wpg = args.walkers_per_gpu or {"gsvmc": 65536, "beta": 65536, "n12": 32768, "c5": 131072}[wl]
    n = nup + ndown
    dim = 3 if wl == "c5" else 2
    gs = G._model(dev, nup, ndown, args.Z) if dim == 2 else G._model(dev, 2, 2, args.Z)
    if wl == "c5":      # the 3-D trap: HO3D orbitals (closed shells 0..2), same flow network
        model = ff.GSVMC(nup, ndown, ff.HO3D(), ff.FreeFermion(device=dev), gs.cnf, ff.CoulombPairPotential(args.Z), sp_potential=ff.HO())
    elif wl == "beta":
        model = ff.BetaVMC(10.0, nup, ndown, 2.0, True, ff.HO2D(), ff.FreeFermion(device=dev), gs.cnf,
                           ff.CoulombPairPotential(args.Z), sp_potential=ff.HO())
        model.to(dev)
    else:
        model = gs
    sens_bits = args.sens_bits or (32 if wl == "c5" else 64)
    native.set_sens_precision(sens_bits)
    from fermiflow_amd.utils import make_adam
    opt = make_adam(model.parameters(), lr=args.lr)
    if args.scaling == "strong":      # fixed global batch: every rank takes 1 / n_gpus of it
        B_glob, wpg = wpg, -(-wpg // n_gpus)
    else:
        B_glob = wpg * n_gpus
    torch.manual_seed(1234)      # same Philox key on every rank; streams are separated by the global walker index

    if args.standalone_leg:      # what roofline_hbm / roofline_pairwise time, for the counter passes (rocprofv3 --pmc -- bench.py --standalone-leg)
        tu_, td_ = model._tables(dev)
        g0, g, u = native.rng_fill(wpg, n, 100, 7, dev)
        xp = torch.randn(16 * wpg, n, 2, dtype=torch.float64, device=dev)
        for _ in range(3):
            native.mcmc_sample_noise(tu_, td_, nup, ndown, g0, g, u)
            native.potential(xp, args.Z, True)
        torch.cuda.synchronize()
        return

    def step():
        g = model(B_glob)
        opt.zero_grad()
        if wl == "beta":
            g[0].backward(); g[1].backward()
        else:
            g.backward()
        opt.step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # The timed region carries ONE event pair per step (around the local-energy pass: roofline.avg_launch_ms is measured live, on the
    # launch stream, inside the region the headline is measured in).  The five stage markers stay out of it: each is a hipEventRecord
    # -- about 8 us of pipeline bubble on this GPU, 3 % of an iteration together (tools/probes/host_ahead.py) -- and stages_ms comes
    # from a second loop of the same length right behind the timed one.
    model.profile = {"stages": False}
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
    E_head, Estd_head = model.E, model.E_std
    _v = gs.cnf.v_wrapper.v      # the headline's weights, for the CPU baseline (the later legs train them on)
    w_head = (tuple(t.detach().cpu().numpy().copy() for t in (_v.eta.fc1.weight, _v.eta.fc1.bias, _v.eta.fc2.weight)),
              tuple(t.detach().cpu().numpy().copy() for t in (_v.mu.fc1.weight, _v.mu.fc1.bias, _v.mu.fc2.weight)))
    beta_head = (model.F, model.F_std, model.S) if wl == "beta" else None
    same = None
    if rank == 0 and n_gpus == 1 and not args.no_extras and args.cpu_walkers > 0 and wl in ("gsvmc", "n12"):
        # One untimed sweep on EXACTLY these weights (no Adam step behind it): its first walkers and their local energies are what the
        # CPU baseline's oracle is given further down, so that E / E_std of both are estimates over the same sample
        with torch.no_grad():
            model(B_glob)
        ns = args.cpu_walkers if wl == "gsvmc" else max(256, args.cpu_walkers // 8)
        same = (model.x[:ns].cpu().numpy().copy(), model.Eloc[:ns].cpu().numpy().copy())
    model.profile = {}
    fence()
    t0s = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt_marked = time.perf_counter() - t0s
    prof_stages, model.profile = model.profile, None

    # ---- the same loop over >= 200 steps (only when the caller's K is shorter): the headline's window at K = 20 is 31 ms
    long_leg = None
    if args.steps < 200 and wl != "c5" and not args.no_extras:
        kl = 200
        model.profile = {"stages": False}
        fence()
        t0l = time.perf_counter()
        for _ in range(kl):
            step()
        fence()
        dtl = time.perf_counter() - t0l
        if world > 1:
            tt = torch.tensor([dtl], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtl = tt.item()
        pl, model.profile = model.profile, None
        long_leg = {"steps": kl, "ms_per_step": dtl / kl * 1e3, "value": B_glob * 100 * kl / dtl, "E": model.E, "E_std": model.E_std,
                    "rhs_evals_per_walker": sum(int(st_[0].item()) for st_ in pl["eloc_stats"]) / kl / wpg,
                    "eloc_pass_ms": sum(a.elapsed_time(b) for a, b in pl["pass1"]) / kl,
                    "note": "the headline loop continued for 200 more steps (behind the stage-marker loop), same lr, same instrumentation; the synthetic "
                            "weights train even at this lr and the flow they become needs more steps -- rhs_evals_per_walker says how many"}

    # ---- reference-semantics leg (VERDICT r03 next #3c): one tolerance for every component of every walker (sens_tol = 1), no
    #      routing by cost class (heavy_class < 0), Hairer's cold start for all three integrations -- what the policy in
    #      config.workload buys, and what it costs in accuracy: max E_loc error of both against a 1e-11 solve on the same base walkers
    def eloc_error(z, mdl=None):
        """max over the batch of |E_loc - E_loc(1e-11 solve)| / |E_loc| of the production sweep (mdl's policy) on base walkers z"""
        mdl = mdl or model
        g_ = mdl.forward_from(z, batch=B_glob)
        x_, e_ = mdl.x, mdl.Eloc.clone()
        tu_, td_ = mdl._tables(dev)
        tight = native.eloc(tu_, td_, nup, ndown, mdl.cnf.v_wrapper.v.net(), x_, 0.0, 1.0, 1e-11, 1e-13, args.Z, True)["eloc"]
        del g_
        return ((e_ - tight).abs() / tight.abs()).max().item()

    ref_leg = None
    if wl == "gsvmc" and not args.no_extras:
        with torch.no_grad():
            zr = model.basedist.sample(model.orbitals_up, model.orbitals_down, (wpg,))
        err_policy = eloc_error(zr)
        saved = (model.sens_tol, model.heavy_class, model.warm_start)
        model.sens_tol, model.heavy_class, model.warm_start = 1.0, -1, False      # (warm_start off: no learned first steps either)
        err_ref = eloc_error(zr)
        kr = max(2, args.steps // 2)
        for _ in range(3):
            step()
        model.profile = {}
        fence()
        t0 = time.perf_counter()
        for _ in range(kr):
            step()
        fence()
        dtr = time.perf_counter() - t0
        pr, model.profile = model.profile, None
        model.sens_tol, model.heavy_class, model.warm_start = saved
        ref_leg = {"policy": "sens_tol = 1 (one tolerance for every component), no routing by cost class, cold (Hairer) start of every integration",
                   "steps": kr, "ms_per_step": dtr / kr * 1e3, "value": B_glob * 100 * kr / dtr,
                   "rhs_evals_per_walker": sum(int(st_[0].item()) for st_ in pr["eloc_stats"]) / kr / wpg,
                   "eloc_max_rel_err_vs_1e-11_solve": err_ref, "headline_policy_eloc_max_rel_err": err_policy,
                   "bar": 1e-5}
        for _ in range(2):
            step()      # (the warm-start state of the headline policy again, before the next leg)

    # ---- configs[3] / configs[4]: the tolerance policy's largest E_loc error against a 1e-11 solve on this line too
    #      (fp64 sensitivity matrices for the reference solve whatever --sens-bits the sweep runs at), two fresh batches
    policy_err = None
    if wl in ("n12", "c5") and not args.no_extras and world == 1:
        nsub = min(wpg, 32768 if wl == "n12" else 16384)
        errs = []
        for sd in (31, 32):
            torch.manual_seed(sd)
            with torch.no_grad():
                zr = model.basedist.sample(model.orbitals_up, model.orbitals_down, (nsub,))
            model.forward_from(zr, batch=nsub)
            x_, e_ = model.x, model.Eloc.clone()
            tu_, td_ = model._tables(dev)
            native.set_sens_precision(64)
            tight = native.eloc(tu_, td_, nup, ndown, model.cnf.v_wrapper.v.net(), x_, 0.0, 1.0, 1e-11, 1e-13, args.Z, True)["eloc"]
            native.set_sens_precision(sens_bits)
            errs.append(((e_ - tight).abs() / tight.abs()).max().item())
        policy_err = {"headline_policy_eloc_max_rel_err": max(errs), "headline_policy_eloc_max_rel_err_by_batch": errs, "walkers": nsub, "bar": 1e-5}
        for _ in range(2):
            step()

    # ---- second leg: the same measurement on a flow that has been trained for a while (the headline's weights are held in
    #      place by the tiny learning rate: its ODE step counts are the best case)
    trained = None
    n_tr = args.trained_iters if wl != "c5" else 0
    if n_tr > 0 and not args.no_extras:
        for g in opt.param_groups:
            g["lr"] = 1e-4
        for _ in range(n_tr):
            step()
        for g in opt.param_groups:
            g["lr"] = args.lr
        k2 = max(2, args.steps // 2)
        step()
        model.profile = {}
        fence()
        t0 = time.perf_counter()
        for _ in range(k2):
            step()
        fence()
        dt2 = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt2], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt2 = tt.item()
        p2, model.profile = model.profile, None
        ev2 = sum(int(st_[0].item()) for st_ in p2["eloc_stats"]) / k2
        trained = {"after": f"{n_tr} training iterations at Adam lr=1e-4 from the headline's weights (untimed), then lr={args.lr}",
                   "steps": k2, "ms_per_step": dt2 / k2 * 1e3, "value": B_glob * 100 * k2 / dt2,
                   "rhs_evals_per_walker": ev2 / wpg, "eloc_kernel_ms": sum(a.elapsed_time(b) for a, b in p2["pass1"]) / k2,
                   "E": model.E, "E_std": model.E_std}
        if wl == "gsvmc":      # the tolerance policy's error on THESE weights (VERDICT r04 next #1a), three fresh batches
            errs = []
            for sd in (11, 12, 13):
                torch.manual_seed(sd)
                with torch.no_grad():
                    zt = model.basedist.sample(model.orbitals_up, model.orbitals_down, (wpg,))
                errs.append(eloc_error(zt))
            trained["headline_policy_eloc_max_rel_err"] = max(errs)
            trained["headline_policy_eloc_max_rel_err_by_batch"] = errs
            trained["bar"] = 1e-5

    # ---- driver leg (VERDICT r04 next #1a; SURVEY 8(d) weights (i)): what a user of the drop-in driver runs -- init_zeros() flow
    #      (src/FermionHO2D.py:40-43), Adam lr = 1e-2 (:61), the training loop of :66-72 for 300 iterations on this workload's walkers
    driver = None
    if wl == "gsvmc" and not args.no_extras:
        dm = G._model(dev, nup, ndown, args.Z)
        dv = dm.cnf.v_wrapper.v
        dv.eta.init_zeros(); dv.mu.init_zeros()
        dm.to(dev)
        dopt = make_adam(dm.parameters(), lr=1e-2)
        torch.manual_seed(4321)

        def dstep():
            g_ = dm(B_glob); dopt.zero_grad(); g_.backward(); dopt.step()
        windows = {}
        it = 0
        for label, upto, width in (("iter_1", 10, 10), ("iter_100", 105, 10), ("iter_300", 300, 10)):
            while it < upto - width:
                dstep(); it += 1
            dm.profile = {"stages": False}
            fence()
            t0d = time.perf_counter()
            for _ in range(width):
                dstep(); it += 1
            fence()
            dtd = time.perf_counter() - t0d
            if world > 1:
                tt = torch.tensor([dtd], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dtd = tt.item()
            pd, dm.profile = dm.profile, None
            windows[label] = {"iterations": f"{it - width + 1}-{it}", "ms_per_step": dtd / width * 1e3, "value": B_glob * 100 * width / dtd,
                              "rhs_evals_per_walker": sum(int(st_[0].item()) for st_ in pd["eloc_stats"]) / width / wpg,
                              "eloc_pass_ms": sum(a.elapsed_time(b) for a, b in pd["pass1"]) / width, "E": dm.E, "E_std": dm.E_std}
        errs = []
        for sd in (21, 22, 23):
            torch.manual_seed(sd)
            with torch.no_grad():
                zt = dm.basedist.sample(dm.orbitals_up, dm.orbitals_down, (wpg,))
            errs.append(eloc_error(zt, dm))
        driver = {"what": "init_zeros() flow, Adam lr=1e-2, 300 iterations of the reference's training loop (src/FermionHO2D.py:40-43,61-72) at "
                          f"{wpg} walkers/GPU, Z={args.Z}; windows of 10 iterations",
                  **windows, "max_abs_fc1_weight": max(dv.eta.fc1.weight.abs().max().item(), dv.mu.fc1.weight.abs().max().item()),
                  "headline_policy_eloc_max_rel_err": max(errs), "headline_policy_eloc_max_rel_err_by_batch": errs, "bar": 1e-5}
        del dm, dopt

    # ---- per-stage times and the dominant kernel's roofline (HIP events recorded on the launch stream)
    names = ["mcmc", "generate", "eloc", "estimator", "adjoint"]
    stages = {k: 0.0 for k in names}
    for ev in prof_stages["events"]:
        prev = ev["t0"]
        for k in names:
            stages[k] += prev.elapsed_time(ev[k]); prev = ev[k]
    stages = {k: v / args.steps for k, v in stages.items()}
    k_ms = sum(a.elapsed_time(b) for a, b in prof["pass1"]) / args.steps
    evals = sum(int(s[0].item()) for s in prof["eloc_stats"]) / args.steps          # RHS evaluations summed over walkers
    M, R, H = dim * n, n * (n - 1) // 2 + n, 50
    radial = L.RADIAL_MODE
    kind, kname = eloc_kernel_name(n, dim)
    flop_per_eval = eloc_flop_per_eval(kind, n, H, radial, dim)
    achieved = evals * flop_per_eval / (k_ms * 1e-3) / 1e12
    f32_path = kind == "wide" and sens_bits == 32 and (n * dim + 4 + 15) // 16 >= 2
    if f32_path:
        kname = kname.replace("true>", "true, float>")
    elif kind == "wide":
        kname = kname.replace("true>", "true, double>")
    peak = PEAK_FP32_TFLOPS if f32_path else PEAK_FP64_TFLOPS
    roofline = {"kernel": kname + " (local-energy sensitivities)",
                "bound": "mfma" if kind in ("mfma", "wide") else "fp64-valu",
                "note": ("one walker per workgroup; J' = A J and S = J J^T on " + ("v_mfma_f32_16x16x4 (fp32 sensitivity matrices: priced against the fp32 "
                         "matrix peak)" if f32_path else "v_mfma_f64_16x16x4") + " (2 x 2 M^3 of the priced flops), the rest fp64 VALU"
                         if kind == "wide" else ("four walkers per wave, two waves per SIMD; J' = A J and S = J J^T on v_mfma_f64_4x4x4 (2 x 2 M^3 of the priced flops), the rest fp64 VALU; "
                          "the flops are those of the S = J J^T formulation this kernel runs (the column sweep of rounds 1-2 priced 13575 per evaluation at 6 particles); "
                          "avg_launch_ms spans the whole pass: the walkers of the highest cost classes (>= 16 at 12 coordinates: 0.04 %; >= 12 below) run beside it on the one-walker-per-wave kernel "
                          "(ff_wide_eloc_kernel<2, 1, true, double, true>, which finishes them too; DESIGN.md 3g), their evaluations are in the count") if kind == "mfma" else
                         "fp64 VALU (instruction-issue) bound: the schema's hbm|mfma do not describe it; no MFMA is issued "
                         "(the MLPs are 1->H->1; FF_ELOC_KERNEL=mfma selects the matrix-core variant of this kernel)") +
                        ("; peak = MI355X fp32 matrix peak" if f32_path else "; peak = MI355X fp64 vector = fp64 matrix peak"),
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": None, "avg_launch_ms": k_ms, "rhs_evals_per_walker": evals / wpg,
                "flop_per_walker_eval": flop_per_eval, "radial_functions": radial,
                # with the fused finish (matrix-core kernel, nup = ndown) J^T stays on chip: x in; z, grad, grad_z logp0 and five scalars out
                # x in; z, grad logp, grad_z logp0 and five scalars out -- for EVERY kernel (the M^2 workspace the kernels
                # without a fused finish write and re-read is traffic, not algorithm; `workspace_bytes` states it)
                "algorithmic_bytes": wpg * 8 * (4 * M + 5),
                "workspace_bytes": 0 if (kind == "mfma" and nup == ndown) else wpg * 8 * (M * M + 4 * M + 1)}
    # every kernel of the pass between the two events (routing: the heavy walkers' kernel runs beside the throughput kernel): their
    # counters are added up
    pass_kernels = [kname.split("<")[0] + "<" + kname.split("<")[1].split(">")[0]]
    if kind in ("mfma", "columns") and dim == 2 and n <= 6 and model.heavy_class >= 0 and radial == "table":
        # (beside a throughput kernel with the fused finish the heavy kernel finishes its walkers itself: its FIN instantiation)
        pass_kernels += ["ff_wide_eloc_kernel<2, 1, true, double, true>" if (kind == "mfma" and nup == ndown) else "ff_wide_eloc_kernel<2, 1, true, double, false>"]
    roofline["kernels_of_the_pass"] = pass_kernels

    out = {"metric": "walker-steps/sec (full VMC iteration: 100 MCMC steps + generate + E_loc + grad + Adam)",
           "value": B_glob * 100 * args.steps / dt, "unit": "walker-steps/s", "n_gpus": n_gpus, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
           "vs_baseline": None, "dtype": "f64" if not f32_path else "f64 (sensitivity matrices J, A, S of the local-energy pass: f32)", "data": "synthetic",
           "config": {"workload": ("BetaVMC beta=10 boltzmann deltaE=2 " if wl == "beta" else "GSVMC ") +
                                  f"nup={nup} ndown={ndown} {dim}D Z={args.Z} H=50 t_span=(0,1) rtol=1e-6 atol=1e-8 "
                                  + (f"(sensitivity components of walkers with flow cost class <= {model.sens_tol_class}: x{model.sens_tol:g}; " if model.sens_tol > 1.0
                                     else "(one tolerance for every component; ") +
                                  (f"first step of the local-energy pass by cost class from the learned table; " if getattr(model, "adaptive_h", False) else "")
                                  + (f"walkers of class >= {model.heavy_class or (16 if n * dim >= 12 else 12)}: x{model.heavy_tol or 0.3:g} on the one-walker-per-wave kernel; "
                                     if (dim == 2 and n <= 6 and model.heavy_class >= 0) else "") +
                                  f"step-size warm start {'on' if model.warm_start else 'off'}; "
                                  f"walker prefetch {'on' if getattr(model, 'prefetch_walkers', False) else 'off'}), "
                                  f"{wpg} walkers/GPU, 100 Metropolis steps/iter, seeded gaussian weights x(30,300), Adam lr={args.lr}",
                      "global_walkers": B_glob, "parallelism": f"walker-dp{n_gpus}"},
           "E": E_head, "E_std": Estd_head, "stages_ms": stages,
           "stages_note": "stage markers (hipEventRecord, ~8 us of pipeline bubble each) are not in the timed region: stages_ms is a second loop "
                          f"of {args.steps} steps right behind it, which ran at {dt_marked / args.steps * 1e3:.4f} ms per step with its markers",
           "roofline": roofline}
    if long_leg is not None:
        out["long_window_leg"] = long_leg
        # the same loop over a 200-step window, next to `value` (the K-step headline is a ~30 ms window on weights that
        # have not moved yet; this is the rate the loop settles at)
        out["value_200"] = long_leg["value"]
        out["ms_per_step_200"] = long_leg["ms_per_step"]
    # E_std is the standard deviation of an estimator with a log-divergent variance (1/r tails: one near-coalescence walker can double
    # it): the legs' values side by side, their median and their maximum, instead of one unstable number
    std_legs = {"headline": Estd_head}
    if long_leg is not None:
        std_legs["long_window"] = long_leg["E_std"]
    if trained is not None:
        std_legs["trained"] = trained["E_std"]
    if driver is not None:
        for k_ in ("iter_100", "iter_300"):
            std_legs["driver_" + k_] = driver[k_]["E_std"]
    vals_ = sorted(v_ for v_ in std_legs.values() if v_ == v_)
    if vals_:
        out["E_std_legs"] = {"median": vals_[len(vals_) // 2] if len(vals_) % 2 else 0.5 * (vals_[len(vals_) // 2 - 1] + vals_[len(vals_) // 2]),
                             "max": vals_[-1], "by_leg": std_legs}
    if trained is not None:
        out["trained_leg"] = trained
    if driver is not None:
        out["driver_leg"] = driver
    if world > 1:
        # ---- the communicator the N > 1 numbers were measured on: backend, ranks, devices, and the two all-reduces
        #      a sweep adds (four estimator sums; the 300-double gradient) timed alone, 100 repetitions each
        comm = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "collectives_per_iteration": 2,
                "rccl": (backend == "nccl")}
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.get_device_name(dev), "uuid": str(getattr(props, "uuid", "")),
                "pci_bus_id": getattr(props, "pci_bus_id", None), "gcn_arch": getattr(props, "gcnArchName", "")}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        comm["ranks"] = gathered
        from fermiflow_amd import dist as Dm
        for label, nel in (("allreduce_4_doubles_us", 4), ("allreduce_300_doubles_us", 300)):
            buf = torch.ones(nel, dtype=torch.float64, device=dev)
            for _ in range(10):
                Dm.all_reduce_sum_(buf)
            fence()
            t0c = time.perf_counter()
            for _ in range(100):
                Dm.all_reduce_sum_(buf)
            torch.cuda.synchronize()
            tt = torch.tensor([(time.perf_counter() - t0c) / 100 * 1e6], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            comm[label] = tt.item()
        out["comm"] = comm
    if ref_leg is not None:
        out["reference_semantics_leg"] = ref_leg
    if policy_err is not None:
        out["policy_error"] = policy_err
    if wl == "beta":
        out.update(F=beta_head[0], F_std=beta_head[1], S=beta_head[2])

    if rank == 0 and n_gpus == 1 and not args.no_extras:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timed(fn, reps=5):
            fn(); e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        v = gs.cnf.v_wrapper.v
        net = v.net()
        tu, td = (model._state_tables(dev) if wl == "beta" else model._tables(dev))
        ws = model._ws if wl == "beta" else None
        if dim == 2:      # (the 3-D sampler is another entry point; config 5 reports its stages in stages_ms)
            # ---- the adjoint and the production Metropolis kernel stand-alone, on the last sweep's walkers
            with torch.no_grad():
                z0 = native.mcmc_sample(tu, td, nup, ndown, wpg, 100, 0.1, 77, dev, walker_state=ws)[0]
                ms_mc = timed(lambda: native.mcmc_sample(tu, td, nup, ndown, wpg, 100, 0.1, 77, dev, walker_state=ws))
                hg = torch.empty(wpg, dtype=torch.float64, device=dev); he = torch.empty_like(hg)
                x = native.cnf_generate(net, z0, 0.0, 1.0, 1e-6, 1e-8, walker_h_out=hg)
                r = native.eloc(tu, td, nup, ndown, net, x, 0.0, 1.0, 1e-6, 1e-8, args.Z, True, walker_state=ws,
                                walker_h_init=hg, walker_h_scale=model._h_scale_eloc, walker_h_out=he)
                w = (r["eloc"] - r["eloc"].mean()) / wpg
                az, ad = w[:, None, None] * r["glogp0"], -w
                st = native.cnf_adjoint(net, r["z"], az, ad, 0.0, 1.0, 1e-6, 1e-8, need_gx=False, want_stats=True,
                                        walker_h_init=he, walker_h_scale=1.25)[2]
                adj_evals = int(st[0].item())
                ms_adj = timed(lambda: native.cnf_adjoint(net, r["z"], az, ad, 0.0, 1.0, 1e-6, 1e-8, need_gx=False,
                                                          walker_h_init=he, walker_h_scale=1.25))
            flop_adj = R * FLOP_ADJ_RADIUS + M * FLOP_ADJ_LANE
            a_adj = adj_evals * flop_adj / (ms_adj * 1e-3) / 1e12
            out["roofline_adjoint"] = {
                "kernel": f"ff_ode_adjtab_kernel<{n}, 2, 2> + deposit reduce/contract (theta-gradient adjoint, whole ff_cnf_adjoint call)",
                "bound": "fp64-valu", "achieved": a_adj, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": a_adj / PEAK_FP64_TFLOPS,
                "avg_launch_ms": ms_adj, "rhs_evals_per_walker": adj_evals / wpg, "flop_per_walker_eval": flop_adj,
                "note": "LDS-atomic and latency bound: two independent waves per workgroup share the deposit table (ticketed deposits); per "
                        "accepted step the 5 stage records of a radius are re-expanded about one node and deposited once (12 LDS atomics; rounds 1-4: "
                        "two rows); stand-alone call with a cold start (17 evaluations) -- inside the iteration the kernel runs 13 evaluations"}
            ratio_kernel = (nup == ndown and 1 <= nup <= 6) or (ndown == 0 and 2 <= nup <= 6)
            # the determinant-ratio kernels (round 4) do less than the reference's step: no exp per particle, no log per determinant
            # -- priced at what they execute: proposal 4 n, r^2 sums 4 n, polynomial rows ~5 n, determinants ~n_s^3,
            # one exp (~30), fp32 Box-Muller 12 per pair of normals
            flop_mc = (n * FLOP_MCMC_STEP_PER_PARTICLE) if not ratio_kernel else (4 * n + 4 * n + 5 * n + 2 * max(nup, ndown) ** 3 + 60 + 12 * n)
            a_mc = wpg * 100 * flop_mc / (ms_mc * 1e-3) / 1e12
            out["roofline_mcmc"] = {
                "kernel": (f"ff_mcmc_spin_philox_kernel<{nup}>" if nup == ndown and 1 <= nup <= 6 else
                           (f"ff_mcmc_pair_kernel<{nup}, false>" if ratio_kernel else f"ff_mcmc_kernel<{nup}, {ndown}, false>")) +
                          " (Philox + fp32 Box-Muller on chip, determinant-ratio accept test: the production sampler)",
                "bound": "fp64-valu", "achieved": a_mc, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": a_mc / PEAK_FP64_TFLOPS,
                "avg_launch_ms": ms_mc, "walker_steps_per_s": wpg * 100 / (ms_mc * 1e-3), "flop_per_walker_step": flop_mc,
                "hbm_bytes_per_walker": 2 * (8 * M + 8),
                "note": "instruction-issue bound, half of the instructions are the generator's integer rounds (two Philox4x32-10 blocks per lane "
                        "and step); HBM sees 208 B per walker per SWEEP, i.e. nothing"}
        if wl == "gsvmc":
            # ---- the HBM-bound kernel of the path: parity-mode Metropolis sweep (noise streamed from HBM)
            S = 100
            g0, g, u = native.rng_fill(wpg, n, S, 7, dev)
            ms = timed(lambda: native.mcmc_sample_noise(tu, td, nup, ndown, g0, g, u))
            nbytes = wpg * S * (8 * M + 8 + 1) + wpg * (2 * 8 * M + 8)     # noise + uniforms + accept mask; init + final x, logp
            del g0, g, u
            out["roofline_hbm"] = {"kernel": (f"ff_mcmc_spin_kernel<{nup}, true>" if nup == ndown and 1 <= nup <= 6 else f"ff_mcmc_kernel<{nup}, {ndown}, true>") + " (parity mode: explicit noise)",
                                   "bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None, "algorithmic_bytes": nbytes, "avg_launch_ms": ms,
                                   "walker_steps_per_s": wpg * S / (ms * 1e-3)}
            # ---- the stand-alone pairwise kernels (potentials.py / equivariant_funs.py entry points; inside the sweep these
            #      terms are fused into the ODE kernels): Coulomb + trap energy is HBM-bound, backflow v + div is fp64-bound
            Bp = 16 * wpg
            xp = torch.randn(Bp, n, 2, dtype=torch.float64, device=dev)
            ms_p = timed(lambda: native.potential(xp, args.Z, True))
            netp = v.net(radial="exact")
            xb = xp[:wpg]
            ms_b = timed(lambda: native.backflow_v_div(netp, xb))
            bytes_p = Bp * (8 * M + 8)
            flop_b = wpg * R * (H * FLOP_PER_SIGMOID_UNIT + 20)
            out["roofline_pairwise"] = {
                "potential": {"kernel": "ff_potential_stream_kernel (HO + Coulomb pairs)", "bound": "hbm", "walkers": Bp, "avg_launch_ms": ms_p,
                              "achieved": bytes_p / (ms_p * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                              "frac": bytes_p / (ms_p * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None, "algorithmic_bytes": bytes_p},
                "backflow": {"kernel": "ff_backflow_kernel (v and div v, direct sigmoids)", "bound": "fp64-valu", "walkers": wpg,
                             "avg_launch_ms": ms_b, "achieved": flop_b / (ms_b * 1e-3) / 1e12, "peak": PEAK_FP64_TFLOPS,
                             "unit": "TFLOP/s", "frac": flop_b / (ms_b * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
                             "note": "fp64 VALU (exp/rcp chains); no MFMA: 1->H->1 layers"}}
            del xp
            # counters of the two HBM-bound kernels (north star: "rocprof must show achieved HBM GB/s on the pairwise kernel"):
            # FETCH_SIZE / WRITE_SIZE passes of `bench.py --standalone-leg`, bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950)
            if not args.no_pmc and os.environ.get("FF_BENCH_CHILD") != "1":
                os.environ["FF_BENCH_CHILD"] = "1"
                argv0 = ["--workload", wl, "--walkers-per-gpu", str(wpg), "--Z", str(args.Z), "--nup", str(nup), "--ndown", str(ndown)]
                for tgt, kn, ms_k in ((out["roofline_hbm"], out["roofline_hbm"]["kernel"].split(" (")[0], ms),
                                      (out["roofline_pairwise"]["potential"], "ff_potential_stream_kernel", ms_p)):
                    ctr, why = pmc_counters(kn, argv0, passes=(("FETCH_SIZE",), ("WRITE_SIZE",)), child_args=("--standalone-leg",))
                    if ctr is None:
                        tgt["traffic_source"] = f"unavailable: {why}"
                    else:
                        tgt["traffic"] = (2.0 * ctr["FETCH_SIZE"] + ctr["WRITE_SIZE"]) * 1024.0
                        tgt["hbm_GBs_from_counters"] = tgt["traffic"] / (ms_k * 1e-3) / 1e9
                        tgt["traffic_source"] = "rocprofv3 --pmc child passes (bench.py --standalone-leg): bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950)"
                del os.environ["FF_BENCH_CHILD"]
        # ---- counters of the dominant kernel, measured now: rocprofv3 --pmc child passes of this script (HBM bytes = 2 x FETCH_SIZE
        #      + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md; then the issue / wait / LDS / matrix-core counters)
        if not args.no_pmc and os.environ.get("FF_BENCH_CHILD") != "1":
            os.environ["FF_BENCH_CHILD"] = "1"
            argv = ["--workload", wl, "--walkers-per-gpu", str(wpg), "--Z", str(args.Z), "--lr", str(args.lr), "--nup", str(nup),
                    "--ndown", str(ndown), "--sens-bits", str(sens_bits)]
            ctr, why = pmc_counters(tuple(pass_kernels), argv)
            if ctr is None:
                roofline["traffic_source"] = f"unavailable: {why}"
            else:
                roofline["traffic"] = (2.0 * ctr["FETCH_SIZE"] + ctr["WRITE_SIZE"]) * 1024.0
                roofline["traffic_source"] = ("rocprofv3 --pmc child passes of this run (counters per launch): bytes = 2 x FETCH_SIZE + WRITE_SIZE "
                                              "(KB; gfx950 correction)")
                roofline["pmc_per_launch"] = ctr
                wc = ctr.get("SQ_WAVE_CYCLES")
                if wc:
                    roofline["valu_active"] = ctr.get("SQ_ACTIVE_INST_VALU", 0.0) / wc      # share of the wave-cycles a VALU instruction is executing
                    roofline["wait_any"] = ctr.get("SQ_WAIT_ANY", 0.0) / wc
                if ctr.get("SQ_LDS_IDX_ACTIVE"):
                    roofline["lds_bank_conflict"] = ctr.get("SQ_LDS_BANK_CONFLICT", 0.0) / ctr["SQ_LDS_IDX_ACTIVE"]
                if "SQ_INSTS_VALU_MFMA_MOPS_F64" in ctr:
                    roofline["mfma_mops_f64"] = ctr["SQ_INSTS_VALU_MFMA_MOPS_F64"]
                if "SQ_INSTS_VALU_MFMA_MOPS_F32" in ctr:
                    roofline["mfma_mops_f32"] = ctr["SQ_INSTS_VALU_MFMA_MOPS_F32"]
                if ctr.get("SQ_VALU_MFMA_BUSY_CYCLES"):      # share of the launch's SIMD-cycles (1024 SIMDs at 2.4 GHz) the matrix pipes work
                    roofline["mfma_busy"] = ctr["SQ_VALU_MFMA_BUSY_CYCLES"] / (k_ms * 1e-3 * 2.4e9 * 1024)
        # ---- CPU baseline: the oracle's full sweep on the host cores, bounded sample
        if args.cpu_walkers > 0 and wl not in ("beta", "c5"):
            from oracle import oracle as O
            onet = O.Net(*w_head)
            ncpu = args.cpu_walkers if wl == "gsvmc" else max(256, args.cpu_walkers // 8)
            O.gsvmc_sweep(64, nup, ndown, onet, args.Z, seed=1)      # thread-pool warm-up
            t1 = time.perf_counter()
            rr = O.gsvmc_sweep(ncpu, nup, ndown, onet, args.Z, seed=2)
            ct = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": ncpu * 100 / ct, "unit": "walker-steps/s", "cores": O.num_threads(),
                                   "kind": "port",
                                   "sample": f"one full iteration (same stages, minus Adam) of {ncpu} walkers, "
                                             f"oracle/ff_oracle.c with OpenMP, {ct:.1f} s", "E": rr["E"], "E_std": rr["E_std"],
                                   "stage_seconds": rr["seconds"]}
            if same is not None:
                # the variance half of BASELINE.json's metric, like for like: the GPU sweep's first walkers x (headline weights) and their
                # E_loc, against the oracle's E_loc of the SAME x at the same tolerances (src/VMC.py:57: mean and unbiased std)
                import numpy as np
                xs, es = same
                t2 = time.perf_counter()
                eo = O.eloc(xs, nup, ndown, onet, args.Z, rtol=1e-6, atol=1e-8)["eloc"]
                out["cpu_baseline"]["same_walkers"] = {
                    "n": int(xs.shape[0]), "what": "the first walkers of one GPU sweep on the headline's weights; the oracle's local energies of the same x "
                                                   "(rtol 1e-6, atol 1e-8, scipy-RK45 restatement)",
                    "E_gpu": float(es.mean()), "E_std_gpu": float(es.std(ddof=1)), "E_cpu": float(eo.mean()), "E_std_cpu": float(eo.std(ddof=1)),
                    "max_rel_eloc_diff": float(np.max(np.abs(es - eo) / np.abs(eo))), "oracle_seconds": time.perf_counter() - t2}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
