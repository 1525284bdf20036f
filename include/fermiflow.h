/* fermiflow.h -- C ABI of libfermiflow_hip.so: the MI355X-native VMC inner loop of FermiFlow.
 *
 * The reference (buwantaiji/FermiFlow) has no FFI layer: its hot path is Python/PyTorch objects
 * (SURVEY.md 8b).  Each entry point below replaces the arithmetic behind one of those Python
 * call sites; the Python classes in fermiflow_amd/ keep the reference signatures and call in here
 * through ctypes with raw device pointers (INTEGRATION.md shows the binding).  Paths are relative
 * to the reference root.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp64 (or int32/uint8 where typed so), contiguous,
 *    row-major; walker coordinates are (B, n, d) exactly as the reference's tensors;
 *  - `stream` is a hipStream_t (0 = default stream); every call only enqueues work, no hidden sync;
 *  - the library allocates no memory: callers pass workspaces where a size query exists.  ff_eloc / ff_eloc_sensitivities
 *    with ff_ode.walker_class set (up to 6 particles, d = 2) run two kernels side by side -- the second on a side stream the
 *    library creates on the first such call on a device (one per device, shared by the process; ff_shutdown releases them),
 *    forked from and joined to `stream` by events inside the call;
 *  - return value: 0 ok, 1 invalid argument, 2 no native instantiation for this configuration,
 *    3 HIP launch failure; ff_last_error() gives a message.  No C++ exception crosses the ABI.
 *  - orbitals are identified by their index k into HO2D().orbitals (src/orbitals.py:81,
 *    k <-> (nx, shell-nx)); an orbital table is int32 [n_states][n_spin]; `walker_state` (int32 [B],
 *    may be NULL = all walkers use row 0) selects the row per walker (BetaVMC, src/VMC.py:89-103).
 */
#ifndef FERMIFLOW_H
#define FERMIFLOW_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* The two scalar MLPs eta, mu of the backflow (src/MLP.py:9-16; Backflow(eta, mu), src/equivariant_funs.py:11-15).
 * w1 = fc1.weight (H,1) flattened, b1 = fc1.bias (H), w2 = fc2.weight (1,H) flattened.  Hm = 0 <=> mu=None. */
typedef struct ff_net {
  int32_t He; const double *ew1, *eb1, *ew2;
  int32_t Hm; const double *mw1, *mb1, *mw2;
  const double* radial_table;   /* optional: built by ff_radial_table_build for THESE weights; NULL = evaluate the
                                   sigmoids directly in every kernel (see fermiflow_amd/csrc/ff_radial.h) */
} ff_net;

/* ODE controls of solve_ivp_nnmodule (src/NeuralODE/nnModule.py:161-162: rtol=1e-6, atol=1e-8). */
typedef struct ff_ode {
  double t0, t1;      /* CNF t_span (src/flow.py:7-40) */
  double rtol, atol;
  int32_t max_steps;  /* safety bound on accepted+rejected steps per walker (0 -> 10000) */
  /* Scheduling aids, both optional (NULL).  Every walker adapts its own step size, so a few walkers (a radius close to
   * zero) need several times the usual number of steps; taking those first keeps them out of the tail of the launch.
   * walker_cost (out, B): a cost class in [0, 32) for integrating along this walker's trajectory again:
   *   min(31, steps attempted in this call + max(0, -log2(r_min^2) - 1)),  r_min = the smallest pair or one-body radius
   *   any stage of the call saw (the backflow field is only C^1 where a radius vanishes, so its second-order
   *   sensitivities -- the local-energy pass -- need many small steps near such points even when this call did not;
   *   ff_cnf_adjoint and the n >= 8 local-energy kernel report the attempted steps only).
   * walker_order (in, B): a permutation of 0..B-1 (ff_walker_order); workgroups take walkers in this order.
   * Results are written to each walker's own slot, so the order changes timing only. */
  int32_t* walker_cost;
  const int32_t* walker_order;
  /* Step-size warm start, optional (NULL).  The cold start (Hairer's rule, as scipy/torchdiffeq) spends one RHS
   * evaluation on a probe and then opens with a step ~20x smaller than the controller settles on, which the x10 growth
   * cap turns into two nearly free steps: 26 evaluations where 14-20 do.  walker_h_init (in, B): first step size to
   * try, times walker_h_scale; entries <= 0 (or NaN) start cold.  walker_h_out (out, B): the largest step size accepted
   * for the walker in this call -- a scale for the next integration along the same trajectory.  The error control of
   * every step is unchanged. */
  const double* walker_h_init;
  double walker_h_scale;
  double* walker_h_out;
  /* Local-energy pass only, optional (walker_class NULL or sens_tol <= 1: off).  The SENSITIVITY components of the integrated
   * state (J = dz/dx, the x-Laplacian of z, the derivatives of Delta) are what needs small steps, and only where a trajectory
   * passes a point at which the field is merely C^1 (a vanishing radius).  walker_class (in, B): a cost class per walker --
   * walker_cost of the flow pass along the same trajectory.  Walkers with class <= sens_tol_class control those components at
   * sens_tol x (rtol, atol) -- their own coordinates keep rtol/atol -- and open with walker_h_init x walker_h_scale_loose
   * (0: walker_h_scale); the others keep one tolerance for everything.  The package's sweeps pass sens_tol = 1 (off: the
   * reference's one tolerance, src/NeuralODE/nnModule.py:161-162) up to 12 coordinates -- rounds 2-4 ran 10 x for class <= 8
   * there and measured 1.3e-5 relative E_loc error on trained flows, past the 1e-5 bar --, sens_tol = 10 with class <= 8 up
   * to 24 coordinates and 5 with class <= 8 beyond (max E_loc error 4.8e-7 / 4.1e-7 on flows trained at those shapes, pinned
   * by tests/test_gpu_parity.py::test_loosened_sensitivity_tolerance_on_shape_trained_flows), and, with ff_walker_schedule's
   * hs_out as walker_h_init, walker_h_scale = walker_h_scale_loose = 1. */
  const int32_t* walker_class;
  double sens_tol;
  double walker_h_scale_loose;
  int32_t sens_tol_class;
  int32_t walker_h_uniform;   /* nonzero: walker_h_init holds ONE entry, the first step size of every walker (a statistic of an
                                 earlier call on other walkers of the same distribution, e.g. the mean of its walker_h_out) */
  /* Local-energy pass with walker_class set, up to 6 particles in d = 2 (0 = the library's default for each).  Walkers of class
   * >= heavy_class (a particle passing the origin: 0.4 % of a batch, 20-30 steps each) are integrated by the one-walker-per-wave
   * kernel, started first, at heavy_tol x (rtol, atol), beside the throughput kernel that takes everyone else -- which kernel
   * integrates a walker depends on its own class only.  Defaults: heavy_class 12 (16 at n d = 12 coordinates, where the pass is bound by
   * its work, not by its longest chain: csrc/ff_cnf_fwd.hip), heavy_tol 0.3; heavy_class < 0: no routing.
   * sum_weight: weight of the two scalar components Delta and lap_x Delta in the error norm of the matrix-core kernel
   * (4-6 particles), which carries each as ONE number where the column and row kernels carry per-lane partial sums (default 4;
   * 1: the plain RMS norm -- fewer steps near a kink, E_loc error up to 3e-6 there). */
  int32_t heavy_class;
  double heavy_tol;
  double sum_weight;
  /* ff_eloc_nd only.  Nonzero: the one-walker-per-workgroup kernels (every shape beyond 12 particles in d = 2 / 8 in d = 3, and the
   * heavy-walker route) finish their walkers in their own epilogue, so that beyond 24 coordinates NO sensitivity goes to HBM and the
   * workspace is the compact one of ff_eloc_nd_workspace_bytes(.., 1): 8 (M + 1) bytes per walker instead of 8 (M^2 + ...) -- 64 MB
   * instead of 3.9 GB at configs[4].  It costs time (the epilogue runs on a workgroup that holds the compute unit alone, where the
   * separate finish kernels fill the chip): local-energy pass + 6 % at configs[4], + 8 % at 12 particles.  0 (the
   * default): those kernels leave the sensitivities in the workspace and the finish kernels run.  The same value must be given
   * to ff_eloc_nd_workspace_bytes. */
  int32_t compact_finish;
  /* ff_cnf_adjoint* only, optional (NULL).  A hipEvent_t the call records on `stream` right behind its dominant kernel -- in front of the
   * small kernels that reduce and contract the parameter gradient.  What waits on it (the sampler of a later iteration, in the
   * package) then runs under those latency-bound kernels instead of beside the adjoint kernel (ABI 106). */
  void* after_main_event;
  /* Flow and adjoint passes with walker_h_init set: nonzero rounds the step a walker opens with, walker_h_init x walker_h_scale, DOWN to
   * t_span / k -- the equal steps that cover the interval in as many steps as it would (ABI 108).  The sweeps open the adjoint with
   * 1.1 x the largest step the walker's flow pass accepted, rounded this way: two steps of 1/2 where 0.42 + 0.58 needed the second
   * step to grow by 1.4 x, which the step-size control grants to half of the walkers only (config 2 after 200 training iterations:
   * 15.9 -> 14.2 evaluations per walker; 12 particles: 13.7 -> 13.1 with 4 % -> 0.06 % of the first steps rejected; configs[4]: 20.7 ->
   * 18.9, 19 % -> 3 %).  The local-energy pass ignores it: ff_walker_schedule rounds its opening steps.  Error control is untouched. */
  int32_t walker_h_equal;
} ff_ode;

int ff_version(void);   /* 109; changes whenever a struct of this header changes layout (the Python binding checks it) */
/* Releases what the library created lazily: the side stream and the two events per device of the routed local-energy pass
 * (created on the first such call on a device, shared by all host threads under a mutex).  Call when no call of this library is
 * in flight; the next routed call creates them again.  Everything else the library touches is caller-owned memory. */
int ff_shutdown(void);
/* The estimator's collectives for a caller that drives this ABI WITHOUT torch: per sweep ONE all-reduce of the
 * buffer ff_energy_estimate (n_global = 0) / ff_beta_state_partials fills, then ONE of the 3(He+Hm)-double gradient of
 * ff_cnf_adjoint* -- thin wrappers of RCCL (ncclCommInitRank / ncclAllReduce(ncclFloat64, ncclSum) / ncclCommDestroy), which is bound at
 * run time (dlopen of the RCCL the process already holds, else librccl.so.1; FF_RCCL_LIB overrides) so that libfermiflow_hip.so itself
 * has no RCCL dependency.  The Python package does not use them: torch.distributed's "nccl" backend is RCCL and owns the communicator
 * there (fermiflow_amd/dist.py); everything else in this library never communicates.
 *   ff_comm_unique_id: rank 0 obtains the 128-byte id and hands it to the other ranks by whatever launched them;
 *   ff_comm_init: collective over the world_size ranks, each on its own current device (one process per GPU);
 *   ff_comm_allreduce: in-place sum of `count` doubles on `stream`;   ff_comm_destroy: NULL is a no-op. */
typedef struct ff_comm ff_comm;
int ff_comm_unique_id(void* id128);
int ff_comm_init(ff_comm** comm, int world_size, int rank, const void* id128);
int ff_comm_allreduce(ff_comm* comm, void* stream, double* buf, int64_t count);
int ff_comm_destroy(ff_comm* comm);
/* order (B) = walker indices sorted by descending cost (ties in a fixed order; classes above 31 count as 31); cost (B) >= 0, e.g. ff_ode.walker_cost. */
size_t ff_walker_order_workspace_bytes(int64_t B);
int ff_walker_order(void* stream, int64_t B, const int32_t* cost, int32_t* order, void* workspace);
/* ff_walker_order that also returns hmean[0] = mean of hval (B) -- the sweeps open the next flow pass with the mean step size this
 * one accepted (ff_ode.walker_h_uniform), and the two launches of the schedule read the per-walker arrays anyway.  Fixed tree. */
int ff_walker_order_mean(void* stream, int64_t B, const int32_t* cost, int32_t* order, void* workspace, const double* hval, double* hmean);
/* ff_walker_order_mean that also chooses the step every walker's local-energy pass opens with (ABI 105; no upstream counterpart -- the
 * reference's solvers start cold, src/NeuralODE/nnModule.py:59-67): hs_out[b] = hval[b] * scale_out[cost[b]], scale a table of 32 factors
 * by cost class (scale_in; entries <= 0 read as 0.6) that FOLLOWS the passes: given the previous pass of the same batch size --
 * prev_cost (its classes), prev_hs (the steps it opened with, i.e. the previous call's hs_out), prev_he (its ff_ode.walker_h_out) --
 * a class of which more than shrink_at (0: 0.10; clamped to [0.02, 0.5]) of the walkers rejected their first step (prev_he < prev_hs) gets
 * 0.93 x its factor -- the caller's choice by how many walkers its local-energy kernel advances in lockstep: a rejection costs the whole
 * wave two more attempts, so the package passes 1 - 0.65^(1/G) within [0.06, 0.25] for G walkers per wave (0.10 at four, 0.25 for the
 * one-walker-per-workgroup kernels; ABI 109) --; one with fewer
 * than half of that gets 1.02 x IF its walkers showed that a plan one step shorter would pass (ABI 107): of those planned for k >= 3 equal steps
 * (at least 16), 70 % accepted a step >= interval / (k - 1) somewhere along the way (prev_he, the largest step the pass accepted;
 * without an interval: every walker votes, with a step >= 1.25 x the one it opened with); within [0.25, 1]; classes with fewer than 64
 * walkers keep theirs.  The updated table -- the one this call applies -- goes to
 * scale_out (a second buffer: workgroups read scale_in while it is written); pass it as scale_in of the next call.  prev_* may be NULL
 * (first call: scale_out = scale_in).  prev_counts (128 doubles, instead of prev_cost / prev_hs / prev_he): the same statistics already
 * counted -- ff_scale_counts of the previous pass, summed over the ranks by the caller (fermiflow_amd adds them to the estimator's
 * all-reduce), so that every rank of a data-parallel run holds the same table.  interval > 0 (= |t1 - t0| of the pass): the step is rounded DOWN to interval / k, the equal steps
 * that cover the interval in as many steps as the scaled one would.  Integer counts and a fixed rule: the table is a deterministic function of the passes before it.  Pass hs_out
 * as ff_ode.walker_h_init with walker_h_scale = walker_h_scale_loose = 1.  The error test of every step is untouched. */
int ff_walker_schedule(void* stream, int64_t B, const int32_t* cost, int32_t* order, void* workspace, const double* hval, double* hmean,
                       const double* scale_in, double* scale_out, const int32_t* prev_cost, const double* prev_hs, const double* prev_he,
                       const double* prev_counts, double interval, double* hs_out, double shrink_at);
/* counts128[c] += walkers of cost class c that opened a local-energy pass with a step (hs > 0) and finished it (he > 0), counts128[32 + c]
 * += those whose first step was rejected (he < hs), counts128[64 + c] += those planned for three or more equal steps of `interval`,
 * counts128[96 + c] += those of them that accepted a step of the plan one shorter (ff_walker_schedule's growth condition; interval as
 * there); the caller zeroes counts128.  Integers held in doubles: exact in any order, and a sum over ranks is the global count. */
int ff_scale_counts(void* stream, int64_t B, const int32_t* cost, const double* hs, const double* he, double interval, double* counts128);
const char* ff_last_error(void);
/* Kernel family of the fused CNF kernels (ff_cnf_generate, ff_cnf_delta_logp, ff_eloc_sensitivities, ff_cnf_adjoint*):
 * 0 (default) = by particle number -- one wave per walker group up to 12 particles in d = 2 / 4 in d = 3, one walker per
 * workgroup beyond (up to 24 particles, n d <= 60); 1 = one walker per workgroup for EVERY particle number (A/B and parity
 * testing; FF_WIDE=1 in the environment selects it at load time).  Returns the previous value. */
int ff_set_kernel_family(int family);
/* Precision of the SENSITIVITY matrices of the local-energy pass in the one-walker-per-workgroup kernels (J = dz/dx with the
 * x-gradient of Delta, A = dv/dz, S = J J^T -- the operands of the two dense products per right-hand side): 64 (default) = fp64 on
 * v_mfma_f64_16x16x4_f64; 32 = fp32 on v_mfma_f32_16x16x4_f32 at twice the rate and half the registers ("fp32 MFMA path" of
 * BASELINE.json configs[4]).  Walker coordinates, radii, the radial functions, Delta, its Laplacian, the x-Laplacian of z and the
 * whole finish stay fp64; E_loc then carries a relative error of ~1e-6 instead of ~1e-9 (printed by the tests).  No effect on the
 * kernels of up to 10 particles.  Returns the previous value. */
int ff_set_sens_precision(int bits);

/* ---- many-body state enumeration (host code, no GPU) ------------------------------------------ */
/* Orbitals.fermion_states (src/orbitals.py:33-54; the subset search of :14-31): all Slater-determinant states of nup
 * spin-up and ndn spin-down fermions in n_orb orbitals with energies orb_E (non-decreasing) whose total energy does
 * not exceed the ground state's by more than deltaE.  The reference implements ndn = 0 only (src/orbitals.py:47-49);
 * for ndn = 0 states and order are the reference's (by energy, ties in lexicographic order of the index tuple);
 * with two species a state is a pair (up subset, down subset) and ties keep the (up, down)-lexicographic order.
 * Returns the number of states (-1: invalid argument).  If it does not exceed `capacity` the states are written:
 * states_up int32 [ns][nup], states_dn int32 [ns][ndn] (orbital indices; either may be NULL), states_E [ns].
 * Call with capacity = 0 to size the buffers. */
int64_t ff_fermion_states(int n_orb, const double* orb_E, int nup, int ndn, double deltaE, int64_t capacity,
                          int32_t* states_up, int32_t* states_dn, double* states_E);

/* ---- Slater determinants / base distribution ------------------------------------------------ */
/* LogAbsSlaterDet.forward (src/slater.py:13-37): logabsdet[b] = log|det phi_j(r_i)|, x (B,n,2). */
int ff_slater_logabsdet_fwd(void* stream, int64_t B, int n, const int32_t* orb_table, const int32_t* walker_state,
                            const double* x, double* logabsdet);
/* LogAbsSlaterDet.backward (src/slater.py:40-62): grad_x[b,i,:] = grad_out[b] * sum_j grad phi_j(r_i) Dinv[j,i]. */
int ff_slater_logabsdet_bwd(void* stream, int64_t B, int n, const int32_t* orb_table, const int32_t* walker_state,
                            const double* x, const double* grad_out, double* grad_x);
/* FreeFermion.log_prob (src/base_dist.py:49-56) = 2*(log|det up| + log|det down|), plus optionally its
 * gradient (B,n,2) and Laplacian (B) wrt x (what y_grad_laplacian, src/utils.py:40-65, extracts). */
int ff_logprob(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
               const int32_t* walker_state, const double* x, double* logp, double* grad, double* lap);

/* ---- MCMC (FreeFermion.sample loop, src/base_dist.py:58-71) -------------------------------- */
/* Parity mode: explicit noise in the reference's draw order: g0 (B,n,2) initial N(0,1) walkers,
 * g (S,B,n,2) proposal noise, u (S,B) uniforms.  accept (S,B) uint8 may be NULL. */
int ff_mcmc_sample_noise(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                         const int32_t* walker_state, int steps, double tau,
                         const double* g0, const double* g, const double* u,
                         double* x_out, double* logp_out, uint8_t* accept);
/* Throughput mode: counter-based Philox4x32-10 + Box-Muller on device; walker b of the global batch
 * uses counter (walker_offset + b), so shards of one batch draw disjoint streams.
 * accept_count (int32 [B], may be NULL) receives the number of accepted proposals. */
int ff_mcmc_sample(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                   const int32_t* walker_state, int steps, double tau, uint64_t seed, int64_t walker_offset,
                   double* x_out, double* logp_out, int32_t* accept_count);
/* Persistent walkers (no upstream counterpart: the reference re-equilibrates from N(0,1) in every iteration,
 * src/base_dist.py:62-64): the same chain, started from x_init (B,n,2) instead of fresh N(0,1) walkers; proposal noise
 * and uniforms are the stream (seed, walker_offset) of ff_mcmc_sample from step 1 on.  x_out may alias x_init. */
int ff_mcmc_continue(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                     const int32_t* walker_state, int steps, double tau, uint64_t seed, int64_t walker_offset,
                     const double* x_init, double* x_out, double* logp_out, int32_t* accept_count);
/* The very noise ff_mcmc_sample consumes, materialised (for tests: feed it to ff_mcmc_sample_noise). */
int ff_rng_fill(void* stream, int64_t B, int n, int steps, uint64_t seed, int64_t walker_offset,
                double* g0, double* g, double* u);

/* ---- backflow / MLP / potentials ------------------------------------------------------------ */
/* MLP.forward / MLP.grad (src/MLP.py:30-45) on N scalars r; dval may be NULL. */
int ff_mlp_eval(void* stream, int64_t N, int H, const double* w1, const double* b1, const double* w2,
                const double* r, double* val, double* dval);
/* Backflow.forward / .divergence (src/equivariant_funs.py:83-102); v (B,n,d) / div (B) may be NULL. */
int ff_backflow_v_div(void* stream, int64_t B, int n, int d, const ff_net* net, const double* x, double* v, double* div);
/* MLP.forward / MLP.grad for any input dimension D_in <= 64 (src/MLP.py:30-45; tests/test_MLP.py builds MLP(15, 40)): x (N, D_in),
 * w1 = fc1.weight (H, D_in) row-major, val (N) and grad (N, D_in) -- either may be NULL. */
int ff_mlp_eval_nd(void* stream, int64_t N, int D_in, int H, const double* w1, const double* b1, const double* w2, const double* x,
                   double* val, double* grad);
/* What autograd needs to differentiate Backflow.forward / .divergence with respect to x (the reference's own check,
 * tests/test_equivariant_funs.py:25-35, takes the divergence of v by autograd): Aw (B,n,d) = (dv/dx)^T w for a cotangent w (B,n,d)
 * and gdiv (B,n,d) = grad_x div v.  (w, Aw) or gdiv may be NULL.  Derivatives with respect to the network parameters are the
 * business of ff_cnf_adjoint*, not of these entry points. */
int ff_backflow_vjp(void* stream, int64_t B, int n, int d, const ff_net* net, const double* x, const double* w, double* Aw, double* gdiv);
/* HO.V + CoulombPairPotential(Z).V (src/potentials.py:13, 23-47); use_ho = 0 drops the trap term. */
int ff_potential(void* stream, int64_t B, int n, int d, double Z, int use_ho, const double* x, double* V);

/* Tabulate eta^(0..8), mu^(0..8) on a uniform grid for the weights in `net` (net->radial_table is ignored);
 * `table` must hold ff_radial_table_bytes() bytes.  Used by ff_cnf_generate / ff_cnf_delta_logp / ff_eloc when
 * net->radial_table points at it: radii on the table are then evaluated by 5th-order Taylor expansion about the
 * nearest node (truncation ~1e-16 relative), all others directly. */
size_t ff_radial_table_bytes(void);
int ff_radial_table_build(void* stream, const ff_net* net, double* table);

/* ---- CNF: fused adaptive Dormand-Prince 5(4) integrations, one walker group per wave ------------ */
/* stats (int32 [4], may be NULL; caller zeroes): [0] += RHS evaluations summed over walkers,
 * [1] = max accepted steps of any walker, [2] += rejected steps, [3] |= failure flags. */
/* CNF.generate (src/flow.py:42-44): x = z + int_{t0}^{t1} v dt. */
int ff_cnf_generate(void* stream, int64_t B, int n, int d, const ff_net* net, const ff_ode* ode,
                    const double* z, double* x_out, int32_t* stats);
/* CNF.delta_logp (src/flow.py:51-55): integrate (x,0) under (v,-div v) from t1 to t0 -> (z, delta). */
int ff_cnf_delta_logp(void* stream, int64_t B, int n, int d, const ff_net* net, const ff_ode* ode,
                      const double* x, double* z_out, double* dlogp_out, int32_t* stats);
/* Backward of delta_logp (SolveIVP.backward + F_augFull, src/NeuralODE/nnModule.py:76-133): given the
 * forward result z(t0) and the incoming gradients a_z (B,n,d), a_d (B), integrates the adjoint system
 * from t0 to t1 and returns grad_x (B,n,d; may be NULL) and the parameter gradient grad_params
 * [3*He + 3*Hm] in Backflow.parameters() order (eta.fc1.weight, eta.fc1.bias, eta.fc2.weight, mu.*).
 * workspace: ff_cnf_adjoint_workspace_bytes(...) bytes of device memory. */
size_t ff_cnf_adjoint_workspace_bytes(int64_t B, int n, int d, int He, int Hm);
int ff_cnf_adjoint(void* stream, int64_t B, int n, int d, const ff_net* net, const ff_ode* ode,
                   const double* z_t0, const double* a_z, const double* a_d,
                   double* grad_x, double* grad_params, void* workspace, int32_t* stats);
/* The same adjoint with the seeds of the energy gradient (src/VMC.py:58-59, gradE = mean(logp (E_loc - E))) formed inside
 * the kernel: with w_b = (eloc[b] - e_mean[k_b]) * scale  (e_mean: DEVICE array, e.g. est of ff_energy_finish; k_b =
 * mean_index[b], or 0 for all walkers if mean_index is NULL; scale = 1 / global batch)  it is
 * ff_cnf_adjoint(a_z = w_b * glogp0[b], a_d = -w_b) -- no (B,n,d) seed array and no host round trip for the mean.
 * mean_index = the walkers' many-body states gives BetaVMC's per-state baseline (src/VMC.py:164-169). */
int ff_cnf_adjoint_energy(void* stream, int64_t B, int n, int d, const ff_net* net, const ff_ode* ode,
                          const double* z_t0, const double* glogp0, const double* eloc, const double* e_mean,
                          const int32_t* mean_index, double scale,
                          double* grad_x, double* grad_params, void* workspace, int32_t* stats);

/* ---- local energy (src/VMC.py:46-55 via src/utils.py:40-65) --------------------------------- */
/* Two launches: (1) a fused per-walker Dormand-Prince pass integrating z, J = dz/dx, the x-Laplacian of z,
 * delta and its x-gradient / x-Laplacian from t1 to t0; (2) a finish kernel contracting them with the
 * Slater gradient / Hessian at z(t0):
 *   logp = logp0(z) - delta,  grad = d logp/dx (B,n,2),  lap = sum_i d2 logp/dx_i^2,
 *   V = Coulomb + trap,  eloc = -lap/4 - |grad|^2/8 + V.
 * Any of logp, grad, lap, V, eloc may be NULL.  Optional extra outputs (may be NULL): z_out (B,n,2) = z(t0),
 * dlogp_out (B) = delta, glogp0_out (B,n,2) = grad_z logp0(z(t0)).
 * workspace: ff_eloc_workspace_bytes(B, n, 2) bytes of device memory (the sensitivities between the launches; more than 12 particles:
 * ff_eloc_nd_workspace_bytes, see ff_eloc_nd).
 * Its head is part of the contract, for callers that want z(t0) and delta without a copy (z_out = dlogp_out = NULL): with
 * M = n d, doubles [0, B M) hold z(t0) (B,n,d) and doubles [B (M^2 + 4 M), B (M^2 + 4 M) + B) hold delta (B) once pass 1 has
 * run; both stay valid until the workspace is reused. */
size_t ff_eloc_workspace_bytes(int64_t B, int n, int d);
int ff_eloc(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
            const int32_t* walker_state, const ff_net* net, const ff_ode* ode, double Z, int use_ho,
            const double* x, double* logp, double* grad, double* lap, double* V, double* eloc,
            double* z_out, double* dlogp_out, double* glogp0_out, void* workspace, int32_t* stats);

/* ff_eloc for d = 2 or 3 (walkers (B, n, d); d = 3: HO3D orbital tables), and the form the sweeps call.  Where the sensitivity
 * kernel implements it -- the matrix-core kernel for nup = ndown <= 3 in d = 2 always; the one-walker-per-workgroup kernels with
 * ode->compact_finish -- the finish is the kernel's own epilogue: logp, grad, lap, V, eloc, glogp0_out are written by the kernel that
 * integrated the walker and J^T never goes to HBM.  workspace: ff_eloc_nd_workspace_bytes(B, n, d, ode->compact_finish) bytes -- the
 * layout of ff_eloc_workspace_bytes, except with compact_finish beyond 24 coordinates (only kernels with that epilogue serve those
 * shapes): the COMPACT layout z(t0) (B, M) | Delta (B) | two counters.  ff_eloc is ff_eloc_nd with d = 2. */
size_t ff_eloc_nd_workspace_bytes(int64_t B, int n, int d, int compact_finish);
int ff_eloc_nd(void* stream, int64_t B, int nup, int ndn, int d, const int32_t* tab_up, const int32_t* tab_dn,
               const int32_t* walker_state, const ff_net* net, const ff_ode* ode, double Z, int use_ho,
               const double* x, double* logp, double* grad, double* lap, double* V, double* eloc,
               double* z_out, double* dlogp_out, double* glogp0_out, void* workspace, int32_t* stats);

/* The two launches of ff_eloc individually (same arguments; results of pass 1 stay in `workspace`; always the full layout). */
int ff_eloc_sensitivities(void* stream, int64_t B, int n, int d, const ff_net* net, const ff_ode* ode,
                          const double* x, void* workspace, int32_t* stats);
int ff_eloc_finish(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                   const int32_t* walker_state, double Z, int use_ho, const double* x, const void* workspace,
                   double* logp, double* grad, double* lap, double* V, double* eloc,
                   double* z_out, double* dlogp_out, double* glogp0_out);

/* ---- estimator reductions (src/VMC.py:57-58) ------------------------------------------------ */
/* out2[0] = sum_b (e[b] - shift), out2[1] = sum_b (e[b] - shift)^2; one workgroup, fixed summation tree
 * (deterministic).  Called with shift = 0 for the mean, then with shift = mean for the unbiased std.
 * shift_dev (device pointer, may be NULL): if given, shift = shift_dev[0] * shift_dev_scale and `shift` is ignored --
 * the mean of the first call (a sum on the device, all-reduced there) feeds the second without a host round trip. */
int ff_reduce_moments(void* stream, int64_t B, const double* e, double shift, const double* shift_dev,
                      double shift_dev_scale, double* out2);
/* GSVMC.forward's estimator (src/VMC.py:56-59) in two small launches around the one all-reduce a multi-GPU run needs:
 * ff_reduce_energy: sums4 = [sum (e - c), sum (e - c)^2, sum logp, sum logp (e - c)] over this rank's B walkers,
 *   c = shift_dev[0] (device; any value all ranks share -- the previous sweep's mean keeps the sums free of cancellation);
 * (the caller adds the sums4 of all ranks;)
 * ff_energy_finish: est3 = [E = mean e, sum (e - E)^2 (E_std^2 = est3[1] / (n - 1)), mean(logp (e - E)) -- the value of
 *   the reference's surrogate gradE] for n_global walkers.  One workgroup, fixed summation tree: deterministic. */
/* Enqueues one idle wave that ends after `microseconds` (0 .. 1e5): the sweep uses it to start the next iteration's
 * ff_mcmc_sample on a side stream a moment AFTER the adjoint kernel of the main stream -- launched the other way round
 * the Metropolis waves fill every SIMD first and the two kernels run one after the other instead of side by side. */
int ff_stream_delay(void* stream, double microseconds);
/* torch.optim.Adam (src/FermionHO2D.py:61, src/BetaFermionHO2D.py) over `ntensors` fp64 device tensors in ONE launch: params, grads,
 * exp_avg, exp_avg_sq are HOST arrays of device pointers, sizes their element counts; `step` is the 1-based count of this update
 * (the caller's: nothing is read back).  The single-tensor formula of torch/optim/adam.py operation for operation, weight_decay as its
 * L2 term; no amsgrad, no maximize.  (PyTorch's fused Adam takes two multi-tensor launches of 24 us for the 300 numbers of the flow.) */
int ff_adam_step(void* stream, int ntensors, const int64_t* sizes, double* const* params, const double* const* grads, double* const* exp_avg,
                 double* const* exp_avg_sq, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step);
int ff_reduce_energy(void* stream, int64_t B, const double* e, const double* logp, const double* shift_dev, double* sums4);
int ff_energy_finish(void* stream, const double* sums4, const double* shift_dev, int64_t n_global, double* est3);
/* The same estimator in ONE launch of many workgroups (ff_reduce_energy is a single workgroup: 19 us at 65 536 walkers): sums4 as
 * ff_reduce_energy; with n_global > 0 -- a single rank, nothing to add from elsewhere -- est3 as ff_energy_finish as well, else
 * (n_global = 0) the caller all-reduces sums4 and calls ff_energy_finish.  Partial sums of fixed segments, added in segment order by
 * whichever workgroup finishes last: deterministic.  workspace: ff_energy_estimate_workspace_bytes(B) bytes that the caller zeroes
 * ONCE before the first call; every call leaves its counter at zero again. */
size_t ff_energy_estimate_workspace_bytes(int64_t B);
int ff_energy_estimate(void* stream, int64_t B, const double* e, const double* logp, const double* shift_dev, int64_t n_global,
                       double* sums4, double* est3, void* workspace);

/* ---- three dimensions (groundwork for a 3-D trap; no upstream counterpart: src/orbitals.py:56 and src/base_dist.py:62
 * hard-code d = 2).  Orbital index k of HO3D: list order "for n in range(8) for nx in range(n+1) for ny in range(n+1-nx):
 * (nx, ny, n-nx-ny)", phi = pi^-3/4 exp(-r^2/2) h_nx(x) h_ny(y) h_nz(z), E = n + 3/2.  Walkers are (B, n, 3). -------------- */
/* FreeFermion.log_prob in d = 3 (src/base_dist.py:49-56 generalised) with, optionally (both or neither), the gradient (B,n,3)
 * and the Laplacian (B) that y_grad_laplacian (src/utils.py:40-65) extracts. */
int ff_logprob3d(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                 const int32_t* walker_state, const double* x, double* logp, double* grad, double* lap);
/* FreeFermion.sample in d = 3 (src/base_dist.py:58-71 generalised): explicit noise g0 (B,n,3), g (S,B,n,3), u (S,B) ... */
int ff_mcmc_sample_noise3d(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                           const int32_t* walker_state, int steps, double tau, const double* g0, const double* g, const double* u,
                           double* x_out, double* logp_out, uint8_t* accept);
/* ... or Philox counters (seed, walker_offset + b), as ff_mcmc_sample. */
int ff_mcmc_sample3d(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                     const int32_t* walker_state, int steps, double tau, uint64_t seed, int64_t walker_offset,
                     double* x_out, double* logp_out, int32_t* accept_count);
/* The very noise ff_mcmc_sample3d consumes, materialised (for tests: feed it to ff_mcmc_sample_noise3d); g0 (B,n,3), g (S,B,n,3), u (S,B). */
int ff_rng_fill3d(void* stream, int64_t B, int n, int steps, uint64_t seed, int64_t walker_offset,
                  double* g0, double* g, double* u);
/* Local-energy finish for d = 3 (pass 1: ff_eloc_sensitivities with d = 3, n = 2..4; workspace: ff_eloc_workspace_bytes(B, n, 3));
 * arguments as ff_eloc_finish, walkers (B, n, 3), HO3D orbital tables. */
int ff_eloc_finish3d(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                     const int32_t* walker_state, double Z, int use_ho, const double* x, const void* workspace,
                     double* logp, double* grad, double* lap, double* V, double* eloc,
                     double* z_out, double* dlogp_out, double* glogp0_out);
/* Backflow.forward / .divergence (src/equivariant_funs.py:83-102) with radii, sigmoid sums and accumulations in fp32
 * (fp64 arrays at the boundary): the single-precision instantiation whose error against ff_backflow_v_div the tests report. */
int ff_backflow_v_div_f32(void* stream, int64_t B, int n, int d, const ff_net* net, const double* x, double* v, double* div);

/* Per-state sums of the finite-temperature estimator (src/VMC.py:155-169): walker_state (int32 [B]) must be SORTED
 * (src/VMC.py:94-96); sums[s] = sum of e[b] over the walkers in state s, counts[s] = their number (as doubles, ready for
 * an all-reduce).  One workgroup per state, fixed summation tree: deterministic. */
int ff_state_sums(void* stream, int64_t B, int nstates, const int32_t* walker_state, const double* e, double* sums, double* counts);
/* BetaVMC.forward's estimator (src/VMC.py:146-171) in two launches around its one all-reduce.  buf holds
 * ff_beta_buffer_doubles(nstates) doubles: [0, 2) the caller's moments of E_loc about shift_dev[0] (ff_reduce_moments),
 * from 2 on the partial per-state sums (sum e, count, sum logp, sum logp e; 16 slices per state) that
 * ff_beta_state_partials writes (walker_state sorted, as for ff_state_sums).  The caller adds the buffers of all ranks;
 * ff_beta_finish then gives est8 = [E, sum (e - E)^2, F, sum (f - F)^2, S, S_analytical, value of gradF_phi, value of
 * gradF_theta], gphi [nstates] = d gradF_phi / d logits, mean_e [nstates] = the per-state baseline of gradF_theta (feed it
 * to ff_cnf_adjoint_energy with mean_index = walker_state) and logp_all [nstates] = log_softmax(logits).
 * One workgroup, fixed summation trees: deterministic. */
size_t ff_beta_buffer_doubles(int nstates);
int ff_beta_state_partials(void* stream, int64_t B, int nstates, const int32_t* walker_state, const double* e, const double* logp,
                           double* buf);
int ff_beta_finish(void* stream, const double* buf, const double* shift_dev, const double* logits, int nstates, double beta,
                   int64_t n_global, double* est8, double* gphi, double* mean_e, double* logp_all);

#ifdef __cplusplus
}
#endif
#endif
