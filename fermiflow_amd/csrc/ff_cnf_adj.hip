// ff_cnf_adj.hip -- fused adjoint of CNF.delta_logp: SolveIVP.backward with F_augFull
// (src/NeuralODE/nnModule.py:76-133), i.e. the theta-gradient pass of `gradE.backward()`
// (src/FermionHO2D.py:71) and the x-gradient of log p.
//
// Augmented system integrated from t0 to t1 per walker (a_D is constant in time):
//     dz/dt      = v(z)
//     da_z/dt    = -(dv/dz)^T a_z + a_D grad_z div v          (dv/dz is symmetric: v is a gradient field)
//     dtheta*/dt = -d/dtheta [ a_z . v(z) - a_D div v(z) ]     (a pure quadrature, summed over walkers)
//
// Mapping: lane (g,i) owns coordinate i of walker g AND the hidden units k = i, i+M, i+2M, ... of eta and mu.
// In the "unit phase" a lane evaluates its own sigmoids at every radius of its walker; per radius the
// partial derivative heads of its units go to LDS (reduced over the group's M lanes afterwards) while the
// parameter-gradient integrands stay in the lane's registers, so the 3*(He+Hm)-wide reduction over
// radii, stages, steps and walkers needs no cross-lane traffic until the kernel's last instruction.
// Rejected steps simply drop the lane-private tentative sums.
#include "ff_common.h"
#include "ff_ode.h"
#include "ff_radial.h"

// FF_STAMPS: diagnostic build only (tools/kbench.py): per-phase s_memtime shares of the tabulated adjoint's RHS loop
#ifdef FF_STAMPS
#define FF_STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += t_ - stamp_prev; stamp_prev = t_; } while (0)
#else
#define FF_STAMP(i) do { } while (0)
#endif

struct ff_adj_args {
  int64_t B;
  ff_net net;
  double ta, tb, rtol, atol;
  int max_steps;
  const double* z_in;   // (B, M)  z(t0)
  const double* az_in;  // (B, M)  incoming gradient wrt z(t0)
  const double* ad_in;  // (B)     incoming gradient wrt Delta
  // energy seeds (ff_cnf_adjoint_energy), optional: with w_b = (w_e[b] - w_mean[0]) * w_scale the seeds are
  // a_z = w_b * az_in[b] and a_Delta = -w_b (ad_in unused) -- the estimator's weights are formed where they are consumed
  const double* w_e;
  const double* w_mean;
  const int32_t* w_index;   // optional (B): which entry of w_mean is walker b's baseline (BetaVMC: its many-body state); NULL: entry 0
  double w_scale;
  double* gx_out;       // (B, M)  gradient wrt x = z(t1); may be NULL
  double* rows;         // (gridDim.x * G, 3He+3Hm) per-(workgroup, group-slot) parameter-gradient partials (direct kernel)
  double* trows;        // (gridDim.x, 2, FF_DEP_NLDS, FF_DEP_ROW) private coefficient tables, then Wtot (tabulated kernel)
  double* off_table;    // one double, zeroed per call: set to 1 when a radius falls off the deposit table
  int32_t* stats;
  const double* h_init;    // optional (B): first step size to try for every walker (ff_ode.walker_h_init), times h_scale
  double h_scale;          // negative: h_init holds ONE entry used by every walker (ff_ode.walker_h_uniform), scale = -h_scale
  int h_equal;             // the opening step rounded down to equal steps of the interval (ff_ode.walker_h_equal)
  double* h_out;           // optional (B): largest step size accepted for every walker in this call (ff_ode.walker_h_out)
  int32_t* wcost;         // optional (B): attempted steps of every walker (ff_ode.walker_cost)
  const int32_t* order;    // optional (B): processing order of the walkers (ff_ode.walker_order)
  int unit0;               // direct kernel: first hidden unit whose parameter gradient THIS launch integrates (chunks of M*MAXU
                           // units; wider nets are served by one launch per chunk -- every launch evaluates all units for the heads)
};

// MAXU: hidden units owned per lane and net (ceil(H/M) for the widths at hand; <= ceil(FF_HMAX/M))
template <int N, int D, int MAXU>
__global__ void __launch_bounds__(FF_WAVE)
ff_ode_adj_kernel(ff_adj_args A) {
  using Gm = ff_geom<N, D>;
  constexpr int M = Gm::M, G = Gm::G, P = Gm::P, R = Gm::RA;
  constexpr int NV = 2;
  {  // a usable radial table means the tabulated kernel (ff_ode_adjtab_kernel) served this call -- unless it met a
     // radius beyond the table and raised off_table, in which case this kernel redoes the call by direct evaluation
    const double* rt = A.net.radial_table;
    if (rt && rt[3] == 0.0 && rt[4] == 0.0 && *A.off_table == 0.0) return;
  }

  __shared__ ff_wtab s_w[2][FF_HPAD];
  __shared__ double s_z[G][M], s_kb[G][M], s_err[G][M], s_ad[G];
  __shared__ double s_rad[G][R], s_rinv[G][R], s_ca[G][R], s_cb[G][R];
  constexpr int CH = 8;   // radii per partial-head reduction chunk (keeps the LDS hand-off buffer small)
  __shared__ double s_ph[G][M][CH][3], s_hd[G][R][3];
  __shared__ int s_pa[R], s_pb[R], s_any;
  __shared__ double s_e2[64];

  __shared__ int s_st[4];   // ODE statistics of this workgroup's walkers (LDS: nothing loop-carried in registers)
#include "ff_adj_direct_body.inc"
}

// LDS of the direct-evaluation kernel as one struct, for the instantiation below that places it in dynamic LDS
template <int N, int D>
struct ff_adj_smem {
  static constexpr int M = ff_geom<N, D>::M, G = ff_geom<N, D>::G, R = ff_geom<N, D>::RA, CH = 8;
  ff_wtab s_w[2][FF_HPAD];
  double s_z[G][M], s_kb[G][M], s_err[G][M], s_ad[G];
  double s_rad[G][R], s_rinv[G][R], s_ca[G][R], s_cb[G][R];
  double s_ph[G][M][CH][3], s_hd[G][R][3];
  double s_e2[64];
  int s_pa[R], s_pb[R], s_any, s_st[4];
};

// The instantiation that is enqueued BEHIND the tabulated kernel as its (normally idle) fallback: compiled for four waves per
// SIMD (<= 128 registers, spilling; its LDS is dynamic so that the compiler does not see an occupancy limit that lets it ignore
// the bound).  An idle fallback has to be SCHEDULED before it can return -- and at 444 registers it could not enter a SIMD while
// the next sweep's Metropolis waves (2 x 164 registers, side stream) were resident: it sat in the queue until they retired, 80-105
// us of every iteration (profiles/r02_f, r03_c; 5 us on an empty GPU).  Calls without a radial table launch the full-speed kernel.
template <int N, int D, int MAXU>
__global__ void __launch_bounds__(FF_WAVE, 4)
ff_ode_adj_lean_kernel(ff_adj_args A) {
  using Gm = ff_geom<N, D>;
  constexpr int M = Gm::M, G = Gm::G, P = Gm::P, R = Gm::RA;
  constexpr int NV = 2;
  {
    const double* rt = A.net.radial_table;
    if (rt && rt[3] == 0.0 && rt[4] == 0.0 && *A.off_table == 0.0) return;
  }
  FF_DYN_LDS(ff_adj_lean_lds);
  ff_adj_smem<N, D>& sm = *reinterpret_cast<ff_adj_smem<N, D>*>(ff_adj_lean_lds);
  auto& s_w = sm.s_w; auto& s_z = sm.s_z; auto& s_kb = sm.s_kb; auto& s_err = sm.s_err; auto& s_ad = sm.s_ad;
  auto& s_rad = sm.s_rad; auto& s_rinv = sm.s_rinv; auto& s_ca = sm.s_ca; auto& s_cb = sm.s_cb;
  constexpr int CH = ff_adj_smem<N, D>::CH;
  auto& s_ph = sm.s_ph; auto& s_hd = sm.s_hd; auto& s_pa = sm.s_pa; auto& s_pb = sm.s_pb; auto& s_any = sm.s_any;
  auto& s_e2 = sm.s_e2; auto& s_st = sm.s_st;
#include "ff_adj_direct_body.inc"
}

// ===================================================================================================
// Tabulated adjoint (used when net.radial_table is valid and the weights are soft enough for the deposit grid).
// Same lane <-> coordinate / lane <-> radius mapping as the forward kernels: the derivative heads come from the radial
// table, and instead of evaluating every hidden unit at every radius for the parameter integrands, each (stage, radius)
// leaves a 4-number record (deposit node, dr, ca, cb).  When a walker's step is accepted its records are added, with
// the Runge-Kutta quadrature weights, into the coefficient table Wacc (ff_radial.h) -- LDS atomics of ONE wave on its
// workgroup-private table, i.e. a fixed order -- and ff_dep_contract_kernel turns the summed table into the
// 3(He+Hm) parameter gradients.  Per RHS evaluation this costs ~50 instructions per radius instead of ~50 sigmoid
// chains per radius.
struct ff_rec { double dr, ca, cb; int j; };
// LDS rows of the deposit table are padded to an odd number of doubles: lanes deposit into rows of different nodes at the
// same column, and a 12-double row stride maps all of them onto four bank groups
#define FF_DEP_LROW (FF_DEP_ROW + 1)

// add one coefficient row to node j of net t: LDS table for the near nodes, global overflow table beyond
// (one branch and one address computation per row, not per coefficient)
// nrow (wave-uniform: header slot 5 of the radial table, ff_radial.h): the coefficients the weights of this launch need -- 6, 8, 10 or all
// 12; an LDS floating-point atomic is served at less than one lane per clock and CU, so the rows are no longer than they have to be
FF_D void ff_row_add(double (*sW)[FF_DEP_NLDS][FF_DEP_LROW], double* __restrict__ ovf, int t, int j, const double* c, int nrow) {
  if (j < FF_DEP_NLDS) {
    double* row = &sW[t][j][0];
#pragma unroll
    for (int k = 0; k < 6; k++) atomicAdd(row + k, c[k]);
#pragma unroll
    for (int k0 = 6; k0 < FF_DEP_ROW; k0 += 2) {
      if (nrow > k0) { atomicAdd(row + k0, c[k0]); atomicAdd(row + k0 + 1, c[k0 + 1]); }
    }
  } else {
    // (hipcc otherwise sinks the two atomic loops into one over a select of an LDS and a global pointer -- a flat pointer
    // whose aperture test it then fails to encode: "Illegal instruction detected" in the one-walker-per-wave kernels)
    double* row = ovf + ((size_t)t * FF_DEP_NTOT + j) * FF_DEP_ROW;
#pragma unroll
    for (int k = 0; k < FF_DEP_ROW; k++) atomicAdd(row + k, c[k]);
    int jo = j;
    FF_OPAQUE(jo);   // an empty asm as this branch's LAST instruction: the two atomic loops stay two
  }
}

FF_D void ff_deposit(double (*sW)[FF_DEP_NLDS][FF_DEP_LROW], double* __restrict__ ovf, int t, const ff_rec& rc, double w, int nrow) {
  if (w == 0.0) return;
  double pk = 1.0, pm = 0.0, c[FF_DEP_ROW];   // dr^k/k!, dr^(k-1)/(k-1)!
#pragma unroll
  for (int k = 0; k < FF_DEP_ROW; k++) {
    c[k] = w * fma(rc.ca, pk, rc.cb * pm);
    pm = pk;
    pk = pk * rc.dr * (1.0 / (k + 1));
  }
  ff_row_add(sW, ovf, t, rc.j, c, nrow);
}

// The five records of one accepted step (stages 0, 2, 3, 4, 5 with the b-weights of the tableau).  Within a step the radius moves a
// little and in one direction: the records are re-expanded about ONE node -- the one nearest to their mean radius -- summed in a
// register accumulator and deposited once (12 LDS atomics per radius and step; rounds 1-4 kept two accumulators for the two
// neighbouring nodes the records fell on: 24 atomics for most radii, 66 M per launch at config 2 -- a third of the kernel, processed at
// less than one lane-operation per cycle and CU).  A record farther than 1.5 node spacings from the centre (a radius that moves by
// more than 0.19 within one step: rare; the branch is skipped when no lane of the wave needs it) goes separately about its own node.
// Truncation: the expansion is of order 11 in w1 dr with |dr| <= 1.5 h_d here, and the table is declared usable only while
// max|w1| h_d <= 0.4 (ff_radial.h: 0.6 while |dr| <= h_d / 2 ... h_d) -- (0.6)^12 / 12! as before.
FF_D void ff_deposit5(double (*sW)[FF_DEP_NLDS][FF_DEP_LROW], double* __restrict__ ovf, int t, const ff_rec& q0, const ff_rec& q2,
                      const ff_rec& q3, const ff_rec& q4, const ff_rec& q5, double hw, int nrow) {
  if (hw == 0.0) return;
  const ff_rec* rc[5] = {&q0, &q2, &q3, &q4, &q5};
  const double bw[5] = {hw * FF_B0, hw * FF_B2, hw * FF_B3, hw * FF_B4, hw * FF_B5};
  constexpr double HD = 1.0 / FF_DEP_INVH;
  double rm = 0.0;
#pragma unroll
  for (int e = 0; e < 5; e++) rm += fma((double)rc[e]->j, HD, rc[e]->dr);
  double jcf = rint(rm * (0.2 * FF_DEP_INVH));
  jcf = fmin(fmax(jcf, 0.0), (double)(FF_DEP_NTOT - 1));
  const int jC = (rm == rm) ? (int)jcf : 0;
  double acc[FF_DEP_ROW];
#pragma unroll
  for (int k = 0; k < FF_DEP_ROW; k++) acc[k] = 0.0;
#pragma unroll
  for (int e = 0; e < 5; e++) {
    const double dr = fma((double)(rc[e]->j - jC), HD, rc[e]->dr);
    if (fabs(dr) <= 1.5 * HD) {
      double pk = bw[e], pm = 0.0;      // w dr^k/k!, w dr^(k-1)/(k-1)!
#pragma unroll
      for (int k = 0; k < FF_DEP_ROW; k++) {
        acc[k] += fma(rc[e]->ca, pk, rc[e]->cb * pm);
        pm = pk;
        pk = pk * dr * (1.0 / (k + 1));
      }
    } else {
      ff_deposit(sW, ovf, t, *rc[e], bw[e], nrow);
    }
  }
  ff_row_add(sW, ovf, t, jC, acc, nrow);
}

#ifndef FF_ADJ_WPS
#define FF_ADJ_WPS 1     // waves per SIMD the tabulated adjoint is compiled for (A/B knob: tools/probes/adj_ab.py)
#endif
#ifndef FF_ADJ_WPW
#define FF_ADJ_WPW 2     // waves per workgroup of the tabulated adjoint (1: the single-wave workgroups of rounds 1-3)
#endif
// Walkers per wave of the tabulated adjoint.  The forward kernels pack 64 / M walkers into a wave; here every lane also
// carries the six stage records of its radii, and at 6 particles (5 walkers x 21 radii = 105 radii: two per lane) that
// put the kernel at 360 registers.  With THREE walkers (63 radii: one per lane) it takes 292 and is exactly as fast
// (0.67 ms per 65 536 walkers either way: the radius phase is what a wave-evaluation waits for) -- and 292 + 164 <= 512:
// a wave of the Metropolis kernel now fits the same SIMD, so the next sweep's walkers are sampled BESIDE the adjoint
// (GSVMC prefetch, DESIGN.md 6: 1.11 -> 0.78 ms for the two together; tools/probes/overlap.py).
#ifndef FF_ADJ_G12
#define FF_ADJ_G12 3
#endif
// (4, 5, 7, 9, 10 and 11 particles likewise -- round 6: with 64 / M walkers their 80-165 radii per wave took two or three record slots
// per lane, 144-556 B of scratch at two waves per SIMD where the allocator had no AGPRs left; as many walkers as keep the radii at one
// per lane -- at least one walker -- instead)
constexpr int ff_adjtab_G(int n, int d) {
  const int M = n * d;
  if (M <= 0 || M > FF_WAVE) return 0;
  const int g = FF_WAVE / M > 16 ? 16 : FF_WAVE / M;
  if (M == 12) return FF_ADJ_G12;
  if (d == 2 && (n == 4 || n == 5 || n == 7 || n == 9 || n == 10 || n == 11)) {      // (measured: 8 and 12 particles are faster at 64 / M)
    const int gr = FF_WAVE / (n * (n + 1) / 2);
    return gr < 1 ? 1 : (gr < g ? gr : g);
  }
  return g;
}
template <int N, int D> struct ff_adjtab_geom { static constexpr int G = ff_adjtab_G(N, D); };
static int adj_tab_G(int n, int d) { return ff_adjtab_G(n, d); }

// WPW waves per workgroup.  A wave at 290 registers and 33 KB of LDS (27 KB of it the deposit table) is alone on its SIMD, and
// every phase of its right-hand side -- radial-table fetch, LDS round trips, dependent fp64 issue -- is exposed: 5 100 cycles per
// evaluation of three walkers (round 4 measurement, s_memtime; without deposits 0.555 ms per launch at one wave per SIMD, 0.404 at
// two).  With WPW = 2 the two waves of a workgroup integrate their OWN walker groups independently -- no workgroup barrier inside
// the loop, only wave-level LDS ordering (FF_WAVE_SYNC) -- and SHARE the workgroup's deposit table: four workgroups per CU are
// then eight waves.  Floating-point LDS atomics of two waves in an arbitrary order would cost the bit-reproducibility of the
// gradient, so deposits take turns: wave v performs its k-th deposit phase when the ticket counter reads 2 k + v (or the other
// wave has finished), i.e. A0 B0 A1 B1 ... -- a fixed order whatever the timing; the right-hand sides in between overlap freely.
template <int N, int D, int WPW>
__global__ void __launch_bounds__(FF_WAVE * WPW, WPW > 1 ? WPW : FF_ADJ_WPS)
ff_ode_adjtab_kernel(ff_adj_args A) {
  using Gm = ff_geom<N, D>;
  constexpr int M = Gm::M, G = ff_adjtab_geom<N, D>::G, P = Gm::P, R = Gm::RA;
  constexpr int NV = 2, NSLOT = (G * R + FF_WAVE - 1) / FF_WAVE;
#ifdef FF_PRIO_ADJTAB
  FF_SETPRIO();
#endif
  const double* __restrict__ rtab = A.net.radial_table;
  if (!(rtab && rtab[3] == 0.0 && rtab[4] == 0.0)) return;   // the direct-evaluation kernel serves this call
  const int dep_nrow = FF_UNIFORM(rtab[5] >= 6.0 && rtab[5] <= (double)FF_DEP_ROW ? (int)rtab[5] : FF_DEP_ROW);   // coefficients per deposit row

  __shared__ double s_z_[WPW][G][M], s_kb_[WPW][G][M], s_err_[WPW][G][M], s_ad_[WPW][G], s_hw_[WPW][G];
  // T[g][a][j][3][D]: what partner j (j = a: the one-body term) contributes to particle a's rows of v, Dv^T[lambda] and
  // grad div -- written by the radius lanes, summed by the component lanes (row padded against bank conflicts)
  constexpr int TROW = N * 3 * D + 1;
  __shared__ double s_T_[WPW][G][N][TROW];
  __shared__ double s_W[2][FF_DEP_NLDS][FF_DEP_LROW];
  __shared__ int s_pa[R], s_pb[R], s_any_[WPW];
  __shared__ int s_turn, s_done[2];

  const int wv = WPW > 1 ? (int)(threadIdx.x >> 6) : 0, lane = threadIdx.x & (FF_WAVE - 1);
  auto& s_z = s_z_[wv]; auto& s_kb = s_kb_[wv]; auto& s_err = s_err_[wv]; auto& s_ad = s_ad_[wv]; auto& s_hw = s_hw_[wv];
  auto& s_T = s_T_[wv]; int& s_any = s_any_[wv];
  const int g = lane / M, i = lane % M;
  const bool ingrp = g < G;
  const int gg = ingrp ? g : 0;
  const int ai = i / D, ci = i % D;
  for (int e = threadIdx.x; e < 2 * FF_DEP_NLDS * FF_DEP_LROW; e += FF_WAVE * WPW) (&s_W[0][0][0])[e] = 0.0;
  if (threadIdx.x == 0) {
    int p = 0;
    for (int a = 0; a < N; a++)
      for (int b = a + 1; b < N; b++) { s_pa[p] = a; s_pb[p] = b; p++; }
    for (int a = 0; a < N; a++) { if (P + a < R) { s_pa[P + a] = a; s_pb[P + a] = -1; } }
    s_turn = 0; s_done[0] = 0; s_done[1] = 0;
  }
  __syncthreads();
  const int He = A.net.He, Hm = A.net.Hm;
  const bool has_mu = Hm > 0;
  const int nrad = has_mu ? (P + N) : P;
  const double tab_inv_h = rtab[0], tab_h = rtab[1];
  const double rtol = A.rtol, atol = A.atol;
  // the radii this lane evaluates (slot sl: radius lane + 64 sl of the wave's G*nrad), fixed for the whole launch:
  // walker slot | particle a << 4 | particle b (15: none) << 8 | radius index << 12
  int rq_id[NSLOT];
#pragma unroll
  for (int sl = 0; sl < NSLOT; sl++) {
    const int q = lane + sl * FF_WAVE;
    const bool act = q < G * nrad;
    const int qg = act ? q / nrad : 0, p = act ? q - qg * nrad : 0;
    rq_id[sl] = act ? (qg | (s_pa[p] << 4) | ((s_pb[p] < 0 ? 15 : s_pb[p]) << 8) | (p << 12)) : -1;
  }
  bool off_any = false;
  constexpr double NT = 2 * M;
  const int64_t ngroups = (A.B + G - 1) / G;
  __shared__ int s_st[4];   // ODE statistics of this workgroup's walkers (LDS: nothing loop-carried in registers)
  if (threadIdx.x < 4) s_st[threadIdx.x] = 0;
  int my_turn = wv;         // ticket of this wave's next deposit phase: wv, wv + 2, ...
  double* const ovf = A.trows + (size_t)gridDim.x * 2 * FF_DEP_NLDS * FF_DEP_ROW;   // Wtot region: [2][NTOT][ROW]
#ifdef FF_STAMPS
  unsigned long long stamp_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev = __builtin_amdgcn_s_memtime();
#endif

  for (int64_t grp = (int64_t)blockIdx.x * WPW + wv; grp < ngroups; grp += (int64_t)gridDim.x * WPW) {
    const int64_t bq = grp * G + g;
    const bool valid = ingrp && bq < A.B;
    const int64_t b = ff_opt_load(A.order, valid, bq, A.z_in, (int32_t)bq);
    double y[NV], k0[NV] = {0.0, 0.0}, k1[NV] = {0.0, 0.0}, k2[NV] = {0.0, 0.0}, k3[NV] = {0.0, 0.0}, k4[NV] = {0.0, 0.0}, k5[NV] = {0.0, 0.0};
    y[0] = ff_opt_load(A.z_in, valid, b * M + i, A.z_in, 0.25 * (i + 1) + 0.125 * ((i * 7) % 5));
    {
      // seeds: given, or formed from the local energies (ff_cnf_adjoint_energy); all loads branch-free (ff_opt_load)
      const bool ws = A.w_e != nullptr;
      const int wi = ff_opt_load(A.w_index, valid && ws, b, A.z_in, (int32_t)0);
      const double wb = (ff_opt_load(A.w_e, valid, b, A.z_in, 0.0) - ff_opt_load(A.w_mean, valid && ws, wi, A.z_in, 0.0)) * A.w_scale;
      const double az0 = ff_opt_load(A.az_in, valid, b * M + i, A.z_in, 0.0), ad0 = ff_opt_load(A.ad_in, valid, b, A.z_in, 0.0);
      y[1] = ws ? wb * az0 : az0;
      if (ingrp && i == 0) s_ad[g] = ws ? -wb : ad0;
    }
    // records of the step under way, per radius slot of this lane: stage 0 (= k0), 2, 3, 4, 5 and 6 (next k0)
    ff_rec r0[NSLOT], r2[NSLOT], r3[NSLOT], r4[NSLOT], r5[NSLOT], r6[NSLOT];
    ff_stepper S;
    S.begin(A.ta, A.tb, valid);
    // warm start (ff_ode.walker_h_init): the step size to try first, instead of the probe evaluation of the Hairer start
    const double hwarm = ff_open_step(ff_opt_load(A.h_init, valid, A.h_scale < 0.0 ? 0 : b, A.z_in, 0.0) * fabs(A.h_scale), A.ta, A.tb, A.h_equal);
    const bool warm = hwarm > 0.0;
    double hmax_acc = 0.0;
    int s = -2, nev = 0;
    double h0v = 0.0, d1v = 0.0;

    auto group_sum = [&](double part) -> double {
      if (ingrp) s_err[g][i] = part;
      FF_WAVE_ORDER();
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < M; j++) t += s_err[gg][j];
      FF_WAVE_ORDER();
      return t;
    };

    // One evaluation and what the step does with it; instantiated per stage (with a run-time stage the step state --
    // y, k0 .. k5 and the six records of every radius -- went through two sets of registers at the end of every evaluation).
    // Returns true when every walker of the wave has finished.
    auto evaluate = [&](auto stage_tag) -> bool {
      constexpr int SG = decltype(stage_tag)::value;
      if constexpr (SG == FF_STAGE_DYN) FF_ASSUME(s <= 0);
      const int sv = SG == FF_STAGE_DYN ? s : SG;
      double in[NV];
      const double h = S.h;
      {
        // y + hsel * sum_k a_k k_k with the stage's tableau row, as a (wave-uniform) switch over literal rows: the row used to come
        // from the constant table FF_ATAB by scalar loads issued where they were needed -- 1 600 of the 5 100 cycles of a
        // wave-evaluation were that wait (s_memtime stamps, profiles/r04_b_adjoint_stamps.json).  Same products in the same
        // order, the zero entries of a row dropped (fma(0, k, x) = x): bit-identical stage inputs.
        const double hsel = (sv == -1) ? h0v * S.dir : h;
#pragma unroll
        for (int v = 0; v < NV; v++) in[v] = y[v];
        switch (sv) {
          case -1:
#pragma unroll
            for (int v = 0; v < NV; v++) in[v] = fma(hsel, k0[v], y[v]);
            break;
          case 1:
#pragma unroll
            for (int v = 0; v < NV; v++) in[v] = fma(hsel * FF_A10, k0[v], y[v]);
            break;
          case 2:
#pragma unroll
            for (int v = 0; v < NV; v++) in[v] = fma(hsel * FF_A21, k1[v], fma(hsel * FF_A20, k0[v], y[v]));
            break;
          case 3:
#pragma unroll
            for (int v = 0; v < NV; v++) in[v] = fma(hsel * FF_A32, k2[v], fma(hsel * FF_A31, k1[v], fma(hsel * FF_A30, k0[v], y[v])));
            break;
          case 4:
#pragma unroll
            for (int v = 0; v < NV; v++)
              in[v] = fma(hsel * FF_A43, k3[v], fma(hsel * FF_A42, k2[v], fma(hsel * FF_A41, k1[v], fma(hsel * FF_A40, k0[v], y[v]))));
            break;
          case 5:
#pragma unroll
            for (int v = 0; v < NV; v++)
              in[v] = fma(hsel * FF_A54, k4[v], fma(hsel * FF_A53, k3[v], fma(hsel * FF_A52, k2[v], fma(hsel * FF_A51, k1[v], fma(hsel * FF_A50, k0[v], y[v])))));
            break;
          case 6:
#pragma unroll
            for (int v = 0; v < NV; v++)
              in[v] = fma(hsel * FF_B5, k5[v], fma(hsel * FF_B4, k4[v], fma(hsel * FF_B3, k3[v], fma(hsel * FF_B2, k2[v], fma(hsel * FF_B0, k0[v], y[v])))));
            break;
          default: break;      // -2, 0: the state itself
        }
      }
      FF_STAMP(0);
      FF_WAVE_ORDER();
      if (ingrp) { s_z[g][i] = in[0]; s_kb[g][i] = in[1]; }
      FF_WAVE_ORDER();
      FF_STAMP(1);
      // ------------------------------------------------------------------ radius phase (lane <-> radius)
      ff_rec cur[NSLOT];
      double rq_rho[NSLOT][D], rq_dl[NSLOT][D], rq_r[NSLOT], rq_ri[NSLOT], rq_T[NSLOT][3 + 5], rq_dr[NSLOT];
      bool rq_ok[NSLOT];
#pragma unroll
      for (int sl = 0; sl < NSLOT; sl++) {   // all radii first, table rows requested ...
        const int id = rq_id[sl];
        const bool act = id >= 0;
        const int qg = act ? (id & 15) : 0, a = act ? ((id >> 4) & 15) : 0, bq = act ? ((id >> 8) & 15) : 15;
        const bool pair = bq != 15;
        const int bb = pair ? bq : a;
        double r2 = 0.0;
#pragma unroll
        for (int c = 0; c < D; c++) {
          rq_rho[sl][c] = s_z[qg][a * D + c] - (pair ? s_z[qg][bb * D + c] : 0.0);
          rq_dl[sl][c] = s_kb[qg][a * D + c] - (pair ? s_kb[qg][bb * D + c] : 0.0);
          r2 = fma(rq_rho[sl][c], rq_rho[sl][c], r2);
        }
        ff_sqrt_rcp(r2, rq_r[sl], rq_ri[sl]);
        rq_dr[sl] = 0.0;
        rq_ok[sl] = ff_table_fetch<3>(rtab, tab_inv_h, tab_h, pair ? 0 : 1, rq_r[sl], rq_T[sl], rq_dr[sl]);
      }
#pragma unroll
      for (int sl = 0; sl < NSLOT; sl++) {   // ... then evaluated
        const int id = rq_id[sl];
        cur[sl].j = 0; cur[sl].dr = 0.0; cur[sl].ca = 0.0; cur[sl].cb = 0.0;
        if (id >= 0) {
          const int qg = id & 15, a = (id >> 4) & 15, bq = (id >> 8) & 15;
          const bool pair = bq != 15;
          const int bb = pair ? bq : a;
          const double* rho = rq_rho[sl];
          const double* dl = rq_dl[sl];
          const double r = rq_r[sl], ri = rq_ri[sl], ad = s_ad[qg];
          double hd[3] = {0.0, 0.0, 0.0}, al = 0.0;
          if (rq_ok[sl]) ff_table_eval<3>(rq_T[sl], rq_dr[sl], hd);
#pragma unroll
          for (int c = 0; c < D; c++) al = fma(dl[c], rho[c], al);
          // parameter integrand  ca * df(r)/dtheta + cb * df'(r)/dtheta, deposited about node jd of the coarse grid
          double jf = rint(r * FF_DEP_INVH);
          // beyond either table (or NaN): the direct kernel redoes the call
          if ((!rq_ok[sl] || !(jf <= (double)(FF_DEP_NTOT - 1))) && (grp * G + qg) < A.B) off_any = true;
          jf = fmin(jf, (double)(FF_DEP_NTOT - 1));
          cur[sl].j = (r == r) ? (int)jf : 0;
          cur[sl].dr = fma(-jf, 1.0 / FF_DEP_INVH, r);
          cur[sl].ca = pair ? -(al - 2.0 * D * ad) : -(al - D * ad);
          cur[sl].cb = pair ? 2.0 * ad * r : ad * r;
          // own rows: +x to particle a from partner bb, -x to particle bb from partner a
          const double f0 = hd[0], f1 = hd[1], f2 = hd[2];
          const double F1 = f1 * (al * ri), gq = (pair ? 2.0 : 1.0) * fma(f2, r, (1.0 + D) * f1) * ri;
          // ONE entry per radius, in the row of its first particle (a < bb): the lanes of particle bb read it there with the other sign
          // (round 6: six LDS stores per pair where the mirrored entry took twelve -- this kernel, too, lives on the LDS)
          double* Ta = &s_T[qg][a][bb * 3 * D];
#pragma unroll
          for (int c = 0; c < D; c++) {
            const double pv = f0 * rho[c], pw = fma(F1, rho[c], f0 * dl[c]), pg = gq * rho[c];
            Ta[c] = pv; Ta[D + c] = pw; Ta[2 * D + c] = pg;
          }
        }
      }
      FF_WAVE_ORDER();
      nev++;
      FF_STAMP(2);
      // ------------------------------------------------------------------ component phase: sum the own rows
      double out[NV];
      {
        double vi = 0.0, dvk = 0.0, gdi = 0.0;
#pragma unroll
        for (int j = 0; j < N; j++) {
          const bool use = has_mu || j != ai;
          const bool lower = j < ai;      // the pair (j, ai) sits in row j, slot ai, written from particle j's side: the other sign
          const double* T = lower ? &s_T[gg][j][ai * 3 * D + ci] : &s_T[gg][ai][j * 3 * D + ci];
          const double sg = lower ? -1.0 : 1.0;
          const double tv = T[0], tw = T[D], tg = T[2 * D];
          vi = fma(sg, use ? tv : 0.0, vi); dvk = fma(sg, use ? tw : 0.0, dvk); gdi = fma(sg, use ? tg : 0.0, gdi);
        }
        out[0] = vi;
        out[1] = fma(s_ad[gg], gdi, -dvk);
      }
      FF_STAMP(3);
      const int s_before = sv;
      // ------------------------------------------------------------------ consume
      if (sv == -2) {
#pragma unroll
        for (int v = 0; v < NV; v++) k0[v] = out[v];
#pragma unroll
        for (int sl = 0; sl < NSLOT; sl++) r0[sl] = cur[sl];
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double isc = ff_rcp(fma(fabs(y[v]), rtol, atol));
          p0 = fma(y[v] * isc, y[v] * isc, p0);
          p1 = fma(k0[v] * isc, k0[v] * isc, p1);
        }
        const double d0 = sqrt(group_sum(p0) * (1.0 / NT));
        d1v = sqrt(group_sum(p1) * (1.0 / NT));
        h0v = S.h0(d0, d1v);
        s = -1;
        // every walker of the wave brings its own first step: no probe evaluation
        if (!ff_wave_or_w(&s_any, lane, (!S.done && !warm) ? 1 : 0)) {
          S.habs = fmin(hwarm, S.interval);
          S.plan();
          s = 1;
        }
      } else if (sv == -1) {
        double p2 = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double t = (out[v] - k0[v]) * ff_rcp(fma(fabs(y[v]), rtol, atol));
          p2 = fma(t, t, p2);
        }
        const double d2 = sqrt(group_sum(p2) * (1.0 / NT)) / h0v;
        S.init_habs(h0v, d1v, d2);
        if (warm) S.habs = fmin(hwarm, S.interval);
        S.plan();
        s = 1;
      } else if (sv == 0) {
#pragma unroll
        for (int v = 0; v < NV; v++) k0[v] = out[v];
#pragma unroll
        for (int sl = 0; sl < NSLOT; sl++) r0[sl] = cur[sl];
        s = 1;
      } else if (sv == 1) {
#pragma unroll
        for (int v = 0; v < NV; v++) k1[v] = out[v];
        s = 2;
      } else if (sv == 2) {
#pragma unroll
        for (int v = 0; v < NV; v++) k2[v] = out[v];
#pragma unroll
        for (int sl = 0; sl < NSLOT; sl++) r2[sl] = cur[sl];
        s = 3;
      } else if (sv == 3) {
#pragma unroll
        for (int v = 0; v < NV; v++) k3[v] = out[v];
#pragma unroll
        for (int sl = 0; sl < NSLOT; sl++) r3[sl] = cur[sl];
        s = 4;
      } else if (sv == 4) {
#pragma unroll
        for (int v = 0; v < NV; v++) k4[v] = out[v];
#pragma unroll
        for (int sl = 0; sl < NSLOT; sl++) r4[sl] = cur[sl];
        s = 5;
      } else if (sv == 5) {
#pragma unroll
        for (int v = 0; v < NV; v++) k5[v] = out[v];
#pragma unroll
        for (int sl = 0; sl < NSLOT; sl++) r5[sl] = cur[sl];
        s = 6;
      } else {
#pragma unroll
        for (int sl = 0; sl < NSLOT; sl++) r6[sl] = cur[sl];
        double pe = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double e = h * (FF_E0 * k0[v] + FF_E2 * k2[v] + FF_E3 * k3[v] + FF_E4 * k4[v] + FF_E5 * k5[v] + FF_E6 * out[v]);
          const double t = e * ff_rcp(fma(fmax(fabs(y[v]), fabs(in[v])), rtol, atol));
          pe = fma(t, t, pe);
        }
        const double err = sqrt(group_sum(pe) * (1.0 / NT));
        const bool was_active = !S.done;
        const bool acc = S.decide(err, A.max_steps);
        if (acc) hmax_acc = fmax(hmax_acc, fabs(h));
        if (acc) {
#pragma unroll
          for (int v = 0; v < NV; v++) { y[v] = in[v]; k0[v] = out[v]; }
        }
        // step size of every walker whose step was accepted (0 otherwise), for the lanes that hold its radii
        if (ingrp && i == 0) s_hw[g] = acc ? h : 0.0;
        FF_WAVE_ORDER();
#ifndef FF_ADJ_NOTICKET      // (timing experiments only: without the turns the sum order, hence the last bits, depend on timing)
        if constexpr (WPW > 1) {      // this wave's turn at the shared table (A0 B0 A1 B1 ...; a finished partner waives its turns)
          while (__hip_atomic_load(&s_turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != my_turn &&
                 !__hip_atomic_load(&s_done[1 - wv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
            __builtin_amdgcn_s_sleep(1);
          FF_WAVE_SYNC();
        }
#endif
#pragma unroll
        for (int sl = 0; sl < NSLOT; sl++) {
          if (rq_id[sl] >= 0) {
            const int qg = rq_id[sl] & 15;
            const int t = ((rq_id[sl] >> 8) & 15) != 15 ? 0 : 1;
            const double hw = s_hw[qg];
#ifndef FF_ADJ_NODEP      // (timing experiments only: the parameter gradient is then wrong)
            ff_deposit5(s_W, ovf, t, r0[sl], r2[sl], r3[sl], r4[sl], r5[sl], hw, dep_nrow);
#endif
            if (hw != 0.0) r0[sl] = r6[sl];   // FSAL: the record of k6 opens that walker's next step
          }
        }
        if constexpr (WPW > 1) {      // deposits landed (lgkmcnt(0)): hand the table on
          FF_WAVE_SYNC();
          if (lane == 0) __hip_atomic_store(&s_turn, my_turn + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          my_turn += 2;
        }
        S.plan();
        const int any = ff_wave_or_w(&s_any, lane, S.done ? 0 : ((was_active && !acc) ? 3 : 1));
        s = (any & 2) ? 0 : 1;
        if (!any) { FF_STAMP(5); return true; }
      }
      if (s_before == 6) FF_STAMP(5); else FF_STAMP(4);
      return false;
    };
#pragma unroll 1
    for (;;) {
#pragma unroll 1
      while (s <= 0) evaluate(ff_stage_c<FF_STAGE_DYN>{});
      evaluate(ff_stage_c<1>{});
      evaluate(ff_stage_c<2>{});
      evaluate(ff_stage_c<3>{});
      evaluate(ff_stage_c<4>{});
      evaluate(ff_stage_c<5>{});
      if (evaluate(ff_stage_c<6>{})) break;
    }
    if (valid) {
      const double bad = S.fail ? __builtin_nan("") : 0.0;   // failed integration -> NaN gradients (see ff_ode_adj_kernel)
      if (A.gx_out) A.gx_out[b * M + i] = y[1] + bad;
      if (i == 0) {
        if (S.fail) atomicAdd(&s_W[0][0][0], bad);
        if (A.h_out) A.h_out[b] = hmax_acc > 0.0 ? hmax_acc : hwarm;
        if (A.wcost) A.wcost[b] = S.nacc + S.nrej;
        if (A.stats) { atomicAdd(&s_st[0], nev); atomicMax(&s_st[1], S.nacc); atomicAdd(&s_st[2], S.nrej); if (S.fail) atomicMax(&s_st[3], 1); }
      }
    }
    FF_WAVE_ORDER();
  }
#ifdef FF_STAMPS
  if (A.stats && lane == 0)
    for (int q = 0; q < 9; q++) atomicAdd((unsigned long long*)(A.stats + 8) + q, stamp_acc[q]);
#endif
  if (off_any) *A.off_table = 1.0;
  if constexpr (WPW > 1) {
    FF_WAVE_SYNC();
    if (lane == 0) __hip_atomic_store(&s_done[wv], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();      // (the one workgroup barrier: every wave has made its last deposit)
  // flush the workgroup-private coefficient table
  {
    double* row = A.trows + (size_t)blockIdx.x * 2 * FF_DEP_NLDS * FF_DEP_ROW;
    for (int e = threadIdx.x; e < 2 * FF_DEP_NLDS * FF_DEP_ROW; e += FF_WAVE * WPW) row[e] = (&s_W[0][0][0])[(e / FF_DEP_ROW) * FF_DEP_LROW + e % FF_DEP_ROW];
  }
  __syncthreads();
  if (A.stats && threadIdx.x == 0 && (s_st[0] || s_st[3])) {
    atomicAdd(&A.stats[0], s_st[0]);
    atomicMax(&A.stats[1], s_st[1]);
    atomicAdd(&A.stats[2], s_st[2]);
    if (s_st[3]) atomicMax(&A.stats[3], 1);
  }
}

extern bool ff_wide_forced();                  // ff_wide.hip
extern int ff_wide_supported(int n, int d);
#include "ff_adj_wide.h"   // one walker per wave, run-time particle number (n > 12 in d = 2, n > 4 in d = 3)

// Wtot[t][j < NLDS][k] = sum over workgroups of their private tables; j >= NLDS was added in place.
#ifndef FF_DEPR_EX
#define FF_DEPR_EX 16      // entries per workgroup of the deposit-table part of ff_adj_reduce_kernel (consecutive: one 128-byte segment per table)
#define FF_DEPR_TY 16      // table subsets summed side by side
#endif
// ONE launch behind the adjoint kernels for both cross-workgroup sums (they were two; exactly one of them has work): workgroups
// [0, P) sum the direct kernel's per-workgroup parameter rows (ff_rows_reduce: only if the direct kernel ran), the others the
// tabulated kernel's private deposit tables (only if the tabulated kernel served the call)
__global__ void __launch_bounds__(FF_DEPR_EX * FF_DEPR_TY)
ff_adj_reduce_kernel(ff_net net, const double* __restrict__ off_table, int nrows, int P, const double* __restrict__ prow,
                     double* __restrict__ out, int nblocks, const double* __restrict__ rows, double* __restrict__ wtot) {
  FF_SETPRIO();
  const double* rtab = net.radial_table;
  const bool tab_served = rtab && rtab[3] == 0.0 && rtab[4] == 0.0 && *off_table == 0.0;
  if ((int)blockIdx.x < P) {
    if (tab_served) return;
    __shared__ double smr[FF_DEPR_EX * FF_DEPR_TY];
    const int k = blockIdx.x;
    double sr = 0.0;
    for (int r = threadIdx.x; r < nrows; r += blockDim.x) sr += prow[(int64_t)r * P + k];
    smr[threadIdx.x] = sr;
    __syncthreads();
    for (int w = blockDim.x / 2; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) smr[threadIdx.x] += smr[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = smr[0];
    return;
  }
  if (!tab_served || rows == nullptr) return;
  const int bid = (int)blockIdx.x - P;
  // thread (tx, ty): entry e0 + tx of the tables ty, ty + TY, ... -- the 16 lanes of a row read one contiguous segment of a
  // table (the first version gave every entry its own workgroup whose lanes strode over the tables, 24.6 KB apart: 38 us for
  // 25 MB); partial sums meet in a fixed tree, so the result depends on (nblocks, TY) only: deterministic
  __shared__ double sm[FF_DEPR_TY][FF_DEPR_EX + 1];
  const int tx = threadIdx.x % FF_DEPR_EX, ty = threadIdx.x / FF_DEPR_EX;
  const int e = bid * FF_DEPR_EX + tx;
  constexpr int NE = 2 * FF_DEP_NLDS * FF_DEP_ROW;
  // eight independent partial sums: eight loads in flight per thread.  (Thirty-two were worse, 64 us against 30: the kernel runs beside
  // the prefetched Metropolis kernel, which owns most of the issue slots -- what decides its time is how soon its waves get placed.)
  double sp[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (e < NE) {
    int b = ty;
    for (; b + 7 * FF_DEPR_TY < nblocks; b += 8 * FF_DEPR_TY) {
#pragma unroll
      for (int u = 0; u < 8; u++) sp[u] += rows[(size_t)(b + u * FF_DEPR_TY) * NE + e];
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (b + u * FF_DEPR_TY < nblocks) sp[u] += rows[(size_t)(b + u * FF_DEPR_TY) * NE + e];
  }
  sm[ty][tx] = ((sp[0] + sp[1]) + (sp[2] + sp[3])) + ((sp[4] + sp[5]) + (sp[6] + sp[7]));
  __syncthreads();
  for (int w = FF_DEPR_TY / 2; w > 0; w >>= 1) {
    if (ty < w) sm[ty][tx] += sm[ty + w][tx];
    __syncthreads();
  }
  if (ty == 0 && e < NE) {
    const int t = e / (FF_DEP_NLDS * FF_DEP_ROW), rem = e - t * FF_DEP_NLDS * FF_DEP_ROW;
    wtot[(size_t)t * FF_DEP_NTOT * FF_DEP_ROW + rem] = sm[0][tx];
  }
}

// grad[theta] = sum_{j,k} Wtot[t][j][k] dT[t][j][k]/dtheta,  T[j][k] = sum_h w2 w1^k sigma^(k)(w1 r_j + b1);
// one workgroup per hidden unit, lanes over the nodes, fixed-tree reduction.
// Written as COMPACT LOOPS over tables staged in LDS (the net's half of Wtot: 48 KB in one coalesced pass; the 182 polynomial
// coefficients of the sigmoid's derivatives).  The first version was 900 instructions of straight-line code per thread -- every
// coefficient a literal or a scalar load, Wtot's rows fetched where they were used -- and took 32-45 us for 2 nodes per thread: a kernel
// that runs once per launch pays the memory latency of every instruction line and of every dependent load it touches.
__constant__ double FF_SIGPOLY_MEM[13][14] = FF_SIGPOLY_INIT;
__global__ void __launch_bounds__(256)
ff_dep_contract_kernel(ff_net net, const double* __restrict__ off_table, const double* __restrict__ wtot, double* __restrict__ grad) {
  FF_SETPRIO();
  const double* rtab = net.radial_table;
  if (!(rtab && rtab[3] == 0.0 && rtab[4] == 0.0 && *off_table == 0.0)) return;
  __shared__ double sm[3][256];
  __shared__ double s_W[FF_DEP_NTOT * FF_DEP_ROW];
  __shared__ double s_P[13 * 14];
  const int t = blockIdx.x < (unsigned)net.He ? 0 : 1, hu = t ? blockIdx.x - net.He : blockIdx.x;
  const int H = t ? net.Hm : net.He;
  const double w1 = (t ? net.mw1 : net.ew1)[hu], b1 = (t ? net.mb1 : net.eb1)[hu], w2 = (t ? net.mw2 : net.ew2)[hu];
  {
    const double* Wt = wtot + (size_t)t * FF_DEP_NTOT * FF_DEP_ROW;
    for (int e = threadIdx.x; e < FF_DEP_NTOT * FF_DEP_ROW; e += blockDim.x) s_W[e] = Wt[e];
    for (int e = threadIdx.x; e < 13 * 14; e += blockDim.x) s_P[e] = (&FF_SIGPOLY_MEM[0][0])[e];
  }
  __syncthreads();
  double gw1 = 0.0, gb1 = 0.0, gw2 = 0.0;
  for (int j = threadIdx.x; j < FF_DEP_NTOT; j += blockDim.x) {
    const double rj = (double)j * (1.0 / FF_DEP_INVH);
    const double sg = ff_sigmoid(fma(w1, rj, b1));
    const double* W = &s_W[j * FF_DEP_ROW];
    {      // a node nothing was deposited on (every node beyond the largest radius of the batch: three quarters of the table) contributes nothing
      bool any = false;
#pragma unroll
      for (int k = 0; k < FF_DEP_ROW; k++) any = any || (W[k] != 0.0);
      if (!any) continue;
    }
    double wk = 1.0, wkm = 0.0, prev = 0.0;   // w1^k, k w1^(k-1), sigma^(k)
#pragma unroll 1
    for (int n = 0; n <= FF_DEP_ROW; n++) {
      double p = s_P[n * 14 + n + 1];
#pragma unroll 1
      for (int c = n; c >= 1; c--) p = fma(p, sg, s_P[n * 14 + c]);
      const double cur = p * sg;                // sigma^(n)
      if (n >= 1) {                             // term k = n - 1: needs sigma^(k) = prev and sigma^(k+1) = cur
        const double Wk = W[n - 1];
        gw2 = fma(Wk, wk * prev, gw2);
        gb1 = fma(Wk, w2 * wk * cur, gb1);
        gw1 = fma(Wk, w2 * fma(wkm, prev, wk * rj * cur), gw1);
        wkm = (double)n * wk;
        wk *= w1;
      }
      prev = cur;
    }
  }
  sm[0][threadIdx.x] = gw1; sm[1][threadIdx.x] = gb1; sm[2][threadIdx.x] = gw2;
  __syncthreads();
  for (int w = blockDim.x / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
      for (int c = 0; c < 3; c++) sm[c][threadIdx.x] += sm[c][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* gout = grad + (t ? 3 * net.He : 0);
    gout[hu] = sm[0][0]; gout[H + hu] = sm[1][0]; gout[2 * H + hu] = sm[2][0];
  }
}

__global__ void __launch_bounds__(256) ff_zero_kernel(double* __restrict__ p, size_t n) {
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) p[k] = 0.0;
}

// =================================================================================================
extern void ff_set_error(const char* msg);
#define FF_CHECK(cond, code, msg) do { if (!(cond)) { ff_set_error(msg); return code; } } while (0)
#define FF_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { ff_set_error(hipGetErrorString(e_)); return FF_ELAUNCH; } } while (0)

#include <stdlib.h>

static int adj_G(int n, int d) { int M = n * d; return M > 0 && M <= FF_WAVE ? (FF_WAVE / M > 16 ? 16 : FF_WAVE / M) : 0; }   // = ff_geom<N,D>::G
// Persistent grid: one wave per SIMD.  Every walker takes the same few steps here, so a static split is balanced, and
// each workgroup flushes a private deposit table (25 KB) at its end -- the fewer workgroups the less HBM traffic
// (measured, 65536 walkers: 4096 workgroups 1.09 ms, 1024 workgroups 0.92 ms).
static int64_t adj_default_blocks() {
  static int64_t n = 0;
  if (n == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    n = 4 * (int64_t)cus;
  }
  return n;
}

static unsigned adj_grid(int64_t B, int G) {
  int64_t ngroups = (B + G - 1) / G;
  const int64_t cap = adj_default_blocks();
  return (unsigned)(ngroups < cap ? ngroups : cap);
}

// Direct-evaluation adjoint: a lane keeps the parameter integrands of MAXU of its hidden units in registers, so one launch
// integrates the gradient of M*MAXU units (all of the reference's default width 50 at once); wider nets (--Deta/--Dmu up to
// FF_HMAX, src/FermionHO2D.py:24-27) take one launch per chunk of units.
template <int N, int D>
static void launch_adj(void* stream, const ff_adj_args& a_in) {
  constexpr int M = ff_geom<N, D>::M;
  constexpr int MU_64 = (64 + M - 1) / M > 16 ? 16 : (64 + M - 1) / M, MU_50 = (50 + M - 1) / M > 16 ? 16 : (50 + M - 1) / M;
  const int hmax = a_in.net.He > a_in.net.Hm ? a_in.net.He : a_in.net.Hm;
  ff_adj_args a = a_in;
  const bool lean = a_in.net.radial_table != nullptr;      // behind the tabulated kernel: the small-footprint fallback
  const size_t lds = sizeof(ff_adj_smem<N, D>);
  if (MU_50 < MU_64 && hmax <= MU_50 * M) {
    a.unit0 = 0;
    if (lean) FF_LAUNCH_LDS((ff_ode_adj_lean_kernel<N, D, MU_50>), adj_grid(a.B, ff_geom<N, D>::G), FF_WAVE, lds, stream, a);
    else FF_LAUNCH((ff_ode_adj_kernel<N, D, MU_50>), adj_grid(a.B, ff_geom<N, D>::G), FF_WAVE, stream, a);
  } else {
    for (int u0 = 0; u0 < hmax; u0 += MU_64 * M) {
      a.unit0 = u0;
      if (lean) FF_LAUNCH_LDS((ff_ode_adj_lean_kernel<N, D, MU_64>), adj_grid(a.B, ff_geom<N, D>::G), FF_WAVE, lds, stream, a);
      else FF_LAUNCH((ff_ode_adj_kernel<N, D, MU_64>), adj_grid(a.B, ff_geom<N, D>::G), FF_WAVE, stream, a);
    }
  }
}

// ff_ode.walker_h_equal for the one-walker-per-workgroup adjoint kernels: their opening steps are rounded HERE, in a launch of its own
// (5 us in front of 1-30 ms), not in their walker prologue as everywhere else.  With ff_open_step in that prologue -- and the flag zero --
// ff_wide_adjtab_kernel<3, 4, 1> failed every walker of a 20-particle batch on the GPU (not in the host simulator; no scratch, four
// more VGPRs; the kernel spills 114 scalar registers into vector lanes and the change moved that; tools/check_agpr_spills.py finds
// nothing): the kernels stay byte for byte what the parity tests pinned.
__global__ void __launch_bounds__(256) ff_open_steps_kernel(int64_t B, const double* __restrict__ h_init, double scale, double ta, double tb,
                                                            double* __restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) out[b] = ff_open_step(h_init[b] * scale, ta, tb, 1);
}

extern "C" {

static size_t adj_table_doubles(int64_t B, int Gtab) {   // one private table per workgroup of the tabulated kernel + Wtot
  return (size_t)adj_grid(B, Gtab) * 2 * FF_DEP_NLDS * FF_DEP_ROW + (size_t)2 * FF_DEP_NTOT * FF_DEP_ROW;
}

// workspace = [direct rows | private tables + Wtot | off-table flag]
static size_t adj_direct_doubles(int64_t B, int G, int He, int Hm) { return (size_t)adj_grid(B, G) * G * (size_t)(3 * He + 3 * Hm); }

// the narrow kernels are instantiated for n = 1..12 in d = 2 and n = 2..4 in d = 3; everything else (and everything under
// FF_WIDE=1) goes to the one-walker-per-wave kernels of ff_adj_wide.h
static bool adj_is_wide(int n, int d) {
  const bool narrow = (d == 2 && n >= 1 && n <= 12) || (d == 3 && n >= 2 && n <= 4);
  return ff_wide_supported(n, d) && (!narrow || ff_wide_forced());
}

// doubles of the layout one kernel family uses (0: that family does not serve (n, d))
static size_t adj_ws_doubles(bool wide, int64_t B, int n, int d, int He, int Hm) {
  if (wide) return ff_wide_supported(n, d) ? adj_direct_doubles(B, 1, He, Hm) + adj_table_doubles(B, 1) + 1 + (size_t)B : 0;   // (+ B: opening steps, ff_ode.walker_h_equal)
  const bool narrow = (d == 2 && n >= 1 && n <= 12) || (d == 3 && n >= 2 && n <= 4);
  const int G = adj_G(n, d);
  return (narrow && G) ? adj_direct_doubles(B, G, He, Hm) + adj_table_doubles(B, adj_tab_G(n, d) * FF_ADJ_WPW) + 1 : 0;
}

// The larger of the two families' layouts: which family a call uses is decided when it runs (ff_set_kernel_family / FF_WIDE may
// change between this query and the call, e.g. from another thread) -- a buffer of this size serves either.
size_t ff_cnf_adjoint_workspace_bytes(int64_t B, int n, int d, int He, int Hm) {
  if (B <= 0) return 0;
  const size_t a = adj_ws_doubles(false, B, n, d, He, Hm), b = adj_ws_doubles(true, B, n, d, He, Hm);
  return sizeof(double) * (a > b ? a : b);
}

static int adjoint_impl(void* stream, int64_t B, int n, int d, const ff_net* net, const ff_ode* ode, const double* z_t0,
                        const double* a_z, const double* a_d, const double* w_e, const double* w_mean, const int32_t* w_index,
                        double w_scale, double* grad_x, double* grad_params, void* workspace, int32_t* stats);

int ff_cnf_adjoint(void* stream, int64_t B, int n, int d, const ff_net* net, const ff_ode* ode, const double* z_t0,
                   const double* a_z, const double* a_d, double* grad_x, double* grad_params, void* workspace,
                   int32_t* stats) {
  FF_CHECK(B == 0 || a_d, FF_EINVAL, "ff_cnf_adjoint: null pointer");
  return adjoint_impl(stream, B, n, d, net, ode, z_t0, a_z, a_d, nullptr, nullptr, nullptr, 0.0, grad_x, grad_params, workspace, stats);
}

int ff_cnf_adjoint_energy(void* stream, int64_t B, int n, int d, const ff_net* net, const ff_ode* ode, const double* z_t0,
                          const double* glogp0, const double* eloc, const double* e_mean, const int32_t* mean_index, double scale,
                          double* grad_x, double* grad_params, void* workspace, int32_t* stats) {
  FF_CHECK(B == 0 || (eloc && e_mean), FF_EINVAL, "ff_cnf_adjoint_energy: null pointer");
  return adjoint_impl(stream, B, n, d, net, ode, z_t0, glogp0, nullptr, eloc, e_mean, mean_index, scale, grad_x, grad_params, workspace,
                      stats);
}

static int adjoint_impl(void* stream, int64_t B, int n, int d, const ff_net* net, const ff_ode* ode, const double* z_t0,
                        const double* a_z, const double* a_d, const double* w_e, const double* w_mean, const int32_t* w_index,
                        double w_scale, double* grad_x, double* grad_params, void* workspace, int32_t* stats) {
  FF_CHECK(B >= 0 && n > 0 && d > 0 && net && ode && grad_params, FF_EINVAL, "ff_cnf_adjoint: bad argument");
  FF_CHECK(net->He > 0 && net->ew1 && net->eb1 && net->ew2 && (net->Hm == 0 || (net->mw1 && net->mb1 && net->mw2)), FF_EINVAL,
           "ff_cnf_adjoint: bad net");
  FF_CHECK(net->He <= FF_HMAX && net->Hm <= FF_HMAX, FF_EUNSUPPORTED, "ff_cnf_adjoint: hidden width > 256");
  FF_CHECK(ode->rtol > 0 && ode->atol > 0, FF_EINVAL, "ff_cnf_adjoint: tolerances must be positive");
  const int P = 3 * net->He + 3 * net->Hm;
  if (B == 0) {
    if (hipMemsetAsync(grad_params, 0, sizeof(double) * P, (hipStream_t)stream) != hipSuccess) return FF_ELAUNCH;
    return FF_OK;
  }
  FF_CHECK(z_t0 && a_z && workspace, FF_EINVAL, "ff_cnf_adjoint: null pointer");
  ff_adj_args a = {};
  a.B = B; a.net = *net; a.ta = ode->t0; a.tb = ode->t1; a.rtol = ode->rtol; a.atol = ode->atol;
  a.max_steps = ode->max_steps > 0 ? ode->max_steps : 10000;
  a.wcost = ode->walker_cost; a.order = ode->walker_order;
  a.h_init = ode->walker_h_init; a.h_scale = ode->walker_h_uniform ? -fabs(ode->walker_h_scale) : fabs(ode->walker_h_scale); a.h_out = ode->walker_h_out; a.h_equal = ode->walker_h_equal;
  a.z_in = z_t0; a.az_in = a_z; a.ad_in = a_d; a.w_e = w_e; a.w_mean = w_mean; a.w_index = w_index; a.w_scale = w_scale; a.gx_out = grad_x; a.rows = (double*)workspace; a.stats = stats;
  const bool wide = adj_is_wide(n, d);      // the family of THIS call, read once: layout, memset and launches below all follow it
  {
    const int Gq = wide ? 1 : adj_G(n, d);
    if (Gq == 0) { ff_set_error("ff_cnf_adjoint: n*d > 64"); return FF_EUNSUPPORTED; }
    a.trows = a.rows + adj_direct_doubles(B, Gq, net->He, net->Hm);
    a.off_table = a.trows + adj_table_doubles(B, wide ? 1 : adj_tab_G(n, d) * FF_ADJ_WPW);
  }
  if (wide) {      // (those kernels accumulate in their global rows and tables)
    if (hipMemsetAsync(workspace, 0, sizeof(double) * adj_ws_doubles(wide, B, n, d, net->He, net->Hm), (hipStream_t)stream) != hipSuccess) return FF_ELAUNCH;
  } else {
    // The narrow kernels write their private rows and tables in full; what must start at zero is the shared overflow table (Wtot:
    // global atomics for the nodes beyond the LDS tables) and the off-table flag behind it -- 98 KB by a kernel of our own: the
    // 37 MB hipMemsetAsync this replaces took 8.7 us plus the 8 us of pipeline bubble every blit costs on this GPU.
    double* wt = a.off_table - (size_t)2 * FF_DEP_NTOT * FF_DEP_ROW;
    FF_LAUNCH(ff_zero_kernel, 48, 256, stream, wt, (size_t)2 * FF_DEP_NTOT * FF_DEP_ROW + 1);
  }
  int G = 0;
  if (wide && a.h_equal && a.h_init) {
    const int64_t nh = a.h_scale < 0.0 ? 1 : B;      // (walker_h_uniform: one entry)
    double* hs = a.off_table + 1;
    FF_LAUNCH(ff_open_steps_kernel, (unsigned)((nh + 255) / 256), 256, stream, nh, a.h_init, fabs(a.h_scale), a.ta, a.tb, hs);
    a.h_init = hs; a.h_scale = a.h_scale < 0.0 ? -1.0 : 1.0; a.h_equal = 0;
  }
  if (wide) {
    // lanes per walker: two waves up to 128 radii (pairs + one-body), four beyond; one radius per lane up to 22 particles
    const int nr = n * (n + 1) / 2, Wv = nr <= 128 ? 2 : 4, nq = (nr + 64 * Wv - 1) / (64 * Wv);
    const unsigned grid = adj_grid(a.B, 1);
#define FF_WA(D_, W_, Q_) if (d == D_ && Wv == W_ && nq == Q_) { if (net->radial_table) FF_LAUNCH((ff_wide_adjtab_kernel<D_, W_, Q_>), grid, FF_WAVE * W_, stream, a, n); \
                                                                  FF_LAUNCH((ff_wide_adj_kernel<D_, W_, Q_>), grid, FF_WAVE * W_, stream, a, n); G = 1; }
    FF_WA(2, 2, 1) FF_WA(2, 4, 1) FF_WA(2, 4, 2) FF_WA(3, 2, 1) FF_WA(3, 4, 1) FF_WA(3, 4, 2)
#undef FF_WA
  } else
  // both variants are enqueued; on the device exactly one of them runs, chosen by the radial-table header
  // (no table / weights too stiff for the deposit grid -> direct evaluation), so the host never has to look at it
#define FF_ND(N_, D_) if (n == N_ && d == D_) { if (net->radial_table) FF_LAUNCH((ff_ode_adjtab_kernel<N_, D_, FF_ADJ_WPW>), adj_grid(a.B, ff_adjtab_geom<N_, D_>::G * FF_ADJ_WPW), FF_WAVE * FF_ADJ_WPW, stream, a); launch_adj<N_, D_>(stream, a); G = ff_geom<N_, D_>::G; }
  FF_ND(6, 2) else FF_ND(3, 2) else FF_ND(12, 2) else FF_ND(2, 2) else FF_ND(4, 2) else FF_ND(5, 2) else FF_ND(8, 2) else FF_ND(10, 2)
  else FF_ND(1, 2) else FF_ND(7, 2) else FF_ND(9, 2) else FF_ND(11, 2) else FF_ND(2, 3) else FF_ND(3, 3) else FF_ND(4, 3)
#undef FF_ND
  if (G == 0) {
    ff_set_error("fused CNF kernels serve n <= 24 particles with n*d <= 60 in d = 2, 3");
    return FF_EUNSUPPORTED;
  }
  FF_LAUNCH_CHECK();
  // ff_ode.after_main_event: whoever waits for it runs under the small kernels below, not beside the one above
  if (ode->after_main_event && hipEventRecord((hipEvent_t)ode->after_main_event, (hipStream_t)stream) != hipSuccess) {
    ff_set_error("ff_cnf_adjoint: hipEventRecord(after_main_event) failed");
    return FF_ELAUNCH;
  }
  const int nblk = (int)adj_grid(B, G);
  {
    const int ntab = (int)adj_grid(B, wide ? 1 : adj_tab_G(n, d) * FF_ADJ_WPW);     // workgroups (= private tables) of the tabulated kernel
    double* wtot = a.trows + (size_t)ntab * 2 * FF_DEP_NLDS * FF_DEP_ROW;
    const unsigned ndep = net->radial_table ? (unsigned)((2 * FF_DEP_NLDS * FF_DEP_ROW + FF_DEPR_EX - 1) / FF_DEPR_EX) : 0u;
    FF_LAUNCH(ff_adj_reduce_kernel, (unsigned)P + ndep, FF_DEPR_EX * FF_DEPR_TY, stream, *net, (const double*)a.off_table, nblk * G, P,
              (const double*)a.rows, grad_params, ntab, (const double*)(net->radial_table ? a.trows : nullptr), wtot);
    FF_LAUNCH_CHECK();
  }
  if (net->radial_table) {
    const int ntab = (int)adj_grid(B, wide ? 1 : adj_tab_G(n, d) * FF_ADJ_WPW);
    double* wtot = a.trows + (size_t)ntab * 2 * FF_DEP_NLDS * FF_DEP_ROW;
    FF_LAUNCH(ff_dep_contract_kernel, (unsigned)(net->He + net->Hm), FF_RBLOCK(256), stream, *net, (const double*)a.off_table,
              (const double*)wtot, grad_params);
    FF_LAUNCH_CHECK();
  }
  return FF_OK;
}

}  // extern "C"
