// ff_cnf_fwd.hip -- fused forward CNF integrations:
//   MODE 0  CNF.generate      (src/flow.py:42-44)     state z                      heads eta
//   MODE 1  CNF.delta_logp    (src/flow.py:51-55)     state (z, Delta)             heads eta, eta'
//   MODE 2  local-energy pass (src/VMC.py:46-49, replaces the 2+2*n*d nested adjoint solves of
//           src/utils.py:40-65)  state (z, J = dz/dx, kbar = lap_x z, Delta, grad_x Delta, lap_x Delta)
//                                                                                heads eta .. eta'''
// All of one walker's stages, error control and accept/reject happen on chip; HBM sees the walker's
// coordinates once on the way in and the results once on the way out.
//
// Per RHS evaluation (all lanes of the wave):
//   1. every lane publishes its stage value z_i (and kbar_i) to LDS;
//   2. "radius phase": the wave's G*R radii (pairs r_ab, one-body r_a) are dealt one per lane (the assignment is fixed for
//      the launch); a lane gets the derivative heads of eta (or mu) at its radius -- from the per-launch table (TAB
//      instantiations) or by evaluating the H sigmoids (direct instantiations) -- and leaves in LDS what does not depend
//      on the direction: MODE 0/1 the heads, MODE 2 a record per radius plus the radius' contributions to its particles'
//      own rows (v, Dv[kbar], grad div);
//   3. MODE 2 "jet sweep": lane (g,i) gathers its own row, then pushes its direction u_i = dz/dx_i through every radius
//      term as a 2nd-order Taylor jet (first-order part -> dJ/dt column, quadratic part -> source of kbar and lap Delta);
//      MODE 0/1 "component phase": lane (g,i) assembles v_i (and div v) from the heads of its particle's radii;
//   4. Dormand-Prince stage bookkeeping; per-walker error norm and step-size control (ff_ode.h).
// n >= 8 uses ff_eloc_split_kernel (two lanes per direction) for MODE 2.  The local-energy finish (Slater table +
// contraction with the sensitivities) is at the end of the file.
#include <atomic>
#include "ff_common.h"
#include "ff_ode.h"
#include "ff_slater.h"
#include "ff_eloc_ws.h"
#define FF_RADIAL_BUILD_KERNELS
#include "ff_radial.h"

// FF_STAMPS: diagnostic build only (tools/kbench.py --stamps): per-phase s_memtime shares of the RHS loop,
// added into stats[8..] as 64-bit counters.  Never defined in the product build.
#ifdef FF_STAMPS
#define FF_STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += t_ - stamp_prev; stamp_prev = t_; } while (0)
#else
#define FF_STAMP(i) do { } while (0)
#endif

// which kernels use the LDS-table exp (ff_exp_tab): tuning knob, see tools/kbench.py A/B runs
#ifndef FF_TAB_MODE
#define FF_TAB_MODE(MODE) true
#endif

#include "ff_fwd_args.h"

#ifndef FF_FORM_EARLY
#define FF_FORM_EARLY 1   // MODE 2: form the stage input between the two halves of the radius phase
#endif
#ifndef FF_SWEEP_CH
#define FF_SWEEP_CH 2   // records per look-ahead chunk of the jet sweep (measured at n = 6: 2 -> 1.49 ms, 3 -> 1.52, 4 spills)
#endif
#ifndef FF_FWD_WAVES_PER_SIMD
#define FF_FWD_WAVES_PER_SIMD 1
#endif
// The tabulated flow kernel (MODE 0) carries one double per lane and hides its table fetches behind other waves: up to 6 particles it
// is compiled for three waves per SIMD (<= 168 registers, no scratch; left alone hipcc drifted from 159 to 169 registers
// with an unrelated change and the flow pass went from 112 to 143 us).  Larger walkers and delta_logp (MODE 1) would spill
// under that bound; the local-energy kernel (MODE 2) needs the whole register file.
#ifndef FF_FLOW_WAVES
#define FF_FLOW_WAVES 3      // waves per SIMD the tabulated flow kernel is compiled for (165 registers; 4: 128 + 140 B of scratch, 5: 95 + 280 B)
#endif
// instantiations whose step is laid out stage by stage (compile-time stages: DESIGN.md 3s)
#ifndef FF_FWD_STATIC
#define FF_FWD_STATIC(N, D, MODE, TAB) ((MODE) == 0 && (TAB) && (N) * (D) <= 12)
#endif
template <int N, int D, int MODE, bool TAB>
__global__ void __launch_bounds__(FF_WAVE, (MODE == 0 && TAB && N * D <= 12) ? FF_FLOW_WAVES : FF_FWD_WAVES_PER_SIMD)
ff_ode_fwd_kernel(ff_fwd_args A) {
  if constexpr (MODE == 0) FF_SETPRIO();      // the flow pass runs beside the tail of the prefetched sampler (ff_common.h)
  using Gm = ff_geom<N, D>;
  constexpr int M = Gm::M, G = Gm::G, P = Gm::P, R = Gm::RA;
  constexpr int NH = MODE == 0 ? 1 : (MODE == 1 ? 2 : 4);
  constexpr int NV = MODE == 0 ? 1 : (MODE == 1 ? 2 : M + 5);
  // state slots of a lane: [0] z_i; MODE1: [1] Delta; MODE2: [1..M] u_k = dz_k/dx_i, [M+1] kbar_i,
  // [M+2] dDelta/dx_i, [M+3] Delta, [M+4] L_i
  constexpr int IDL = MODE == 1 ? 1 : M + 3;  // slot of the replicated Delta

  // TAB: radial functions from the per-launch table only (no weights, no exp table in LDS); !TAB: direct evaluation
  __shared__ ff_wtab s_w[TAB ? 1 : 2][TAB ? 1 : FF_HPAD];
  __shared__ double s_e2[TAB ? 1 : 64];
  __shared__ double s_z[G][M], s_kb[G][M], s_err[G][M];
  constexpr bool JET = (MODE == 2);
  __shared__ __attribute__((aligned(16))) double s_rr[JET ? 1 : G][JET ? 1 : R][2];   // radius and its reciprocal
  __shared__ __attribute__((aligned(16))) double s_hd[JET ? 1 : G][JET ? 1 : R][NH];
  // MODE 2: one record per radius with everything about it that does not depend on the direction:
  //   rho (D), 1/r, eta, eta', eta'', A = c (eta'' r + (1+D) eta'), B = c (eta''' r + (2+D) eta''), c (eta' r + D eta)
  //   (c = 2 for pairs, 1 for one-body radii)
  constexpr int RECW = (D + 7 + 1) & ~1;   // + the radius' share of div v
  // walker stride padded so that the G records a wave reads together (one address per walker, broadcast to its M lanes)
  // fall into different LDS banks: stride mod 32 dwords is an odd multiple of 4
  constexpr int RECS0 = R * RECW, RECS = RECS0 + ((RECS0 % 4 == 2) ? 0 : ((RECS0 % 4 == 0) ? 2 : 1));
  static_assert(!JET || ((2 * RECS) % 32) % 8 == 4, "record stride");
  __shared__ __attribute__((aligned(16))) double s_rec[JET ? G * RECS : 1];
  // s_qt is used twice per evaluation: first as T[g][a][j][3][D], the contributions of partner j (j = a: one-body term)
  // to particle a's own rows (v, Dv[kbar], grad div), written in the radius phase and gathered before the sweep; then
  // as the (M x (M+1)) transposition buffer of the quadratic sources
  constexpr int QTW = ((3 * N > M) ? 3 * N + 1 : M + 1) | 1;   // odd row length: conflict-free transposition
  constexpr int TROW = N * 3 * D + 1;                              // T row of one particle (padded likewise)
  static_assert(N * TROW <= M * QTW, "T must fit the transposition buffer");
  __shared__ __attribute__((aligned(16))) double s_qt[JET ? G * M * QTW : 1];
  __shared__ int s_pa[R], s_pb[R], s_any;
  __shared__ double s_rmin[G][R];   // smallest value each radius took during the walker's integration (walker_cost)
  // lane-private LDS columns for y and the error accumulator: only while 4 workgroups still fit a CU's LDS
  constexpr bool LDS_STATE = (MODE == 2 && M <= 12);
  __shared__ double s_yv[LDS_STATE ? NV : 1][FF_WAVE], s_cv[LDS_STATE ? NV : 1][FF_WAVE];

  const int lane = threadIdx.x;
  const int g = lane / M, i = lane % M;
  const bool ingrp = g < G;
  const int gg = ingrp ? g : 0;  // safe LDS row for the idle tail lanes
  const int ai = i / D, ci = i % D;
  const double* __restrict__ rtab = A.net.radial_table;
  if constexpr (TAB) {
    if (rtab[3] != 0.0) {   // table unusable for these weights: leave the call to the direct kernel
      if (lane == 0 && blockIdx.x == 0) *A.evt = A.evt_id;
      return;
    }
  } else {
    if (A.evt && *A.evt != A.evt_id) return;   // fallback launch that is not needed
    ff_fill_exp2_table(s_e2, lane);
    ff_load_weights(s_w, A.net, lane);
  }
  bool off_table = false;
  if (lane == 0) {
    int p = 0;
    for (int a = 0; a < N; a++)
      for (int b = a + 1; b < N; b++) { s_pa[p] = a; s_pb[p] = b; p++; }
    for (int a = 0; a < N; a++) { if (P + a < R) { s_pa[P + a] = a; s_pb[P + a] = -1; } }
  }
  FF_WG1_SYNC();
  const int He = A.net.He, Hm = A.net.Hm;
  const bool has_mu = Hm > 0;
  const int nrad = has_mu ? (P + N) : P;
  const double tab_inv_h = TAB ? rtab[0] : 0.0, tab_h = TAB ? rtab[1] : 0.0;
  // NH derivative heads of eta (t = 0) / mu (t = 1) at radius r
  auto heads = [&](int t, double r, double* hd) {
    if constexpr (TAB) {
      if (!ff_heads_table<NH>(rtab, tab_inv_h, tab_h, t, r, hd)) {
        off_table = true;
#pragma unroll
        for (int m = 0; m < NH; m++) hd[m] = 0.0;
      }
    } else {
      ff_heads<NH, FF_TAB_MODE(MODE)>(s_w[t], s_e2, t ? Hm : He, r, hd);
    }
  };
  const double rtol = A.rtol, atol = A.atol;
  constexpr double NT = MODE == 0 ? M : (MODE == 1 ? M + 1 : M * (M + 4) + 1);
  const int64_t ngroups = (A.B + G - 1) / G;
  // ODE statistics of this workgroup's walkers, kept in LDS (nothing loop-carried in registers across the persistent loop)
  __shared__ int s_st[4];
  if (lane < 4) s_st[lane] = 0;
  // the radii this lane evaluates in the radius phase (slot qk: radius lane + 64 qk of the wave's G*nrad), fixed for the
  // whole launch: walker slot, particles, radius index packed into one register each
  constexpr int NQ = (G * R + FF_WAVE - 1) / FF_WAVE;
  int rq_id[NQ];
#pragma unroll
  for (int qk = 0; qk < NQ; qk++) {
    const int q = lane + qk * FF_WAVE;
    const bool act = q < G * nrad;
    const int qg = act ? q / nrad : 0, p = act ? q - qg * nrad : 0;
    rq_id[qk] = act ? (qg | (s_pa[p] << 4) | ((s_pb[p] < 0 ? 15 : s_pb[p]) << 8) | (p << 12)) : -1;   // N <= 12, G <= 16
  }
#ifdef FF_STAMPS
  unsigned long long stamp_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev = __builtin_amdgcn_s_memtime();
  unsigned long long stage_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, stage_cnt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long stamp_t0 = stamp_prev, stamp_r0 = __builtin_readcyclecounter() * 0 + wall_clock64();
#endif

  __shared__ long long s_next;
  for (int64_t grp = blockIdx.x;; grp += gridDim.x) {
    if (A.queue) {   // persistent grid: next group from the launch's work counter (the order is by schedule key: cost class + 4 x planned steps, costliest first)
      FF_WG1_SYNC();
      if (lane == 0) s_next = (long long)atomicAdd(A.queue + (TAB ? 0 : 1), 1ULL);
      FF_WG1_SYNC();
      grp = s_next;
    }
    if (grp >= ngroups) break;
    const int64_t bq = grp * G + g;
    const bool inb = ingrp && bq < A.B;
    const int64_t b = ff_opt_load(A.order, inb, bq, A.y_in, (int32_t)bq);   // the walker this lane group integrates
    // (routing by cost class, launch_routed below: with heavy_mode = 2 the walkers of class >= heavy_class belong to another launch)
    // (only where this kernel is the local-energy pass's default, up to 3 particles: the larger instantiations sit at the edge of the
    // register allocator's bug of docs/LOG.md (round 2) and are left exactly as they were)
    const bool valid = inb && !(MODE == 2 && N <= 3 && A.heavy_mode == 2 && ff_opt_load(A.wclass, inb, b, A.y_in, (int32_t)0) >= A.heavy_class);
    // Stage storage, 5 vectors instead of the textbook 7 (y, k0..k5): c0..c2 hold k0..k2 up to stage 3; once k3 is
    // known the remaining stage inputs and the error accumulator are formed and overwrite them:
    //   c0 = input of stage 4, c1 = partial input of stage 5, c2 = partial y_new, c3 = partial error.
    // k0 is therefore gone at the accept/reject decision: an accepted step takes k6 (FSAL), a rejected one
    // re-evaluates f(y) (stage 0; rejections are ~1 % of the wave-steps).
    // In the local-energy kernel y and c3 (touched a few times per step) live in lane-private LDS columns: that is
    // what lets the remaining state fit VGPRs + AGPRs without scratch round trips in every evaluation.
    ff_lane_vec<NV, LDS_STATE> y(&s_yv[0][0], lane), c3(&s_cv[0][0], lane);
    double c0[NV], c1[NV], c2[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) { y[v] = 0.0; c0[v] = 0.0; c1[v] = 0.0; c2[v] = 0.0; c3[v] = 0.0; }
    y[0] = ff_opt_load(A.y_in, valid, b * M + i, A.y_in, 0.25 * (i + 1) + 0.125 * ((i * 7) % 5));  // idle rows: finite, distinct
    if constexpr (MODE == 2) {
#pragma unroll
      for (int k = 0; k < M; k++) y[1 + k] = (k == i) ? 1.0 : 0.0;
    }
    ff_stepper S;
    S.begin(A.ta, A.tb, valid);
    // warm start (ff_ode.walker_h_init): the step size to try first, instead of the probe evaluation of the Hairer start
    // (local-energy pass) walkers of a low cost class: looser tolerance for the sensitivity components, larger first step
    const bool loose = MODE == 2 && ff_opt_load(A.wclass, valid, b, A.y_in, (int32_t)0x7fffffff) <= A.sens_class;
    const double hwarm = ff_open_step(ff_opt_load(A.h_init, valid, A.h_scale < 0.0 ? 0 : b, A.y_in, 0.0) * (loose ? A.h_scale_loose : fabs(A.h_scale)),
                                      A.ta, A.tb, MODE == 2 ? 0 : A.h_equal);
    const bool warm = hwarm > 0.0;
    double hmax_acc = 0.0;
    int s = -2, nev = 0;
    double h0v = 0.0, d1v = 0.0;
    double rmin_q[NQ];   // this lane's radii keep their slots from evaluation to evaluation
#pragma unroll
    for (int k = 0; k < NQ; k++) rmin_q[k] = 1e300;

    // group-wide sum of a per-lane partial (all lanes of a walker get the identical result)
    auto group_sum = [&](double part) -> double {
      if (ingrp) s_err[g][i] = part;
      FF_WG1_SYNC();
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < M; j++) t += s_err[gg][j];
      FF_WG1_SYNC();
      return t;
    };
    const double sens_w = loose ? A.sens_w : 1.0;
    auto wgt = [&](int v) -> double {   // Delta is replicated: count it once; sensitivity components: their own tolerance
      if (MODE >= 1 && v == IDL && i != 0) return 0.0;
      return (MODE == 2 && v >= 1) ? sens_w : 1.0;
    };

    // One evaluation and what the Dormand-Prince step does with it, as a body that can be instantiated per stage (SG = 1 .. 6: the
    // stage is a compile-time constant; FF_STAGE_DYN: the run-time value s).  Which instantiations lay their step out
    // stage by stage is decided by FF_FWD_STATIC -- the others run this body in the loop over a run-time stage they always had.
    // Returns true when every walker of the wave has finished.
    constexpr bool STATIC_STAGES = FF_FWD_STATIC(N, D, MODE, TAB);
    auto evaluate = [&](auto stage_tag) -> bool {
      constexpr int SG = decltype(stage_tag)::value;
      if constexpr (SG == FF_STAGE_DYN && STATIC_STAGES) FF_ASSUME(s <= 0);
      const int sv = SG == FF_STAGE_DYN ? s : SG;
      // ------------------------------------------------------------------ stage input
      // The candidate state for stage s (a single wave-uniform switch; code size matters: the whole RHS loop has
      // to stay inside the 64 KB instruction cache that two CUs share).  Only the two published slots are formed
      // before the radius phase; the full vector is formed after it, so the radius loop runs without `in[]` live.
      const double h = S.h;
      // in = gy*y + g0*c0 + g1*c1 + g2*c2 with wave-uniform stage coefficients: one code path for all stages
      // (c1..c3 are zero-initialised, so unused terms are exact zeros)
      double gy = 1.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
      switch (sv) {
        case -1: g0 = h0v * S.dir; break;
        case 1: g0 = h * FF_A10; break;
        case 2: g0 = h * FF_A20; g1 = h * FF_A21; break;
        case 3: g0 = h * FF_A30; g1 = h * FF_A31; g2 = h * FF_A32; break;
        case 4: gy = 0.0; g0 = 1.0; break;
        case 5: gy = 0.0; g1 = 1.0; break;
        case 6: gy = 0.0; g2 = 1.0; break;
        default: break;   // -2, 0: the state itself
      }
      auto form = [&](double* dst, const int v0, const int v1, const int stride) {
#pragma unroll
        for (int v = v0; v < v1; v += stride) dst[v] = fma(g2, c2[v], fma(g1, c1[v], fma(g0, c0[v], gy * y[v])));
      };
      FF_STAMP(0);
      // ------------------------------------------------------------------ publish
      FF_WG1_SYNC();
      {
        double pub[NV];
        form(pub, 0, MODE == 2 ? M + 2 : 1, MODE == 2 ? M + 1 : 1);   // slots 0 and (MODE 2) M+1 only
        if (ingrp) {
          s_z[g][i] = pub[0];
          if constexpr (MODE == 2) s_kb[g][i] = pub[M + 1];
        }
      }
      FF_WG1_SYNC();
      FF_STAMP(1);
      // ------------------------------------------------------------------ radius phase
      // (two halves: every radius of the wave is formed and -- table kernels -- its table row requested before the
      //  first one is evaluated, so the passes over the G*R radii share one memory round trip)
      double rq_rho[NQ][D], rq_dk[JET ? NQ : 1][D], rq_r[NQ], rq_ri[NQ], rq_T[NQ][TAB ? NH + 5 : 1], rq_dr[NQ];
      bool rq_ok[NQ];
#pragma unroll
      for (int qk = 0; qk < NQ; qk++) {
        const int id = rq_id[qk];
        const bool act = id >= 0;
        const int qg = act ? (id & 15) : 0, a = act ? ((id >> 4) & 15) : 0, bq = act ? ((id >> 8) & 15) : 15;
        const bool pair = bq != 15;
        const int bb = pair ? bq : a;
        double r2 = 0.0;
#pragma unroll
        for (int c = 0; c < D; c++) {
          rq_rho[qk][c] = s_z[qg][a * D + c] - (pair ? s_z[qg][bb * D + c] : 0.0);
          if constexpr (JET) rq_dk[qk][c] = s_kb[qg][a * D + c] - (pair ? s_kb[qg][bb * D + c] : 0.0);
          r2 = fma(rq_rho[qk][c], rq_rho[qk][c], r2);
        }
        ff_sqrt_rcp(r2, rq_r[qk], rq_ri[qk]);
        if (act) rmin_q[qk] = fmin(rmin_q[qk], rq_r[qk]);
        rq_dr[qk] = 0.0;
        rq_ok[qk] = true;
        if constexpr (TAB) {
          rq_ok[qk] = ff_table_fetch<NH>(rtab, tab_inv_h, tab_h, pair ? 0 : 1, rq_r[qk], rq_T[qk], rq_dr[qk]);
          if (act && !rq_ok[qk]) off_table = true;
        }
      }
      // the full stage input is formed here, while the table rows requested above are on their way
      double in[NV], out[NV];
      if constexpr (MODE == 2 && FF_FORM_EARLY) form(in, 0, NV, 1);
#pragma unroll
      for (int qk = 0; qk < NQ; qk++) {
        const int id = rq_id[qk];
        if (id < 0) break;
        const int qg = id & 15, a = (id >> 4) & 15, bq = (id >> 8) & 15, p = id >> 12;
        const bool pair = bq != 15;
        const int bb = pair ? bq : a;
        const double* rho = rq_rho[qk];
        const double r = rq_r[qk], ri = rq_ri[qk];
        double hd[NH];
        if constexpr (TAB) {
          if (rq_ok[qk]) ff_table_eval<NH>(rq_T[qk], rq_dr[qk], hd);
          else {
#pragma unroll
            for (int m = 0; m < NH; m++) hd[m] = 0.0;
          }
        } else {
          heads(pair ? 0 : 1, r, hd);
        }
        if constexpr (!JET) {
          s_rr[qg][p][0] = r;
          s_rr[qg][p][1] = ri;
#pragma unroll
          for (int m = 0; m < NH; m++) s_hd[qg][p][m] = hd[m];
        } else {
          const double cf = pair ? 2.0 : 1.0;
          const double f0 = hd[0], f1 = hd[1], f2 = hd[NH > 2 ? 2 : 0], f3 = hd[NH > 3 ? 3 : 0];
          const double Ac = cf * fma(f2, r, (1.0 + D) * f1), Bc = cf * fma(f3, r, (2.0 + D) * f2);
          double* rec = &s_rec[qg * RECS + p * RECW];
#pragma unroll
          for (int c = 0; c < D; c++) rec[c] = rho[c];
          rec[D] = ri; rec[D + 1] = f0; rec[D + 2] = f1; rec[D + 3] = f2; rec[D + 4] = Ac; rec[D + 5] = Bc;
          rec[D + 6] = cf * fma(f1, r, D * f0);                 // this radius' share of div v
          // own-row contributions of this radius: +x to particle a from partner bb, -x to particle bb from partner a
          const double* dk = rq_dk[qk];
          double rdk = 0.0;
#pragma unroll
          for (int c = 0; c < D; c++) rdk = fma(rho[c], dk[c], rdk);
          const double F1k = f1 * (rdk * ri), gq = Ac * ri;
          double* Ta = &s_qt[qg * M * QTW + a * TROW + bb * 3 * D];
          double* Tb = &s_qt[qg * M * QTW + bb * TROW + a * 3 * D];
#pragma unroll
          for (int c = 0; c < D; c++) {
            const double pv = f0 * rho[c], pw = fma(F1k, rho[c], f0 * dk[c]), pg = gq * rho[c];
            Ta[c] = pv; Ta[D + c] = pw; Ta[2 * D + c] = pg;
            if (pair) { Tb[c] = -pv; Tb[D + c] = -pw; Tb[2 * D + c] = -pg; }
          }
        }
      }
      FF_WG1_SYNC();
      nev++;
      FF_STAMP(2);
      // ------------------------------------------------------------------ right-hand side
      if constexpr (!(MODE == 2 && FF_FORM_EARLY)) form(in, 0, NV, 1);   // at stage 6 this is the candidate new state
      FF_STAMP(6);
      const double* sz = s_z[gg];
      double sumq = 0.0, ddiv = 0.0, qdiv = 0.0, divv = 0.0;
      double vi = 0.0, dvk = 0.0, gdi = 0.0;
      if constexpr (MODE == 2) {
        // jet phase: this lane's direction u = in[1..M].  One static sweep over the radii does three things:
        //  - first-order jet of every pair term along u  -> du  (column of dJ/dt)
        //  - quadratic part of the second-order jet      -> qv  (source of kbar), qdiv (source of lap Delta)
        //  - the lane's own coordinate (ai,ci) of v, Dv[kbar], grad div (picked with selects), so no second,
        //    latency-bound pass over LDS is needed.
        // Scheduling fences every few radii keep hipcc from hoisting all LDS reads of the sweep at once
        // (that would need > 400 VGPRs and spill to scratch).
        // own rows first (the buffer they sit in is reused by the transposition below)
        {
          const double* T = &s_qt[gg * M * QTW + ai * TROW + ci];
#pragma unroll
          for (int j = 0; j < N; j++) {
            const bool use = has_mu || j != ai;
            const double tv = T[j * 3 * D], tw = T[j * 3 * D + D], tg = T[j * 3 * D + 2 * D];
            vi += use ? tv : 0.0; dvk += use ? tw : 0.0; gdi += use ? tg : 0.0;
          }
        }
        FF_WG1_SYNC();
        double du[M], qv[M];
#pragma unroll
        for (int k = 0; k < M; k++) { du[k] = 0.0; qv[k] = 0.0; }
        const double* u = &in[1];
        // the records stream through a two-deep register buffer: the LDS reads of chunk c+1 are issued before chunk c
        // is computed, so the single resident wave does not sit out an LDS round trip per radius
        constexpr bool PREF = (M <= 12);   // look-ahead only where the registers are there for it
        constexpr int CH = PREF ? FF_SWEEP_CH : 2, RT = P + N, NCH = (RT + CH - 1) / CH;
        constexpr ff_pair_table<N> PT{};
        double hb[2][CH][RECW];
        auto load_chunk = [&](int c, double (*buf)[RECW]) {
#pragma unroll
          for (int q = 0; q < CH; q++) {
            const int p = c * CH + q;
            if (p < RT) {
#pragma unroll
              for (int m = 0; m < D + 7; m++) buf[q][m] = s_rec[gg * RECS + p * RECW + m];
            }
          }
        };
        if constexpr (PREF) load_chunk(0, hb[0]);
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          if constexpr (PREF) { if (c + 1 < NCH) load_chunk(c + 1, hb[(c + 1) & 1]); }
          else load_chunk(c, hb[0]);
#pragma unroll
          for (int q = 0; q < CH; q++) {
            const int p = c * CH + q;
            if (p < RT && (p < P || has_mu)) {
              const bool pair = p < P;
              const int a = pair ? PT.a[p < P ? p : 0] : p - P, bq = pair ? PT.b[p < P ? p : 0] : 0;
              const double* hq = hb[PREF ? (c & 1) : 0][q];
              const double* rho = hq;
              const double ri = hq[D], f0 = hq[D + 1], f1 = hq[D + 2], f2 = hq[D + 3], Ac = hq[D + 4], Bc = hq[D + 5];
              double dl[D], rd = 0.0, dd = 0.0;
#pragma unroll
              for (int cc = 0; cc < D; cc++) {
                dl[cc] = pair ? u[a * D + cc] - u[bq * D + cc] : u[a * D + cc];
                rd = fma(rho[cc], dl[cc], rd);
                dd = fma(dl[cc], dl[cc], dd);
              }
              const double r1 = rd * ri, r1s = r1 * r1;
              const double r2q = (dd - r1s) * ri;
              const double F1 = f1 * r1, F2 = fma(f2, r1s, f1 * r2q), F1x2 = F1 + F1;
#pragma unroll
              for (int cc = 0; cc < D; cc++) {
                const double g1 = fma(F1, rho[cc], f0 * dl[cc]);
                const double g2 = fma(F2, rho[cc], F1x2 * dl[cc]);
                du[a * D + cc] += g1;
                qv[a * D + cc] += g2;
                if (pair) { du[bq * D + cc] -= g1; qv[bq * D + cc] -= g2; }
              }
              ddiv = fma(Ac, r1, ddiv);
              qdiv = fma(Bc, r1s, fma(Ac, r2q, qdiv));
              divv += hq[D + 6];
            }
          }
          FF_SCHED_FENCE();
        }
        FF_STAMP(7);
#pragma unroll
        for (int k = 0; k < M; k++) out[1 + k] = du[k];
        // transpose-reduce the quadratic sources: lane c needs sum_j qv_j[c]
        if (ingrp) {
#pragma unroll
          for (int k = 0; k < M; k++) s_qt[(g * M + i) * QTW + k] = qv[k];
        }
        FF_WG1_SYNC();
#pragma unroll
        for (int j = 0; j < M; j++) sumq += s_qt[(gg * M + j) * QTW + i];
        FF_STAMP(3);
      } else {
        if constexpr (MODE == 1) {
          for (int p = 0; p < nrad; p++) {
            const double c = p < P ? 2.0 : 1.0;
            divv = fma(c, fma(s_hd[gg][p][NH > 1 ? 1 : 0], s_rr[gg][p][0], D * s_hd[gg][p][0]), divv);
          }
        }
        // component phase (generate / delta_logp): coordinate (ai, ci) of v
        const double zc = sz[ai * D + ci];
#pragma unroll
        for (int bq = 0; bq < N; bq++) {
          const bool self = (bq == ai);
          const int lo = bq < ai ? bq : ai, hi = bq < ai ? ai : bq;
          const int p = self ? 0 : ff_pair_index(N, lo, hi);
          vi = fma(self ? 0.0 : s_hd[gg][p][0], zc - sz[bq * D + ci], vi);
        }
        if (has_mu) vi = fma(s_hd[gg][P + ai][0], zc, vi);
      }
      out[0] = vi;
      if constexpr (MODE == 1) out[1] = -divv;
      if constexpr (MODE == 2) {
        out[M + 1] = sumq + dvk;
        out[M + 2] = -ddiv;
        out[M + 3] = -divv;
        out[M + 4] = -fma(gdi, in[M + 1], qdiv);
      }
      FF_STAMP(4);
      // ------------------------------------------------------------------ consume
#if defined(FF_STAMPS) && defined(FF_STAMPS_TRACE)
      const int s_prev = sv;
#endif
      if (sv == -2) {
#pragma unroll
        for (int v = 0; v < NV; v++) c0[v] = out[v];
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double isc = wgt(v) * ff_rcp(fma(fabs(y[v]), rtol, atol));
          p0 = fma(y[v] * isc, y[v] * isc, p0);
          p1 = fma(c0[v] * isc, c0[v] * isc, p1);
        }
        const double d0 = sqrt(group_sum(p0) * (1.0 / NT));
        d1v = sqrt(group_sum(p1) * (1.0 / NT));
        h0v = S.h0(d0, d1v);
        s = -1;
        // every walker of the wave brings its own first step: no probe evaluation
        if (!ff_wave_or(&s_any, lane, (!S.done && !warm) ? 1 : 0)) {
          S.habs = fmin(hwarm, S.interval);
          S.plan();
          s = 1;
        }
      } else if (sv == -1) {
        double p2 = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double t = (out[v] - c0[v]) * wgt(v) * ff_rcp(fma(fabs(y[v]), rtol, atol));
          p2 = fma(t, t, p2);
        }
        const double d2 = sqrt(group_sum(p2) * (1.0 / NT)) / h0v;
        S.init_habs(h0v, d1v, d2);
        if (warm) S.habs = fmin(hwarm, S.interval);
        S.plan();
        s = 1;
      } else if (sv == 0) {
#pragma unroll
        for (int v = 0; v < NV; v++) c0[v] = out[v];
        s = 1;
      } else if (sv == 1) {
#pragma unroll
        for (int v = 0; v < NV; v++) c1[v] = out[v];
        s = 2;
      } else if (sv == 2) {
#pragma unroll
        for (int v = 0; v < NV; v++) c2[v] = out[v];
        s = 3;
      } else if (sv == 3) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double k0v = c0[v], k1v = c1[v], k2v = c2[v], k3v = out[v];
          c0[v] = fma(h, FF_A40 * k0v + FF_A41 * k1v + FF_A42 * k2v + FF_A43 * k3v, y[v]);
          c1[v] = fma(h, FF_A50 * k0v + FF_A51 * k1v + FF_A52 * k2v + FF_A53 * k3v, y[v]);
          c2[v] = fma(h, FF_B0 * k0v + FF_B2 * k2v + FF_B3 * k3v, y[v]);
          c3[v] = h * (FF_E0 * k0v + FF_E2 * k2v + FF_E3 * k3v);
          if constexpr (STATIC_STAGES) { FF_OPAQUE(c0[v]); FF_OPAQUE(c1[v]); FF_OPAQUE(c2[v]); }      // (not sunk into the stages that use them)
        }
        s = 4;
      } else if (sv == 4) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
          c1[v] = fma(h * FF_A54, out[v], c1[v]);
          c2[v] = fma(h * FF_B4, out[v], c2[v]);
          c3[v] = fma(h * FF_E4, out[v], c3[v]);
          if constexpr (STATIC_STAGES) { FF_OPAQUE(c1[v]); FF_OPAQUE(c2[v]); }
        }
        s = 5;
      } else if (sv == 5) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
          c2[v] = fma(h * FF_B5, out[v], c2[v]);
          c3[v] = fma(h * FF_E5, out[v], c3[v]);
          if constexpr (STATIC_STAGES) { FF_OPAQUE(c2[v]); }
        }
        s = 6;
      } else {
        double pe = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double e = fma(h * FF_E6, out[v], c3[v]);
          const double t = e * wgt(v) * ff_rcp(fma(fmax(fabs(y[v]), fabs(in[v])), rtol, atol));
          pe = fma(t, t, pe);
        }
        const double err = sqrt(group_sum(pe) * (1.0 / NT));
        const bool was_active = !S.done;
        const bool acc = S.decide(err, A.max_steps);
        if (acc) hmax_acc = fmax(hmax_acc, fabs(h));
        if (acc) {
#pragma unroll
          for (int v = 0; v < NV; v++) { y[v] = in[v]; c0[v] = out[v]; }
        }
        S.plan();
        // wave-wide: anybody still integrating?  anybody rejected (then everyone passes through stage 0)?
        const int any = ff_wave_or(&s_any, lane, S.done ? 0 : ((was_active && !acc) ? 3 : 1));
        s = (any & 2) ? 0 : 1;
        if (!any) { FF_STAMP(5); return true; }
      }
#if defined(FF_STAMPS) && defined(FF_STAMPS_TRACE)
      { unsigned long long t_ = __builtin_amdgcn_s_memtime(); stage_acc[s_prev + 2] += t_ - stamp_prev; stage_cnt[s_prev + 2]++; }
#endif
      FF_STAMP(5);
      return false;
    };
    if constexpr (STATIC_STAGES) {
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
    } else {
#pragma unroll 1
      for (;;) {
        if (evaluate(ff_stage_c<FF_STAGE_DYN>{})) break;
      }
    }
    // ---------------------------------------------------------------------- results
    if (A.wcost) {   // wave-uniform
#pragma unroll
      for (int qk = 0; qk < NQ; qk++) {
        if (rq_id[qk] >= 0) s_rmin[rq_id[qk] & 15][rq_id[qk] >> 12] = rmin_q[qk];
      }
      FF_WG1_SYNC();
    }
    if (valid) {
      // a walker whose integration failed (NaN error norm, max_steps) must not pass for a result: its outputs are NaN,
      // which every consumer (finish kernel, estimator, parameter gradient) propagates -- the stats word is only a diagnostic
      const bool failed = S.fail != 0;
#ifdef FF_NO_POISON
      const double bad = 0.0;
#else
      const double bad = failed ? __builtin_nan("") : 0.0;
#endif
      A.y_out[b * M + i] = y[0] + bad;
      if constexpr (MODE >= 1) { if (i == 0) A.dl_out[b] = y[IDL] + bad; }
      if constexpr (MODE == 2) {
#pragma unroll
        for (int k = 0; k < M; k++) A.Jt[(b * M + i) * M + k] = y[1 + k];
        A.kbar[b * M + i] = y[M + 1];
        A.dD[b * M + i] = y[M + 2];
        A.Lpart[b * M + i] = y[M + 4];
      }
      if (i == 0) {
        if (A.h_out) A.h_out[b] = hmax_acc > 0.0 ? hmax_acc : hwarm;
        if (A.wcost) {
          double rm = 1e300;
          for (int p = 0; p < nrad; p++) rm = fmin(rm, s_rmin[g][p]);
          A.wcost[b] = ff_cost_class(S.nacc + S.nrej, rm);
        }
        if (A.stats) { atomicAdd(&s_st[0], nev); atomicMax(&s_st[1], S.nacc); atomicAdd(&s_st[2], S.nrej); if (failed) atomicMax(&s_st[3], 1); }
      }
    }
    FF_WG1_SYNC();
  }
#ifdef FF_STAMPS
  if (A.stats && lane == 0)
    for (int q = 0; q < 9; q++) atomicAdd((unsigned long long*)(A.stats + 8) + q, stamp_acc[q]);
  if (A.stats && lane == 0 && blockIdx.x == 0) {   // core ticks and 100 MHz ticks over this wave's life: the core clock
    A.stats[26] = (int)(__builtin_amdgcn_s_memtime() - stamp_t0);
    A.stats[27] = (int)(wall_clock64() - stamp_r0);
  }
#ifdef FF_STAMPS_TRACE
  if (A.stats && lane == 0)   // consume-phase ticks and counts by stage, behind the per-workgroup trace
    for (int q = 0; q < 9; q++) {
      atomicAdd((unsigned long long*)(A.stats + 65600) + q, stage_acc[q]);
      atomicAdd((unsigned long long*)(A.stats + 65600) + 9 + q, stage_cnt[q]);
    }
#endif
#ifdef FF_STAMPS_TRACE   // per-workgroup (start, end, core ticks, hw id) behind the 32 stats words: tools/kbench.py --trace
  if (A.stats && lane == 0 && blockIdx.x < 16384) {
    int* t = A.stats + 32 + 4 * blockIdx.x;
    t[0] = (int)(stamp_r0 & 0x7fffffff);
    t[1] = (int)(wall_clock64() & 0x7fffffff);
    t[2] = (int)(__builtin_amdgcn_s_memtime() - stamp_t0);
    t[3] = (int)(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) & 0xffff) | ((int)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf) << 16);
  }
#endif
#endif
  if constexpr (TAB) { if (off_table) *A.evt = A.evt_id; }
  FF_WG1_SYNC();
  if (A.stats && lane == 0 && (s_st[0] || s_st[3])) {
    atomicAdd(&A.stats[0], s_st[0]);
    atomicMax(&A.stats[1], s_st[1]);
    atomicAdd(&A.stats[2], s_st[2]);
    if (s_st[3]) atomicMax(&A.stats[3], 1);
  }
}


// ===================================================================================================
// Local-energy sensitivities for larger walkers (M = N*D > 12): TWO lanes per direction.
// With one lane per direction the lane state is M+5 doubles and no longer fits the register file next to the working
// set (n = 12: 29 doubles x 5 stage vectors).  Here lane (i, h) holds the half of the column u_i = dz/dx_i that belongs
// to the particles of half h (h = 0: particles [0, N/2), h = 1: the rest), so the lane state is M/2 + 5 doubles again.
// The two lanes of a direction publish their halves to LDS; a lane sweeps the pairs inside its half two-sided and the
// pairs across the halves one-sided (the partner lane does the other side), counting the scalar sources of cross
// pairs half each.  z_i, kbar_i and the own-coordinate picks live on the lane whose half contains particle(i).
// Delta, grad Delta and lap Delta are kept as per-lane partial sums (their equations are linear) and combined at the
// end.  L = 2M lanes per walker, G = 64 / L walkers per wave.
template <int N, int D, bool TAB>
__global__ void __launch_bounds__(FF_WAVE)
ff_eloc_split_kernel(ff_fwd_args A) {
  static_assert(N % 2 == 0, "split kernel needs an even particle number");
  constexpr int M = N * D, NP = N / 2, MH = NP * D, L = 2 * M, G = FF_WAVE / L;
  constexpr int P = N * (N - 1) / 2, R = P + N;
  constexpr int NV = MH + 5;   // [0] z_i (owner lanes), [1..MH] u half, [MH+1] kbar_i (owner), [MH+2] dDelta part, [MH+3] Delta part, [MH+4] lap part
  static_assert(G >= 1, "walker does not fit a wave");

  constexpr int NH = 4;
  __shared__ ff_wtab s_w[TAB ? 1 : 2][TAB ? 1 : FF_HPAD];
  __shared__ double s_e2[TAB ? 1 : 64];
  __shared__ double s_z[G][M], s_kb[G][M], s_err[G][L];
  __shared__ double s_u[G][M][M + 1];
  // one record per radius, as in ff_ode_fwd_kernel: rho (D), 1/r, eta, eta', eta'', A, B, share of div v
  constexpr int RECW = (D + 7 + 1) & ~1;
  constexpr int RECS0 = R * RECW, RECS = RECS0 + ((RECS0 % 4 == 2) ? 0 : ((RECS0 % 4 == 0) ? 2 : 1));
  __shared__ __attribute__((aligned(16))) double s_rec[G * RECS];
  // s_qt: first T[g][a][j][3][D] (own-row contributions of partner j to particle a, j = a: one-body), gathered before the
  // sweep; then the transposition buffer [g][i][h][MH+1] of the quadratic sources
  constexpr int TROW = N * 3 * D + 1;
  constexpr int QTS = (N * TROW > M * 2 * (MH + 1)) ? N * TROW : M * 2 * (MH + 1);
  __shared__ double s_qt[G * QTS];
  __shared__ double s_yv[NV][FF_WAVE], s_cv[NV][FF_WAVE];
  __shared__ int s_pa[R], s_pb[R], s_any;

  const int lane = threadIdx.x;
  const int g = lane / L, idx = lane % L;
  const int h = idx / M, i = idx % M;            // half, direction
  const bool ingrp = g < G;
  const int gg = ingrp ? g : 0;
  const int ai = i / D, ci = i % D;              // particle / component of direction i
  const bool owner = (ai / NP) == h;             // this lane carries z_i, kbar_i and the own-coordinate picks
  const int la = owner ? ai - h * NP : -1;       // local index of particle(i) inside this half
  const double* __restrict__ rtab = A.net.radial_table;
  if constexpr (TAB) {
    if (rtab[3] != 0.0) {
      if (lane == 0 && blockIdx.x == 0) *A.evt = A.evt_id;
      return;
    }
  } else {
    if (A.evt && *A.evt != A.evt_id) return;
    ff_load_weights(s_w, A.net, lane);
    ff_fill_exp2_table(s_e2, lane);
  }
  bool off_table = false;
  if (lane == 0) {
    int p = 0;
    for (int a = 0; a < N; a++)
      for (int b = a + 1; b < N; b++) { s_pa[p] = a; s_pb[p] = b; p++; }
    for (int a = 0; a < N; a++) { s_pa[P + a] = a; s_pb[P + a] = -1; }
  }
  FF_WG1_SYNC();
  const int He = A.net.He, Hm = A.net.Hm;
  const bool has_mu = Hm > 0;
  const int nrad = has_mu ? R : P;
  const double tab_inv_h = TAB ? rtab[0] : 0.0, tab_h = TAB ? rtab[1] : 0.0;
  const double rtol = A.rtol, atol = A.atol;
  constexpr int NQ = (G * R + FF_WAVE - 1) / FF_WAVE;
  int rq_id[NQ];   // the radii this lane evaluates: walker slot | particle a << 4 | particle b (15: none) << 8 | radius << 12
#pragma unroll
  for (int qk = 0; qk < NQ; qk++) {
    const int q = lane + qk * FF_WAVE;
    const bool act = q < G * nrad;
    const int qg = act ? q / nrad : 0, p = act ? q - qg * nrad : 0;
    rq_id[qk] = act ? (qg | (s_pa[p] << 4) | ((s_pb[p] < 0 ? 15 : s_pb[p]) << 8) | (p << 12)) : -1;
  }
  constexpr double NT = 2.0 * M + (double)M * M + 3.0 * L;   // z, kbar (owners), u, and three partial scalars per lane
  const int64_t ngroups = (A.B + G - 1) / G;
  // ODE statistics of this workgroup's walkers, kept in LDS (nothing loop-carried in registers across the persistent loop)
  __shared__ int s_st[4];
  if (lane < 4) s_st[lane] = 0;

  __shared__ long long s_next;
  for (int64_t grp = blockIdx.x;; grp += gridDim.x) {
    if (A.queue) {   // persistent grid: next group from the launch's work counter (the order is by schedule key: cost class + 4 x planned steps, costliest first)
      FF_WG1_SYNC();
      if (lane == 0) s_next = (long long)atomicAdd(A.queue + (TAB ? 0 : 1), 1ULL);
      FF_WG1_SYNC();
      grp = s_next;
    }
    if (grp >= ngroups) break;
    const int64_t bq = grp * G + g;
    const bool valid = ingrp && bq < A.B;
    const int64_t b = ff_opt_load(A.order, valid, bq, A.y_in, (int32_t)bq);
    ff_lane_vec<NV, true> y(&s_yv[0][0], lane), c3(&s_cv[0][0], lane);
    double c0[NV], c1[NV], c2[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) { y[v] = 0.0; c0[v] = 0.0; c1[v] = 0.0; c2[v] = 0.0; c3[v] = 0.0; }
    {
      const double y0 = ff_opt_load(A.y_in, valid && owner, b * M + i, A.y_in, 0.25 * (i + 1) + 0.125 * ((i * 7) % 5));
      y[0] = owner ? y0 : y[0];
    }
#pragma unroll
    for (int k = 0; k < MH; k++) y[1 + k] = (h * MH + k == i) ? 1.0 : 0.0;
    ff_stepper S;
    S.begin(A.ta, A.tb, valid);
    // warm start (ff_ode.walker_h_init): the step size to try first, instead of the probe evaluation of the Hairer start
    // walkers of a low cost class: looser tolerance for the sensitivity components, larger first step (ff_ode.walker_class)
    const bool loose = ff_opt_load(A.wclass, valid, b, A.y_in, (int32_t)0x7fffffff) <= A.sens_class;
    const double hwarm = ff_opt_load(A.h_init, valid, A.h_scale < 0.0 ? 0 : b, A.y_in, 0.0) * (loose ? A.h_scale_loose : fabs(A.h_scale));
    const bool warm = hwarm > 0.0;
    double hmax_acc = 0.0;
    int s = -2, nev = 0;
    double h0v = 0.0, d1v = 0.0;

    auto group_sum = [&](double part) -> double {
      if (ingrp) s_err[g][idx] = part;
      FF_WG1_SYNC();
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < L; j++) t += s_err[gg][j];
      FF_WG1_SYNC();
      return t;
    };
    // slots 0 and MH+1 exist on owner lanes only; sensitivity components: their own tolerance (ff_ode.sens_tol)
    const double sens_w = loose ? A.sens_w : 1.0;
    auto wgt = [&](int v) -> double { return ((v == 0 || v == MH + 1) && !owner) ? 0.0 : (v >= 1 ? sens_w : 1.0); };

    // One evaluation and what the step does with it, instantiated per stage.  Returns true when every walker has finished.
    auto evaluate = [&](auto stage_tag) -> bool {
      constexpr int SG = decltype(stage_tag)::value;
      if constexpr (SG == FF_STAGE_DYN) FF_ASSUME(s <= 0);
      const int sv = SG == FF_STAGE_DYN ? s : SG;
      const double hs = S.h;
      double gy = 1.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
      switch (sv) {
        case -1: g0 = h0v * S.dir; break;
        case 1: g0 = hs * FF_A10; break;
        case 2: g0 = hs * FF_A20; g1 = hs * FF_A21; break;
        case 3: g0 = hs * FF_A30; g1 = hs * FF_A31; g2 = hs * FF_A32; break;
        case 4: gy = 0.0; g0 = 1.0; break;
        case 5: gy = 0.0; g1 = 1.0; break;
        case 6: gy = 0.0; g2 = 1.0; break;
        default: break;
      }
      double in[NV], out[NV];
#pragma unroll
      for (int v = 0; v < NV; v++) in[v] = fma(g2, c2[v], fma(g1, c1[v], fma(g0, c0[v], gy * y[v])));
      // ------------------------------------------------------------------ publish z, kbar (owners) and the u halves
      FF_WG1_SYNC();
      if (ingrp) {
        if (owner) { s_z[g][i] = in[0]; s_kb[g][i] = in[MH + 1]; }
#pragma unroll
        for (int k = 0; k < MH; k++) s_u[g][i][h * MH + k] = in[1 + k];
      }
      FF_WG1_SYNC();
      // ------------------------------------------------------------------ radius phase (lane <-> radius)
      double rq_rho[NQ][D], rq_dk[NQ][D], rq_r[NQ], rq_ri[NQ], rq_T[NQ][TAB ? NH + 5 : 1], rq_dr[NQ];
      bool rq_ok[NQ];
#pragma unroll
      for (int qk = 0; qk < NQ; qk++) {
        const int id = rq_id[qk];
        const bool act = id >= 0;
        const int qg = act ? (id & 15) : 0, a = act ? ((id >> 4) & 15) : 0, bq = act ? ((id >> 8) & 15) : 15;
        const bool pair = bq != 15;
        const int bb = pair ? bq : a;
        double r2 = 0.0;
#pragma unroll
        for (int c = 0; c < D; c++) {
          rq_rho[qk][c] = s_z[qg][a * D + c] - (pair ? s_z[qg][bb * D + c] : 0.0);
          rq_dk[qk][c] = s_kb[qg][a * D + c] - (pair ? s_kb[qg][bb * D + c] : 0.0);
          r2 = fma(rq_rho[qk][c], rq_rho[qk][c], r2);
        }
        ff_sqrt_rcp(r2, rq_r[qk], rq_ri[qk]);
        rq_dr[qk] = 0.0;
        rq_ok[qk] = true;
        if constexpr (TAB) {
          rq_ok[qk] = ff_table_fetch<NH>(rtab, tab_inv_h, tab_h, pair ? 0 : 1, rq_r[qk], rq_T[qk], rq_dr[qk]);
          if (act && !rq_ok[qk]) off_table = true;
        }
      }
#pragma unroll
      for (int qk = 0; qk < NQ; qk++) {
        const int id = rq_id[qk];
        if (id < 0) break;
        const int qg = id & 15, a = (id >> 4) & 15, bq = (id >> 8) & 15, p = id >> 12;
        const bool pair = bq != 15;
        const int bb = pair ? bq : a;
        const double* rho = rq_rho[qk];
        const double* dk = rq_dk[qk];
        const double r = rq_r[qk], ri = rq_ri[qk];
        double hd[NH];
        if constexpr (TAB) {
          if (rq_ok[qk]) ff_table_eval<NH>(rq_T[qk], rq_dr[qk], hd);
          else {
#pragma unroll
            for (int m = 0; m < NH; m++) hd[m] = 0.0;
          }
        } else {
          ff_heads<NH, true>(s_w[pair ? 0 : 1], s_e2, pair ? He : Hm, r, hd);
        }
        const double cf = pair ? 2.0 : 1.0;
        const double f0 = hd[0], f1 = hd[1], f2 = hd[2], f3 = hd[3];
        const double Ac = cf * fma(f2, r, (1.0 + D) * f1), Bc = cf * fma(f3, r, (2.0 + D) * f2);
        double* rec = &s_rec[qg * RECS + p * RECW];
#pragma unroll
        for (int c = 0; c < D; c++) rec[c] = rho[c];
        rec[D] = ri; rec[D + 1] = f0; rec[D + 2] = f1; rec[D + 3] = f2; rec[D + 4] = Ac; rec[D + 5] = Bc;
        rec[D + 6] = cf * fma(f1, r, D * f0);
        double rdk = 0.0;
#pragma unroll
        for (int c = 0; c < D; c++) rdk = fma(rho[c], dk[c], rdk);
        const double F1k = f1 * (rdk * ri), gq = Ac * ri;
        double* Ta = &s_qt[qg * QTS + a * TROW + bb * 3 * D];
        double* Tb = &s_qt[qg * QTS + bb * TROW + a * 3 * D];
#pragma unroll
        for (int c = 0; c < D; c++) {
          const double pv = f0 * rho[c], pw = fma(F1k, rho[c], f0 * dk[c]), pg = gq * rho[c];
          Ta[c] = pv; Ta[D + c] = pw; Ta[2 * D + c] = pg;
          if (pair) { Tb[c] = -pv; Tb[D + c] = -pw; Tb[2 * D + c] = -pg; }
        }
      }
      FF_WG1_SYNC();
      nev++;
      // ------------------------------------------------------------------ own rows (owner lanes), then the jet sweep
      double ddiv = 0.0, qdiv = 0.0, divv = 0.0, vi = 0.0, dvk = 0.0, gdi = 0.0;
      {
        const double* T = &s_qt[gg * QTS + ai * TROW + ci];
#pragma unroll
        for (int j = 0; j < N; j++) {
          const bool use = owner && (has_mu || j != ai);
          const double tv = T[j * 3 * D], tw = T[j * 3 * D + D], tg = T[j * 3 * D + 2 * D];
          vi += use ? tv : 0.0; dvk += use ? tw : 0.0; gdi += use ? tg : 0.0;
        }
      }
      FF_WG1_SYNC();   // s_qt is reused by the transposition below
      double du[MH], qv[MH];
#pragma unroll
      for (int k = 0; k < MH; k++) { du[k] = 0.0; qv[k] = 0.0; }
      const double* u = &in[1];
      const double* uo = s_u[gg][i] + (1 - h) * MH;  // partner lane's half of u_i
      const int pbase = h * NP, obase = (1 - h) * NP;
      const double sg = h ? -1.0 : 1.0;              // records hold rho = z_a - z_b with a < b; half 1's particles are the b's
      // one radius term; two: both particles are mine (accumulate both sides); wsc: weight of the scalar sources
      auto term = [&](const double* rho, const double* dl, const double* rec, int m1, int m2, bool two, bool halfw) {
        double rd = 0.0, dd = 0.0;
#pragma unroll
        for (int c = 0; c < D; c++) { rd = fma(rho[c], dl[c], rd); dd = fma(dl[c], dl[c], dd); }
        const double ri = rec[D], f0 = rec[D + 1], f1 = rec[D + 2], f2 = rec[D + 3];
        const double Ac = halfw ? 0.5 * rec[D + 4] : rec[D + 4], Bc = halfw ? 0.5 * rec[D + 5] : rec[D + 5];
        const double r1 = rd * ri, r1s = r1 * r1, r2q = (dd - r1s) * ri;
        const double F1 = f1 * r1, F2 = fma(f2, r1s, f1 * r2q), F1x2 = F1 + F1;
#pragma unroll
        for (int c = 0; c < D; c++) {
          const double a1 = fma(F1, rho[c], f0 * dl[c]), a2 = fma(F2, rho[c], F1x2 * dl[c]);
          du[m1 * D + c] += a1; qv[m1 * D + c] += a2;
          if (two) { du[m2 * D + c] -= a1; qv[m2 * D + c] -= a2; }
        }
        ddiv = fma(Ac, r1, ddiv);
        qdiv = fma(Bc, r1s, fma(Ac, r2q, qdiv));
        divv = halfw ? fma(0.5, rec[D + 6], divv) : divv + rec[D + 6];
      };
      // pairs inside my half
#pragma unroll
      for (int m1 = 0; m1 < NP; m1++) {
#pragma unroll
        for (int m2 = m1 + 1; m2 < NP; m2++) {
          const double* rec = &s_rec[gg * RECS + ff_pair_index(N, pbase + m1, pbase + m2) * RECW];
          double dl[D];
#pragma unroll
          for (int c = 0; c < D; c++) dl[c] = u[m1 * D + c] - u[m2 * D + c];
          term(rec, dl, rec, m1, m2, true, false);
        }
        if ((m1 & 1) == 1) FF_SCHED_FENCE();
      }
      // pairs across the halves: my particle m, the other half's particle o (one-sided; scalar sources count half)
#pragma unroll
      for (int m = 0; m < NP; m++) {
#pragma unroll
        for (int o = 0; o < NP; o++) {
          const int pm = pbase + m, po = obase + o;
          const double* rec = &s_rec[gg * RECS + ff_pair_index(N, pm < po ? pm : po, pm < po ? po : pm) * RECW];
          double rho[D], dl[D];
#pragma unroll
          for (int c = 0; c < D; c++) { rho[c] = sg * rec[c]; dl[c] = u[m * D + c] - uo[o * D + c]; }
          term(rho, dl, rec, m, 0, false, true);
        }
        FF_SCHED_FENCE();
      }
      if (has_mu) {
#pragma unroll
        for (int m = 0; m < NP; m++) {
          const double* rec = &s_rec[gg * RECS + (P + pbase + m) * RECW];
          term(rec, &u[m * D], rec, m, 0, false, false);
        }
      }
#pragma unroll
      for (int k = 0; k < MH; k++) out[1 + k] = du[k];
      // transpose-reduce the quadratic sources: the owner of coordinate c needs sum_i qv_(i, half(c))[c]
      if (ingrp) {
#pragma unroll
        for (int k = 0; k < MH; k++) s_qt[g * QTS + (i * 2 + h) * (MH + 1) + k] = qv[k];
      }
      FF_WG1_SYNC();
      double sumq = 0.0;
      if (owner) {
        const int lc = i - h * MH;
        for (int j = 0; j < M; j++) sumq += s_qt[gg * QTS + (j * 2 + h) * (MH + 1) + lc];
      }
      out[0] = owner ? vi : 0.0;
      out[MH + 1] = owner ? sumq + dvk : 0.0;
      out[MH + 2] = -ddiv;
      out[MH + 3] = -divv;
      out[MH + 4] = -(qdiv + (owner ? gdi * in[MH + 1] : 0.0));
      // ------------------------------------------------------------------ consume
      if (sv == -2) {
#pragma unroll
        for (int v = 0; v < NV; v++) c0[v] = out[v];
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double isc = wgt(v) * ff_rcp(fma(fabs(y[v]), rtol, atol));
          p0 = fma(y[v] * isc, y[v] * isc, p0);
          p1 = fma(c0[v] * isc, c0[v] * isc, p1);
        }
        const double d0 = sqrt(group_sum(p0) * (1.0 / NT));
        d1v = sqrt(group_sum(p1) * (1.0 / NT));
        h0v = S.h0(d0, d1v);
        s = -1;
        // every walker of the wave brings its own first step: no probe evaluation
        if (!ff_wave_or(&s_any, lane, (!S.done && !warm) ? 1 : 0)) {
          S.habs = fmin(hwarm, S.interval);
          S.plan();
          s = 1;
        }
      } else if (sv == -1) {
        double p2 = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double t = (out[v] - c0[v]) * wgt(v) * ff_rcp(fma(fabs(y[v]), rtol, atol));
          p2 = fma(t, t, p2);
        }
        const double d2 = sqrt(group_sum(p2) * (1.0 / NT)) / h0v;
        S.init_habs(h0v, d1v, d2);
        if (warm) S.habs = fmin(hwarm, S.interval);
        S.plan();
        s = 1;
      } else if (sv == 0) {
#pragma unroll
        for (int v = 0; v < NV; v++) c0[v] = out[v];
        s = 1;
      } else if (sv == 1) {
#pragma unroll
        for (int v = 0; v < NV; v++) c1[v] = out[v];
        s = 2;
      } else if (sv == 2) {
#pragma unroll
        for (int v = 0; v < NV; v++) c2[v] = out[v];
        s = 3;
      } else if (sv == 3) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double k0v = c0[v], k1v = c1[v], k2v = c2[v], k3v = out[v], yv = y[v];
          c0[v] = fma(hs, FF_A40 * k0v + FF_A41 * k1v + FF_A42 * k2v + FF_A43 * k3v, yv);
          c1[v] = fma(hs, FF_A50 * k0v + FF_A51 * k1v + FF_A52 * k2v + FF_A53 * k3v, yv);
          c2[v] = fma(hs, FF_B0 * k0v + FF_B2 * k2v + FF_B3 * k3v, yv);
          c3[v] = hs * (FF_E0 * k0v + FF_E2 * k2v + FF_E3 * k3v);
          FF_OPAQUE(c0[v]); FF_OPAQUE(c1[v]); FF_OPAQUE(c2[v]);      // (not sunk into the stages that use them)
        }
        s = 4;
      } else if (sv == 4) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
          c1[v] = fma(hs * FF_A54, out[v], c1[v]);
          c2[v] = fma(hs * FF_B4, out[v], c2[v]);
          c3[v] = fma(hs * FF_E4, out[v], c3[v]);
          FF_OPAQUE(c1[v]); FF_OPAQUE(c2[v]);
        }
        s = 5;
      } else if (sv == 5) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
          c2[v] = fma(hs * FF_B5, out[v], c2[v]);
          c3[v] = fma(hs * FF_E5, out[v], c3[v]);
          FF_OPAQUE(c2[v]);
        }
        s = 6;
      } else {
        double pe = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double e = fma(hs * FF_E6, out[v], c3[v]);
          const double t = e * wgt(v) * ff_rcp(fma(fmax(fabs(y[v]), fabs(in[v])), rtol, atol));
          pe = fma(t, t, pe);
        }
        const double err = sqrt(group_sum(pe) * (1.0 / NT));
        const bool was_active = !S.done;
        const bool acc = S.decide(err, A.max_steps);
        if (acc) hmax_acc = fmax(hmax_acc, fabs(hs));
        if (acc) {
#pragma unroll
          for (int v = 0; v < NV; v++) { y[v] = in[v]; c0[v] = out[v]; }
        }
        S.plan();
        const int any = ff_wave_or(&s_any, lane, S.done ? 0 : ((was_active && !acc) ? 3 : 1));
        s = (any & 2) ? 0 : 1;
        if (!any) return true;
      }
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
    // ---------------------------------------------------------------------- results: combine the per-lane partials
    const double dpart = y[MH + 2], delpart = y[MH + 3], lpart = y[MH + 4];
    const double delta = group_sum(delpart) * (1.0 / M);    // every direction's two lanes cover all pairs once
    if (ingrp) s_err[g][idx] = dpart;
    FF_WG1_SYNC();
    const double dD_i = s_err[gg][i] + s_err[gg][M + i];
    FF_WG1_SYNC();
    if (ingrp) s_err[g][idx] = lpart;
    FF_WG1_SYNC();
    const double L_i = s_err[gg][i] + s_err[gg][M + i];
    FF_WG1_SYNC();
    if (valid) {
      const double bad = S.fail ? __builtin_nan("") : 0.0;   // failed integration -> NaN results (see ff_ode_fwd_kernel)
      if (owner) { A.y_out[b * M + i] = y[0] + bad; A.kbar[b * M + i] = y[MH + 1]; }
#pragma unroll
      for (int k = 0; k < MH; k++) A.Jt[(b * M + i) * M + h * MH + k] = y[1 + k];
      if (h == 0) { A.dD[b * M + i] = dD_i; A.Lpart[b * M + i] = L_i; }
      if (idx == 0) {
        A.dl_out[b] = delta + bad;
        if (A.h_out) A.h_out[b] = hmax_acc > 0.0 ? hmax_acc : hwarm;
        if (A.wcost) A.wcost[b] = S.nacc + S.nrej;
        if (A.stats) { atomicAdd(&s_st[0], nev); atomicMax(&s_st[1], S.nacc); atomicAdd(&s_st[2], S.nrej); if (S.fail) atomicMax(&s_st[3], 1); }
      }
    }
    FF_WG1_SYNC();
  }
  if constexpr (TAB) { if (off_table) *A.evt = A.evt_id; }
  FF_WG1_SYNC();
  if (A.stats && lane == 0 && (s_st[0] || s_st[3])) {
    atomicAdd(&A.stats[0], s_st[0]);
    atomicMax(&A.stats[1], s_st[1]);
    atomicAdd(&A.stats[2], s_st[2]);
    if (s_st[3]) atomicMax(&A.stats[3], 1);
  }
}

#include "ff_eloc_rows.h"
#include "ff_eloc_mfma.h"

// ---------------------------------------------------------------------------------------------------
// Local-energy finish: Slater gradient/Hessian at z(t0) contracted with the sensitivities from the MODE-2 pass.
// With g0 = grad_z logp0, H0 = Hess_z logp0:
//   grad_i = g0 . u_i - dDelta_i
//   lap    = sum_i u_i^T H0 u_i + g0 . kbar - sum_i L_i
//   E_loc  = -lap/4 - |grad|^2/8 + V(x)            (src/VMC.py:49-55)
// Two launches.  (1) the Slater quantities of z(t0) into a table Q[walker][slot]: ff_eloc_slater_fixed_kernel (determinants up
// to 4 x 4 of one size: everything in registers, one lane per (walker, spin)) or ff_eloc_slater_rows_kernel (ff_ho3d.hip: sixteen
// lanes per determinant, any sizes up to 12).  (2) ff_eloc_contract_kernel, M = 2n lanes per walker as in the sensitivity kernels: the wave's G*M*M
// block of J^T is ONE contiguous read (the one-lane-per-walker version of this contraction fetched every cache line
// ~8 times: 730 MB per launch against 100 MB of sensitivities), each lane contracts its own direction, three LDS sums
// finish the walker.
// slots of Q: [0,M) g0 | [M, M+3n) S (particle-major) | then T_up (2 nup^2), T_dn (2 ndn^2) | last two: 2 log|det| per spin
// equal (or single) determinant sizes known at compile time: everything in registers (ff_slater_fixed)
// With the fused finish (ff_fwd_args::fin) the throughput kernel writes the local energies itself; what is left for the two kernels
// below are the walkers the routed pass sent to the one-walker-per-wave kernel (class >= heavy_class), whose sensitivities are in the
// workspace -- unless the table kernels raised the off-table event and the (unrouted, fused) direct kernel redid every walker.
struct ff_fin_filter {
  const int32_t* wclass;     // NULL: every walker
  int heavy_class;
  const double* evt;         // with wclass: skip everything if *evt == evt_id
  double evt_id;
  FF_D bool skip(int64_t b) const { return wclass != nullptr && (wclass[b] < heavy_class || *evt == evt_id); }
};
template <int NS>
__global__ void __launch_bounds__(128)
ff_eloc_slater_fixed_kernel(int64_t B, int nup, int ndn, const int* __restrict__ tab_up, const int* __restrict__ tab_dn,
                            const int* __restrict__ wstate, const double* __restrict__ z0, double* __restrict__ Q, ff_fin_filter flt) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = gid >> 1;
  const int sp = (int)(gid & 1);
  if (b >= B || flt.skip(b)) return;
  const int n = nup + ndn, M = 2 * n, st = wstate ? wstate[b] : 0;
  const int64_t nqs = M + 3 * n + 2 * (nup * nup + ndn * ndn) + 2;   // slots per walker
  const int lpq = M + 3 * n + 2 * (nup * nup + ndn * ndn) + sp;
  const int ns = sp ? ndn : nup, off = sp ? nup : 0;
  if (!ns) { Q[b * nqs + lpq] = 0.0; return; }      // ns is NS here
  double zl[2 * NS], T[2 * NS * NS], S[3 * NS];
#pragma unroll
  for (int k = 0; k < 2 * NS; k++) zl[k] = z0[b * M + 2 * off + k];
  double lp0;
  if constexpr (NS <= 3) {
    // up to 3 x 3: Hermite functions and both derivatives from one recurrence pass per coordinate, the inverse by the adjugate
    // (ff_slater_fixed -- Horner on the coefficient table by a per-lane index, Gauss-Jordan with a pivot search -- takes 55 us per
    // 131 072 determinants at NS = 3 and every register; this takes a tenth)
    const int* orb = (sp ? tab_dn : tab_up) + st * NS;
    int deg[2 * NS], md = 1;
#pragma unroll
    for (int j = 0; j < NS; j++) { ff_orb_decode(orb[j], deg[j], deg[NS + j]); }
#pragma unroll
    for (int j = 0; j < 2 * NS; j++) md = deg[j] > md ? deg[j] : md;
    double Dm[NS][NS], Di[NS][NS], gx[NS][NS], gy[NS][NS], hxx[NS][NS], hxy[NS][NS], hyy[NS][NS];
#pragma unroll
    for (int a = 0; a < NS; a++) {
      const double zx = zl[2 * a], zy = zl[2 * a + 1];
      double hx[NS], hx1[NS], hx2[NS], hy[NS], hy1[NS], hy2[NS];
      ff_herm_rec_d2<NS>(deg, zx, md, hx, hx1, hx2);
      ff_herm_rec_d2<NS>(deg + NS, zy, md, hy, hy1, hy2);
      const double gs = ff_gauss2d(zx, zy);
#pragma unroll
      for (int j = 0; j < NS; j++) {
        const double px1 = fma(-zx, hx[j], hx1[j]), py1 = fma(-zy, hy[j], hy1[j]);
        const double px2 = fma(fma(zx, zx, -1.0), hx[j], fma(-2.0 * zx, hx1[j], hx2[j]));
        const double py2 = fma(fma(zy, zy, -1.0), hy[j], fma(-2.0 * zy, hy1[j], hy2[j]));
        const double ex = gs * hx[j], ey = gs * hy[j];
        Dm[a][j] = ex * hy[j];
        gx[a][j] = gs * px1 * hy[j]; gy[a][j] = ex * py1;
        hxx[a][j] = gs * px2 * hy[j]; hxy[a][j] = gs * px1 * py1; hyy[a][j] = ex * py2;
      }
    }
    lp0 = log(fabs(ff_inv_small<NS>(Dm, Di)));
#pragma unroll
    for (int a = 0; a < NS; a++) {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int bq = 0; bq < NS; bq++) {
        double tx = 0.0, ty = 0.0;
#pragma unroll
        for (int j = 0; j < NS; j++) { tx = fma(gx[a][j], Di[j][bq], tx); ty = fma(gy[a][j], Di[j][bq], ty); }
        T[a * NS + bq] = tx; T[NS * NS + a * NS + bq] = ty;
      }
#pragma unroll
      for (int j = 0; j < NS; j++) { s0 = fma(hxx[a][j], Di[j][a], s0); s1 = fma(hxy[a][j], Di[j][a], s1); s2 = fma(hyy[a][j], Di[j][a], s2); }
      S[3 * a] = s0; S[3 * a + 1] = s1; S[3 * a + 2] = s2;
    }
  } else {
    lp0 = ff_slater_fixed<NS>((sp ? tab_dn : tab_up) + st * NS, zl, T, S);
  }
#pragma unroll
  for (int a = 0; a < NS; a++) {
    Q[b * nqs + (2 * (off + a))] = 2.0 * T[a * NS + a];
    Q[b * nqs + (2 * (off + a) + 1)] = 2.0 * T[NS * NS + a * NS + a];
#pragma unroll
    for (int k = 0; k < 3; k++) Q[b * nqs + (M + 3 * (off + a) + k)] = S[3 * a + k];
  }
  const int tq = M + 3 * n + (sp ? 2 * nup * nup : 0);
#pragma unroll
  for (int k = 0; k < 2 * NS * NS; k++) Q[b * nqs + (tq + k)] = T[k];
  Q[b * nqs + lpq] = 2.0 * lp0;
}

// dynamic LDS: [G*M*M J^T block | G*nq Slater slots | 64 x | 3*64 partial sums]  (10 KB at n = 6: many waves per CU)
static size_t ff_contract_lds_bytes(int nup, int ndn) {
  const int n = nup + ndn, M = 2 * n, G = FF_WAVE / M, nq = M + 3 * n + 2 * (nup * nup + ndn * ndn) + 2;
  return sizeof(double) * ((size_t)G * M * M + (size_t)G * nq + 4 * FF_WAVE);
}
__global__ void __launch_bounds__(FF_WAVE)
ff_eloc_contract_kernel(int64_t B, int nup, int ndn, double Zc, int use_ho, const double* __restrict__ x,
                        const double* __restrict__ Q, const double* __restrict__ Jt, const double* __restrict__ kbar,
                        const double* __restrict__ dD, const double* __restrict__ delta, const double* __restrict__ Lpart,
                        double* __restrict__ logp, double* __restrict__ grad, double* __restrict__ lap,
                        double* __restrict__ V, double* __restrict__ eloc, double* __restrict__ glogp0, ff_fin_filter flt) {
  FF_DYN_LDS(ff_fin_lds);
  const int n = nup + ndn, M = 2 * n, G = FF_WAVE / M;
  const int lane = threadIdx.x, g = lane / M, i = lane - g * M;
  const bool ingrp = g < G;
  const int tsz = 2 * (nup * nup + ndn * ndn), nq = M + 3 * n + tsz + 2;
  double* const s_u = ff_fin_lds;                            // [g][i][k] = dz_k/dx_i
  double* const s_q = s_u + G * M * M;                       // [g][slot]
  double* const s_x = s_q + G * nq;
  double (*const s_red)[FF_WAVE] = (double (*)[FF_WAVE])(s_x + FF_WAVE);
  const int64_t ngroups = (B + G - 1) / G;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t b0 = grp * G, b = b0 + g;
    const bool valid = ingrp && b < B && !flt.skip(b);
    const int nw = (int)((B - b0) < G ? (B - b0) : G);       // walkers of this wave
    if (flt.wclass && !__ballot(valid)) continue;            // (filtered finish: most groups hold no walker of the heavy route)
    __syncthreads();
    {   // the wave's J^T block: one contiguous span, 16-byte loads (M*M is even, so the span is whole pairs)
      const double2* __restrict__ J2 = reinterpret_cast<const double2*>(Jt + b0 * M * M);
      double2* u2 = reinterpret_cast<double2*>(s_u);
      for (int e = lane; e < nw * M * M / 2; e += FF_WAVE) u2[e] = J2[e];
    }
    for (int e = lane; e < nw * nq; e += FF_WAVE) s_q[e] = Q[b0 * nq + e];   // the wave's walkers: one contiguous span
    if (valid) s_x[lane] = x[b * M + i];
    __syncthreads();
    double gi = 0.0, lap_i = 0.0, v_i = 0.0;
    if (valid) {
      const double* u = s_u + (g * M + i) * M;
      const double* qw = s_q + g * nq;
      const double* g0 = qw;
      for (int k = 0; k < M; k++) gi = fma(g0[k], u[k], gi);
      gi -= dD[b * M + i];
      double hq = 0.0;
      for (int sp = 0; sp < 2; sp++) {
        const int ns = sp ? ndn : nup, off = sp ? nup : 0;
        if (!ns) continue;
        const double* Tx = qw + M + 3 * n + (sp ? 2 * nup * nup : 0);
        const double* Ty = Tx + ns * ns;
        double q = 0.0;
        for (int a = 0; a < ns; a++) {
          const double ux = u[2 * (off + a)], uy = u[2 * (off + a) + 1];
          const double* Sa = qw + M + 3 * (off + a);
          q += ux * ux * Sa[0] + 2.0 * ux * uy * Sa[1] + uy * uy * Sa[2];
          for (int c = 0; c < ns; c++) {
            const double vx = u[2 * (off + c)], vy = u[2 * (off + c) + 1];
            const double Wac = ux * Tx[a * ns + c] + uy * Ty[a * ns + c];
            const double Wca = vx * Tx[c * ns + a] + vy * Ty[c * ns + a];
            q -= Wac * Wca;
          }
        }
        hq += 2.0 * q;
      }
      lap_i = hq - Lpart[b * M + i] + g0[i] * kbar[b * M + i];
      // potential: the lane of particle a's x-coordinate takes a's trap term and its pairs with the later particles
      if ((i & 1) == 0) {
        const int a = i >> 1;
        const double* xs = s_x + g * M;
        const double xa = xs[2 * a], ya = xs[2 * a + 1];
        double pair = 0.0;
        for (int c = a + 1; c < n; c++) {
          const double dx = xa - xs[2 * c], dy = ya - xs[2 * c + 1];
          pair += Zc / sqrt(dx * dx + dy * dy);
        }
        v_i = pair + (use_ho ? 0.5 * (xa * xa + ya * ya) : 0.0);
      }
      if (grad) grad[b * M + i] = gi;
      if (glogp0) glogp0[b * M + i] = g0[i];
    }
    s_red[0][lane] = gi * gi; s_red[1][lane] = lap_i; s_red[2][lane] = v_i;
    __syncthreads();
    if (valid && i == 0) {
      double g2 = 0.0, lapv = 0.0, Vv = 0.0;
      for (int k = 0; k < M; k++) { g2 += s_red[0][g * M + k]; lapv += s_red[1][g * M + k]; Vv += s_red[2][g * M + k]; }
      if (logp) logp[b] = (s_q[g * nq + nq - 2] + s_q[g * nq + nq - 1]) - delta[b];
      if (lap) lap[b] = lapv;
      if (V) V[b] = Vv;
      if (eloc) eloc[b] = -0.25 * lapv - 0.125 * g2 + Vv;
    }
  }
}

// =================================================================================================
extern void ff_set_error(const char* msg);
extern int ff_slater_rows_launch(void* stream, int d, int64_t B, int nup, int ndn, const int* tab_up, const int* tab_dn, const int* wstate,
                                 const double* z0, double* Q);
#define FF_CHECK(cond, code, msg) do { if (!(cond)) { ff_set_error(msg); return code; } } while (0)
#define FF_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { ff_set_error(hipGetErrorString(e_)); return FF_ELAUNCH; } } while (0)

#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>
static constexpr int64_t FF_GRID_CAP = 1 << 20;      // without a work queue: one workgroup per walker group, up to this many

// With a radial table: the table kernel, then the direct-evaluation kernel as its (normally idle) fallback -- it returns
// in its first instructions unless the table kernel left this launch's id in the event slot.  Without: direct only.
// persistent grid of the queue mode: one wave per SIMD
static int64_t fwd_queue_blocks() {
  static int64_t n = 0;
  if (n == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    n = 4 * (int64_t)cus;
  }
  return n;
}

// Routing of the local-energy pass by cost class.  A launch cannot end before its longest chain of steps has: a walker with a
// particle passing the origin takes 20-30 steps of 7 dependent evaluations, 3.4-5.5 us each for a wave of the several-walkers-
// per-wave kernels -- as long as the other 99.6 % of the walkers need the whole GPU.  The one-walker-per-wave kernel of ff_wide.hip
// runs one evaluation in 2.6 us.  With cost classes at hand (ff_ode.walker_class: the sweeps pass the flow pass's) the walkers of
// class >= ff_ode.heavy_class therefore go to that kernel, launched first on the caller's stream (one wave per
// walker; its 292 registers keep the SIMD to itself), and everyone else to the throughput kernel on a side stream, both joined
// before anything else runs.  Which kernel integrates a walker depends on the walker's own class only, never on the batch it is
// part of or on the order of work.  The heavy walkers are also among the ones whose E_loc error is largest (the embedded error estimate
// underrates the kink they pass) and their chain has slack now: they are integrated at ff_ode.heavy_tol (default 0.3) x (rtol, atol).
// The default threshold by system size (round 5, bench.py at 20 and at 200 steps, thresholds 12 .. 20 and none): routing pays while
// the pass is bound by its longest chain and costs once it is bound by the work -- every routed walker holds a SIMD that two waves
// of the throughput kernel would share (292 + 253 registers do not fit one file), and the fork and the join add 12 us each end.
// Up to 4 particles (5 in d = 2) the chain rules: class >= 12 (0.4-0.6 % of a batch; 3 particles: pass 0.507 -> 0.463 ms, none:
// 0.507).  At 12 coordinates the work does -- and more so as the flow trains: class >= 16 (0.04-0.1 %): pass 0.806 -> 0.796 ms on the
// benchmark's weights (none: 0.833), 1.08 -> 1.02 ms 200 iterations later (none: 1.00), trained flows 1.33 -> 1.25 (none: 1.20).
#define FF_HEAVY_CLASS_DEFAULT(M_) ((M_) >= 12 ? 16 : 12)
#define FF_HEAVY_TOL_DEFAULT 0.3
#define FF_SUM_WEIGHT_DEFAULT 4.0
// One side stream + two events per DEVICE, created on the first routed call on that device and shared by every host thread:
// the whole fork / launch / join sequence of a routed pass runs under g_side_mutex (two threads driving the same device then
// take turns on the side stream; their own streams stay independent).  ff_shutdown() releases everything.
struct ff_side_lane { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
static std::mutex g_side_mutex;
static std::vector<ff_side_lane> g_side_lanes;
static void ff_side_destroy(ff_side_lane& l) {
  if (l.fork) (void)hipEventDestroy(l.fork);
  if (l.join) (void)hipEventDestroy(l.join);
  if (l.stream) (void)hipStreamDestroy(l.stream);
  l = ff_side_lane();
}
// caller holds g_side_mutex
static ff_side_lane* ff_side() {
  int dev = 0, ndev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
  if ((size_t)dev >= g_side_lanes.size()) {
    if (hipGetDeviceCount(&ndev) != hipSuccess || dev >= ndev) return nullptr;
    g_side_lanes.resize((size_t)ndev);
  }
  ff_side_lane& l = g_side_lanes[(size_t)dev];
  if (!l.stream) {
    if (hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) != hipSuccess) { l = ff_side_lane(); return nullptr; }
    if (hipEventCreateWithFlags(&l.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&l.join, hipEventDisableTiming) != hipSuccess) { ff_side_destroy(l); return nullptr; }
  }
  return &l;
}
extern "C" {
static int eloc_finish_impl(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                            const int32_t* walker_state, double Z, int use_ho, const double* x, const void* workspace,
                            double* logp, double* grad, double* lap, double* V, double* eloc, double* z_out, double* dlogp_out,
                            double* glogp0_out, ff_fin_filter flt);
}
enum { FF_ROUTE_NONE = 0, FF_ROUTE_DONE = 1, FF_ROUTE_FAILED = 2 };
// What the local-energy dispatch of THIS host thread did, for ff_eloc to read back right after it: did the kernel it chose take the
// fused finish (ff_fwd_args::fin), and was the pass routed (then the walkers of class >= heavy_class still need the finish kernels).
struct ff_eloc_feedback { bool fused, routed; const double* evt; double evt_id; };
static thread_local ff_eloc_feedback t_eloc_fb = {false, false, nullptr, 0.0};
// launch_table(stream, args): the table kernel of the throughput family.
// FF_ROUTE_DONE: both launches are enqueued and joined (the caller's fallback launch then redoes EVERY walker should it have to run:
// heavy_mode stays 0 in its arguments).  FF_ROUTE_NONE: nothing was launched -- no classes, routing switched off, no side stream, or
// the heavy launch was refused -- and the caller launches its table kernel for every walker.  FF_ROUTE_FAILED: the throughput
// launch or the join failed after the heavy kernel was enqueued; both streams have been drained and the caller returns FF_ELAUNCH
// (the outputs of this call are undefined).
// throughput_fuses: the throughput kernel takes the fused finish -- only then may the heavy kernel finish ITS walkers itself (bit 1 of
// fin.on): otherwise the caller's unfiltered finish kernels run over every walker and need the heavy walkers' sensitivities in the workspace.
template <class F>
static int launch_routed(void* stream, int n, int d, const ff_fwd_args& a, F launch_table, bool throughput_fuses) {
  if (!(a.evt && a.wclass && a.heavy_class > 0 && n * d <= 12 && ff_wide_supported(n, d))) return FF_ROUTE_NONE;
  std::lock_guard<std::mutex> lock(g_side_mutex);
  ff_side_lane* side = ff_side();
  if (!side || hipEventRecord(side->fork, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(side->stream, side->fork, 0) != hipSuccess) {
    (void)hipGetLastError();
    return FF_ROUTE_NONE;
  }
  ff_fwd_args h = a, l = a;
  h.queue = nullptr; h.heavy_mode = 1;      // grid-stride over every walker, the light ones skipped
  if (!throughput_fuses) h.fin.on = 0;
#ifndef FF_WIDE_NO_FIN
  // Beside a throughput kernel that finishes its walkers itself the heavy kernel does too (its FIN instantiation): two launches less
  // -- the filtered finish kernels walked the whole batch for a few dozen walkers -- and no sensitivities in the workspace.  Round 4
  // measured that as slower (pass 0.809 -> 0.832 ms) and left it to ff_ode.compact_finish; what it had measured was the heavy kernel's
  // residency (342 instead of 292 registers on a thousand SIMDs for ~100 us: see ff_wide_next_heavy) -- since round 5: 0.795 either way.
  else h.fin.on |= 2;
#endif
  if (a.heavy_tol > 0.0) { h.rtol *= a.heavy_tol; h.atol *= a.heavy_tol; }
  l.heavy_mode = 2;
  // placed first: a persistent grid takes every register file it finds
#ifndef FF_DIAG_NO_HEAVY      // (timing diagnostic: the throughput kernel without its neighbour; the heavy walkers' outputs are then garbage)
  if (ff_wide_eloc_heavy(stream, n, d, h, 1024) != FF_OK) { (void)hipGetLastError(); return FF_ROUTE_NONE; }
#endif
  launch_table(side->stream, l);
  // With the fused finish the throughput kernel writes its walkers' local energies itself; the heavy route's walkers leave their
  // sensitivities in the workspace and take the two finish kernels, filtered by class -- enqueued HERE, behind the heavy kernel on
  // the caller's stream and in front of the join: the heavy kernel ends well before the throughput kernel does (0.53 against
  // 0.82 ms at config 2), so their 46 us run in its shadow.  (Should the table kernels raise the off-table event, the unrouted
  // fused direct kernel behind the join redoes every walker and overwrites these outputs.)
  if (throughput_fuses && !(h.fin.on & 2) && a.fin.workspace) {      // (with bit 1 the heavy kernel finishes its walkers itself)
    const ff_fwd_args::ff_fin_args& f = a.fin;
    if (eloc_finish_impl(stream, a.B, f.nup, f.ndn, f.tab_up, f.tab_dn, f.wstate, f.Z, f.use_ho, a.y_in, f.workspace, f.logp, f.grad, f.lap,
                         f.V, f.eloc, nullptr, nullptr, f.glogp0, ff_fin_filter{a.wclass, a.heavy_class, a.evt, a.evt_id}) != FF_OK)
      (void)hipGetLastError();      // (falls through to the join; the failure resurfaces in the caller's launch check)
  }
  const bool ok = hipGetLastError() == hipSuccess && hipEventRecord(side->join, side->stream) == hipSuccess &&
                  hipStreamWaitEvent((hipStream_t)stream, side->join, 0) == hipSuccess;
  if (!ok) {
    // the heavy kernel is running (or queued) on `stream`, possibly the throughput kernel on the side stream: let both finish
    // before the caller sees the error, so that nothing of this call still writes when it returns
    (void)hipStreamSynchronize(side->stream);
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipGetLastError();
    ff_set_error("ff_eloc: the routed local-energy launch failed (throughput kernel or stream join)");
    return FF_ROUTE_FAILED;
  }
  return FF_ROUTE_DONE;
}

template <int N, int D, int MODE>
static int launch_fwd(void* stream, const ff_fwd_args& a) {
  constexpr int G = ff_geom<N, D>::G;
  int64_t ngroups = (a.B + G - 1) / G;
  const int64_t cap = a.queue ? fwd_queue_blocks() : FF_GRID_CAP;   // without a queue: one workgroup per walker group
  unsigned grid = (unsigned)(ngroups < cap ? ngroups : cap);
  auto table = [&](void* st, const ff_fwd_args& aa) { FF_LAUNCH((ff_ode_fwd_kernel<N, D, MODE, true>), grid, FF_WAVE, st, aa); };
  int routed = FF_ROUTE_NONE;
  if constexpr (MODE == 2 && N <= 3) routed = launch_routed(stream, N, D, a, table, false);
  if (routed == FF_ROUTE_FAILED) return FF_ELAUNCH;
  if (routed == FF_ROUTE_NONE && a.evt) table(stream, a);
  // behind a table kernel the direct kernel is a fallback that almost always finds nothing to do: a grid-stride launch of two waves per
  // SIMD instead of one workgroup per walker group (13 108 workgroups at config 2: 6 us just to start and retire them)
  const unsigned grid_fb = a.evt && grid > 2048u ? 2048u : grid;
  FF_LAUNCH((ff_ode_fwd_kernel<N, D, MODE, false>), grid_fb, FF_WAVE, stream, a);
  return FF_OK;
}

// n >= 8 uses the two-lanes-per-direction local-energy kernel (measured, 32768 walkers: n = 8 6.5 -> 4.5 ms,
// n = 10 48 -> 9.3 ms, n = 12 95 -> 14.7 ms); FF_NO_SPLIT=1 forces the one-lane-per-direction kernel (A/B testing)
template <int N, int D>
static void launch_split(void* stream, const ff_fwd_args& a) {
  constexpr int G = FF_WAVE / (2 * N * D);
  int64_t ngroups = (a.B + G - 1) / G;
  const int64_t cap = a.queue ? fwd_queue_blocks() : FF_GRID_CAP;
  const unsigned grid = (unsigned)(ngroups < cap ? ngroups : cap);
  if (a.evt) FF_LAUNCH((ff_eloc_split_kernel<N, D, true>), grid, FF_WAVE, stream, a);
  FF_LAUNCH((ff_eloc_split_kernel<N, D, false>), grid, FF_WAVE, stream, a);
}

// Row-layout local-energy kernel (ff_eloc_rows.h): SPLIT lanes per row chosen so that a walker group fills the wave and
// the workgroup's LDS stays under 40 KB (four single-wave workgroups per CU)
template <int N, int D, int SPLIT>
static void launch_rows(void* stream, const ff_fwd_args& a) {
  constexpr int G = FF_WAVE / (N * D * SPLIT) > 16 ? 16 : FF_WAVE / (N * D * SPLIT);
  int64_t ngroups = (a.B + G - 1) / G;
  const int64_t cap = a.queue ? fwd_queue_blocks() : FF_GRID_CAP;
  const unsigned grid = (unsigned)(ngroups < cap ? ngroups : cap);
  if (a.evt) FF_LAUNCH((ff_eloc_rows_kernel<N, D, SPLIT, true>), grid, FF_WAVE, stream, a);
  FF_LAUNCH((ff_eloc_rows_kernel<N, D, SPLIT, false>), grid, FF_WAVE, stream, a);
}

#ifndef FF_MFMA_WPS
#define FF_MFMA_WPS 2   // waves per SIMD the matrix-core kernel is compiled for (A/B knob)
#endif
// Matrix-core local-energy kernel (ff_eloc_mfma.h): four walkers per wave, M = n d <= 12, two waves per SIMD
template <int N, int D>
static int launch_mfma(void* stream, const ff_fwd_args& a) {
  const int64_t ngroups = (a.B + 3) / 4;
  const int64_t cap = a.queue ? FF_MFMA_WPS * fwd_queue_blocks() : FF_GRID_CAP;
  auto table = [&](void* st, const ff_fwd_args& aa) {
    FF_LAUNCH((ff_eloc_mfma_kernel<N, D, true, FF_MFMA_WPS>), (unsigned)(ngroups < cap ? ngroups : cap), FF_WAVE, st, aa);
  };
  const int routed = launch_routed(stream, N, D, a, table, (a.fin.on & 1) && N % 2 == 0 && D == 2);
  if (routed == FF_ROUTE_FAILED) return FF_ELAUNCH;
  if (routed == FF_ROUTE_NONE && a.evt) table(stream, a);
  t_eloc_fb.fused = (a.fin.on & 1) && N % 2 == 0 && D == 2;      // (what the kernel's epilogue tests)
  t_eloc_fb.routed = routed == FF_ROUTE_DONE;
  t_eloc_fb.evt = a.evt; t_eloc_fb.evt_id = a.evt_id;
  const int64_t cap1 = a.queue ? fwd_queue_blocks() : FF_GRID_CAP;
  FF_LAUNCH((ff_eloc_mfma_kernel<N, D, false, 1>), (unsigned)(ngroups < cap1 ? ngroups : cap1), FF_WAVE, stream, a);
  return FF_OK;
}

static std::atomic<uint64_t> g_evt_counter{1};

template <int MODE>
static int dispatch_fwd(void* stream, int n, int d, const ff_fwd_args& a_in) {
  ff_fwd_args a = a_in;
  a.evt = nullptr;
  a.evt_id = 0.0;
  if (a.net.radial_table) {
    const uint64_t id = g_evt_counter.fetch_add(1);
    a.evt = const_cast<double*>(a.net.radial_table) + FF_TAB_EVT0 + (id % FF_TAB_NEVT);
    a.evt_id = (double)(id & ((1ull << 52) - 1)) + 1.0;
  }
  // Local-energy pass: three kernels compute it (tests/test_hostsim.py::test_three_local_energy_kernels_agree);
  // FF_ELOC_KERNEL = auto (default) | mfma | rows | columns forces one where it is instantiated.  auto takes the fastest
  // measured on MI355X (tools/probes/eloc_ab.py): the column sweep up to 8 particles, the row layout from 9 on (and for
  // every particle number the column sweep is not instantiated for).
  auto wide = [&]() -> int {      // the one-walker-per-workgroup family; its local-energy kernels take the fused finish (bit 1 of fin.on)
    const int st_ = ff_wide_dispatch_fwd(MODE, stream, n, d, a);
    if (MODE == 2 && st_ == FF_OK) { t_eloc_fb.fused = (a.fin.on & 2) != 0; t_eloc_fb.routed = false; }
    return st_;
  };
  if (ff_wide_forced() && ff_wide_supported(n, d)) return wide();   // FF_WIDE=1: A/B and parity testing
  static const int eloc_kind = [] {
    const char* e = getenv("FF_ELOC_KERNEL");
    return !e ? 0 : (!strcmp(e, "mfma") ? 1 : (!strcmp(e, "rows") ? 2 : (!strcmp(e, "columns") ? 3 : (!strcmp(e, "wide") ? 4 : 0))));
  }();
  // one walker per workgroup, both products on v_mfma_f64_16x16x4 (ff_wide.hip): measured faster than the row layout from
  // 11 particles on (tools/probes/wide_c5.py; 16 384 walkers: 11 particles 2.12 against 2.68 ms, 12 particles 2.25 against 2.91; 10 particles 2.04 against 1.88)
  constexpr int wide_from = 11;
  if (MODE == 2 && d == 2 && (eloc_kind == 4 || (eloc_kind == 0 && n >= wide_from)) && ff_wide_supported(n, d))
    return wide();
#ifndef FF_MFMA_FROM
#define FF_MFMA_FROM 4      // (2-3 particles tie with the column sweep and keep it: DESIGN.md 3g; the host simulator builds with 99)
#endif
  constexpr int mfma_from = FF_MFMA_FROM;
  if (MODE == 2 && (eloc_kind == 1 || (eloc_kind == 0 && d == 2 && n >= mfma_from && n <= 6))) {
#define FF_MF(N_, D_) if (n == N_ && d == D_) { const int s_ = launch_mfma<N_, D_>(stream, a); if (s_) return s_; FF_LAUNCH_CHECK(); return FF_OK; }
    FF_MF(6, 2) FF_MF(2, 2) FF_MF(3, 2) FF_MF(4, 2) FF_MF(5, 2)
#undef FF_MF
  }
  const bool no_columns = n == 1 || n == 7 || n == 9 || n == 11 || d == 3;
  if (MODE == 2 && (eloc_kind == 2 || no_columns || (eloc_kind == 0 && n >= 9))) {
#define FF_RW(N_, D_, S_) if (n == N_ && d == D_) { launch_rows<N_, D_, S_>(stream, a); FF_LAUNCH_CHECK(); return FF_OK; }
    FF_RW(6, 2, 1) FF_RW(2, 2, 1) FF_RW(3, 2, 1) FF_RW(4, 2, 1) FF_RW(5, 2, 1) FF_RW(7, 2, 2) FF_RW(8, 2, 2) FF_RW(9, 2, 3)
    FF_RW(10, 2, 3) FF_RW(11, 2, 2) FF_RW(12, 2, 2) FF_RW(1, 2, 1)
    FF_RW(2, 3, 1) FF_RW(3, 3, 1) FF_RW(4, 3, 1)      // three dimensions (small systems; finish: ff_eloc_finish3d)
#undef FF_RW
  }
  if (MODE == 2 && d == 2) {
#define FF_SP(N_) if (n == N_) { launch_split<N_, 2>(stream, a); FF_LAUNCH_CHECK(); return FF_OK; }
    FF_SP(8) FF_SP(10) FF_SP(12)
#undef FF_SP
  }
#define FF_ND(N_, D_) if (n == N_ && d == D_) { const int s_ = launch_fwd<N_, D_, MODE>(stream, a); if (s_) return s_; FF_LAUNCH_CHECK(); return FF_OK; }
  FF_ND(6, 2) FF_ND(3, 2) FF_ND(12, 2) FF_ND(2, 2) FF_ND(4, 2) FF_ND(5, 2) FF_ND(8, 2) FF_ND(10, 2)
  if constexpr (MODE != 2) { FF_ND(1, 2) FF_ND(7, 2) FF_ND(9, 2) FF_ND(11, 2) FF_ND(2, 3) FF_ND(3, 3) FF_ND(4, 3) }   // (their local-energy pass is the row-layout kernel above)
#undef FF_ND
  // everything else: one walker per workgroup (ff_wide.hip: n <= 24, n d <= 60)
  return wide();
}

static int check_common(int64_t B, int n, int d, const ff_net* net, const ff_ode* ode) {
  FF_CHECK(B >= 0 && n > 0 && d > 0 && net && ode, FF_EINVAL, "ff_cnf: bad argument");
  FF_CHECK(net->He > 0 && net->ew1 && net->eb1 && net->ew2 && (net->Hm == 0 || (net->mw1 && net->mb1 && net->mw2)), FF_EINVAL,
           "ff_cnf: bad net");
  FF_CHECK(net->He <= FF_HMAX && net->Hm <= FF_HMAX, FF_EUNSUPPORTED, "ff_cnf: hidden width > 256");
  FF_CHECK(ode->rtol > 0 && ode->atol > 0, FF_EINVAL, "ff_cnf: tolerances must be positive");
  return FF_OK;
}

extern "C" {

int ff_shutdown(void) {
  std::lock_guard<std::mutex> lock(g_side_mutex);
  int dev0 = 0;
  const bool have = hipGetDevice(&dev0) == hipSuccess;
  for (size_t k = 0; k < g_side_lanes.size(); k++) {
    if (!g_side_lanes[k].stream) continue;
    if (hipSetDevice((int)k) == hipSuccess) (void)hipStreamSynchronize(g_side_lanes[k].stream);
    ff_side_destroy(g_side_lanes[k]);
  }
  if (have) (void)hipSetDevice(dev0);
  (void)hipGetLastError();
  return FF_OK;
}

size_t ff_radial_table_bytes(void) { return sizeof(double) * (size_t)FF_TAB_DOUBLES; }

int ff_radial_table_build(void* stream, const ff_net* net, double* table) {
  FF_CHECK(net && table, FF_EINVAL, "ff_radial_table_build: null pointer");
  FF_CHECK(net->He > 0 && net->ew1 && net->eb1 && net->ew2 && (net->Hm == 0 || (net->mw1 && net->mb1 && net->mw2)), FF_EINVAL,
           "ff_radial_table_build: bad net");
  FF_LAUNCH(ff_table_kernel, FF_TAB_GRID, 128, stream, *net, table);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_cnf_generate(void* stream, int64_t B, int n, int d, const ff_net* net, const ff_ode* ode, const double* z,
                    double* x_out, int32_t* stats) {
  int st = check_common(B, n, d, net, ode);
  if (st) return st;
  FF_CHECK(z && x_out, FF_EINVAL, "ff_cnf_generate: null pointer");
  if (B == 0) return FF_OK;
  ff_fwd_args a = {};
  a.B = B; a.net = *net; a.ta = ode->t0; a.tb = ode->t1; a.rtol = ode->rtol; a.atol = ode->atol;
  a.max_steps = ode->max_steps > 0 ? ode->max_steps : 10000;
  a.wcost = ode->walker_cost; a.order = ode->walker_order;
  a.h_init = ode->walker_h_init; a.h_scale = ode->walker_h_uniform ? -fabs(ode->walker_h_scale) : fabs(ode->walker_h_scale); a.h_out = ode->walker_h_out; a.h_equal = ode->walker_h_equal;
  a.y_in = z; a.y_out = x_out; a.stats = stats;
  return dispatch_fwd<0>(stream, n, d, a);
}

int ff_cnf_delta_logp(void* stream, int64_t B, int n, int d, const ff_net* net, const ff_ode* ode, const double* x,
                      double* z_out, double* dlogp_out, int32_t* stats) {
  int st = check_common(B, n, d, net, ode);
  if (st) return st;
  FF_CHECK(x && z_out && dlogp_out, FF_EINVAL, "ff_cnf_delta_logp: null pointer");
  if (B == 0) return FF_OK;
  ff_fwd_args a = {};
  a.B = B; a.net = *net; a.ta = ode->t1; a.tb = ode->t0; a.rtol = ode->rtol; a.atol = ode->atol;
  a.max_steps = ode->max_steps > 0 ? ode->max_steps : 10000;
  a.wcost = ode->walker_cost; a.order = ode->walker_order;
  a.h_init = ode->walker_h_init; a.h_scale = ode->walker_h_uniform ? -fabs(ode->walker_h_scale) : fabs(ode->walker_h_scale); a.h_out = ode->walker_h_out; a.h_equal = ode->walker_h_equal;
  a.y_in = x; a.y_out = z_out; a.dl_out = dlogp_out; a.stats = stats;
  return dispatch_fwd<1>(stream, n, d, a);
}

// sensitivities (z0, Jt, kbar, dD, Lpart, Delta) + the Slater table of the finish: layout in ff_eloc_ws.h
size_t ff_eloc_workspace_bytes(int64_t B, int n, int d) {
  return sizeof(double) * ff_eloc_ws_doubles(B, (size_t)n, (size_t)d);
}
size_t ff_eloc_nd_workspace_bytes(int64_t B, int n, int d, int compact_finish) {
  return sizeof(double) * ff_eloc_ws_doubles(B, (size_t)n, (size_t)d, compact_finish && ff_eloc_ws_compact((size_t)n, (size_t)d));
}

static int eloc_sensitivities_impl(void* stream, int64_t B, int n, int d, const ff_net* net, const ff_ode* ode, const double* x,
                                   void* workspace, int32_t* stats, const ff_fwd_args::ff_fin_args* fin);
}
// the two work counters of a launch (table kernel, direct fallback) back to zero
__global__ void ff_queue_reset_kernel(unsigned long long* q) { if (threadIdx.x < 2) q[threadIdx.x] = 0ULL; }
extern "C" {

/* pass 1 of ff_eloc: the fused sensitivity integration (results stay in `workspace`) */
int ff_eloc_sensitivities(void* stream, int64_t B, int n, int d, const ff_net* net, const ff_ode* ode, const double* x,
                          void* workspace, int32_t* stats) {
  return eloc_sensitivities_impl(stream, B, n, d, net, ode, x, workspace, stats, nullptr);
}

static int eloc_sensitivities_impl(void* stream, int64_t B, int n, int d, const ff_net* net, const ff_ode* ode, const double* x,
                                   void* workspace, int32_t* stats, const ff_fwd_args::ff_fin_args* fin) {
  t_eloc_fb = {false, false, nullptr, 0.0};
  int st = check_common(B, n, d, net, ode);
  if (st) return st;
  FF_CHECK(x && workspace, FF_EINVAL, "ff_eloc_sensitivities: null pointer");
  if (B == 0) return FF_OK;
  ff_eloc_ws w = ff_eloc_carve(workspace, B, (size_t)n, (size_t)d, fin != nullptr && (fin->on & 2) && ff_eloc_ws_compact((size_t)n, (size_t)d));
  ff_fwd_args a = {};
  a.B = B; a.net = *net; a.ta = ode->t1; a.tb = ode->t0; a.rtol = ode->rtol; a.atol = ode->atol;
  a.max_steps = ode->max_steps > 0 ? ode->max_steps : 10000;
  a.wcost = ode->walker_cost; a.order = ode->walker_order;
  a.h_init = ode->walker_h_init; a.h_scale = ode->walker_h_uniform ? -fabs(ode->walker_h_scale) : fabs(ode->walker_h_scale); a.h_out = ode->walker_h_out; a.h_equal = ode->walker_h_equal;
  a.y_in = x; a.y_out = w.z0; a.dl_out = w.dl; a.Jt = w.Jt; a.kbar = w.kbar; a.dD = w.dD; a.Lpart = w.Lp; a.stats = stats;
  a.wclass = ode->walker_class; a.sens_class = ode->sens_tol_class;
  a.sens_w = ode->sens_tol > 1.0 ? 1.0 / ode->sens_tol : 1.0;
  a.h_scale_loose = ode->walker_h_scale_loose > 0.0 ? ode->walker_h_scale_loose : fabs(ode->walker_h_scale);
  a.heavy_class = ode->heavy_class == 0 ? FF_HEAVY_CLASS_DEFAULT(n * d) : ode->heavy_class;      // (< 0: no routing)
  a.heavy_tol = ode->heavy_tol > 0.0 ? ode->heavy_tol : FF_HEAVY_TOL_DEFAULT;
  a.sum_w = ode->sum_weight > 0.0 ? ode->sum_weight : FF_SUM_WEIGHT_DEFAULT;
  if (fin) a.fin = *fin;
  FF_LAUNCH(ff_queue_reset_kernel, 1, FF_WAVE, stream, w.queue);      // (a kernel of our own, not hipMemsetAsync: 1 us instead of a 5 us blit + 8 us of bubble)
  a.queue = w.queue;
  return dispatch_fwd<2>(stream, n, d, a);
}

/* pass 2 of ff_eloc: Slater gradient/Hessian contraction, potentials, E_loc */
int ff_eloc_finish(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                   const int32_t* walker_state, double Z, int use_ho, const double* x, const void* workspace,
                   double* logp, double* grad, double* lap, double* V, double* eloc, double* z_out, double* dlogp_out,
                   double* glogp0_out) {
  return eloc_finish_impl(stream, B, nup, ndn, tab_up, tab_dn, walker_state, Z, use_ho, x, workspace, logp, grad, lap, V, eloc, z_out,
                          dlogp_out, glogp0_out, ff_fin_filter{nullptr, 0, nullptr, 0.0});
}

static int eloc_finish_impl(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                            const int32_t* walker_state, double Z, int use_ho, const double* x, const void* workspace,
                            double* logp, double* grad, double* lap, double* V, double* eloc, double* z_out, double* dlogp_out,
                            double* glogp0_out, ff_fin_filter flt) {
  const int n = nup + ndn;
  FF_CHECK(B >= 0 && nup >= 0 && ndn >= 0 && n > 0 && x && workspace, FF_EINVAL, "ff_eloc_finish: bad argument");
  FF_CHECK((nup == 0 || tab_up) && (ndn == 0 || tab_dn), FF_EINVAL, "ff_eloc_finish: null orbital table");
  FF_CHECK(nup <= FF_MAX_NS && ndn <= FF_MAX_NS, FF_EUNSUPPORTED, "ff_eloc_finish: determinant larger than FF_MAX_NS");
  if (B == 0) return FF_OK;
  const size_t M = (size_t)n * 2;
  ff_eloc_ws w = ff_eloc_carve((void*)workspace, B, (size_t)n, 2);
  FF_CHECK(2 * n <= FF_WAVE, FF_EUNSUPPORTED, "ff_eloc_finish: n*d > 64");
  {
    const int nsf = (nup == ndn || ndn == 0) ? nup : (nup == 0 ? ndn : 0);   // one determinant size for both spin species
    const unsigned sgrid = (unsigned)((2 * B + 127) / 128);
#define FF_SF(NS_) case NS_: FF_LAUNCH((ff_eloc_slater_fixed_kernel<NS_>), sgrid, 128, stream, B, nup, ndn, tab_up, tab_dn, walker_state, (const double*)w.z0, w.Q, flt); break;
    switch (nsf) {
      FF_SF(1) FF_SF(2) FF_SF(3) FF_SF(4)
      default:      // larger or unequal determinants: sixteen lanes per determinant (ff_ho3d.hip)
        if (ff_slater_rows_launch(stream, 2, B, nup, ndn, tab_up, tab_dn, walker_state, (const double*)w.z0, w.Q) != FF_OK) return FF_ELAUNCH;
    }
#undef FF_SF
  }
  FF_LAUNCH_CHECK();
  {
    const int Gf = FF_WAVE / (2 * n);
    const int64_t ng = (B + Gf - 1) / Gf;
    // (the filtered launch of the heavy route skips 99.6 % of the walker groups: a small grid striding over them instead of one
    // workgroup per group, whose starting and retiring alone took 26 us beside the throughput kernel)
    const int64_t gcap = flt.wclass ? 1024 : 32768;
    FF_LAUNCH_LDS(ff_eloc_contract_kernel, (unsigned)(ng < gcap ? ng : gcap), FF_WAVE, ff_contract_lds_bytes(nup, ndn), stream, B, nup, ndn, Z, use_ho, x,
              (const double*)w.Q, (const double*)w.Jt, (const double*)w.kbar, (const double*)w.dD, (const double*)w.dl,
              (const double*)w.Lp, logp, grad, lap, V, eloc, glogp0_out, flt);
  }
  FF_LAUNCH_CHECK();
  if (z_out && hipMemcpyAsync(z_out, w.z0, sizeof(double) * (size_t)B * M, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return FF_ELAUNCH;
  if (dlogp_out && hipMemcpyAsync(dlogp_out, w.dl, sizeof(double) * (size_t)B, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return FF_ELAUNCH;
  return FF_OK;
}

extern "C" int ff_eloc_finish3d(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                                const int32_t* walker_state, double Z, int use_ho, const double* x, const void* workspace,
                                double* logp, double* grad, double* lap, double* V, double* eloc, double* z_out, double* dlogp_out,
                                double* glogp0_out);      // ff_ho3d.hip

int ff_eloc_nd(void* stream, int64_t B, int nup, int ndn, int d, const int32_t* tab_up, const int32_t* tab_dn,
               const int32_t* walker_state, const ff_net* net, const ff_ode* ode, double Z, int use_ho, const double* x,
               double* logp, double* grad, double* lap, double* V, double* eloc, double* z_out, double* dlogp_out,
               double* glogp0_out, void* workspace, int32_t* stats) {
  FF_CHECK(nup >= 0 && ndn >= 0 && nup + ndn > 0 && (d == 2 || d == 3), FF_EINVAL, "ff_eloc: bad particle numbers or dimension");
  FF_CHECK((nup == 0 || tab_up) && (ndn == 0 || tab_dn), FF_EINVAL, "ff_eloc: null orbital table");
  FF_CHECK(nup <= FF_MAX_NS && ndn <= FF_MAX_NS, FF_EUNSUPPORTED, "ff_eloc: determinant larger than FF_MAX_NS");
  // Offer the fused finish: a sensitivity kernel that implements it writes logp, grad, lap, V, E_loc and grad_z logp0 from its
  // epilogue, and J^T (8 M^2 bytes per walker) never leaves the chip -- the matrix-core kernel for nup = ndown <= 3 in d = 2 (bit 0:
  // config 2); on request (ff_ode::compact_finish: it costs time, include/fermiflow.h) the one-walker-per-workgroup kernels for every
  // shape they serve (bit 1: configs 3-4, everything beyond 12 particles, the heavy-walker route).
  ff_fwd_args::ff_fin_args fin = {};
  const int n = nup + ndn;
  const bool compact = ode && ode->compact_finish && ff_eloc_ws_compact((size_t)n, (size_t)d);
  fin.on = ((d == 2 && nup == ndn && nup >= 1 && nup <= 3) ? 1 : 0) | ((ode && ode->compact_finish) ? 2 : 0);
  fin.nup = nup; fin.ndn = ndn; fin.use_ho = use_ho; fin.tab_up = tab_up; fin.tab_dn = tab_dn; fin.wstate = walker_state; fin.Z = Z;
  fin.logp = logp; fin.grad = grad; fin.lap = lap; fin.V = V; fin.eloc = eloc; fin.glogp0 = glogp0_out; fin.workspace = workspace;
  int st = eloc_sensitivities_impl(stream, B, n, d, net, ode, x, workspace, stats, &fin);
  if (st) return st;
  const ff_eloc_feedback fb = t_eloc_fb;
  if (B == 0) return FF_OK;
  if (!fb.fused) {
    FF_CHECK(!compact, FF_EUNSUPPORTED, "ff_eloc: compact_finish, but the kernel of this shape has no fused finish");
    if (d == 3)
      return ff_eloc_finish3d(stream, B, nup, ndn, tab_up, tab_dn, walker_state, Z, use_ho, x, workspace, logp, grad, lap, V, eloc, z_out,
                              dlogp_out, glogp0_out);
    return ff_eloc_finish(stream, B, nup, ndn, tab_up, tab_dn, walker_state, Z, use_ho, x, workspace, logp, grad, lap, V, eloc,
                          z_out, dlogp_out, glogp0_out);
  }
  // (routed pass: the walkers of the heavy route were finished inside launch_routed -- by their own kernel's epilogue)
  const size_t M = (size_t)n * d;
  ff_eloc_ws w = ff_eloc_carve(workspace, B, (size_t)n, (size_t)d, compact);
  if (z_out && hipMemcpyAsync(z_out, w.z0, sizeof(double) * (size_t)B * M, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return FF_ELAUNCH;
  if (dlogp_out && hipMemcpyAsync(dlogp_out, w.dl, sizeof(double) * (size_t)B, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return FF_ELAUNCH;
  return FF_OK;
}

int ff_eloc(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
            const int32_t* walker_state, const ff_net* net, const ff_ode* ode, double Z, int use_ho, const double* x,
            double* logp, double* grad, double* lap, double* V, double* eloc, double* z_out, double* dlogp_out,
            double* glogp0_out, void* workspace, int32_t* stats) {
  return ff_eloc_nd(stream, B, nup, ndn, 2, tab_up, tab_dn, walker_state, net, ode, Z, use_ho, x, logp, grad, lap, V, eloc, z_out, dlogp_out,
                    glogp0_out, workspace, stats);
}

}  // extern "C"
