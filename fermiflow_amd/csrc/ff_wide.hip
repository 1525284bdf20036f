// ff_wide.hip -- the forward CNF integrations for walkers that do not fit one wave's layouts: ONE WALKER PER WORKGROUP.
//
//   CNF.generate / CNF.delta_logp (src/flow.py:42-55)           ff_wide_flow_kernel<D, MODE, TAB>    64 lanes per walker
//   local-energy sensitivities (src/VMC.py:46-49 through
//   src/utils.py:40-65, by forward sensitivities as ff_eloc)     ff_wide_eloc_kernel<D, T, TAB>       64 T lanes per walker
//
// Particle numbers are run-time values here (n <= 24, M = n d <= 60): BASELINE.json configs[4] (nup = ndown = 10 in a 3-D
// trap, M = 60) and every 2-D system beyond 12 particles (the reference is shape-generic: src/equivariant_funs.py:17-102).
//
// The local-energy kernel integrates the system of ff_eloc_rows.h,
//     z' = v(z)      J' = A J  (A = dv/dz, J = dz/dx)      kbar' = A kbar + sum_i D2v[u_i, u_i]
//     Delta' = -div v      (grad Delta)' = -J^T g  (g = grad_z div v)      (lap Delta)' = -(sum_i D2div[u_i, u_i] + g . kbar)
// with the quadratic sources taken from S = J J^T (O(1) work per radius), so that per right-hand side two dense
// MP x MP x MP products remain -- J' = A J and S = J J^T -- and both run on the matrix cores (v_mfma_f64_16x16x4_f64):
//   * MP = 16 T >= M + 4 is the padded size, the workgroup has T waves, wave w owns COLUMN block w of J: lane (g, c)
//     (g = lane / 16, c = lane % 16) keeps J[16 K + 4 v + g][16 w + c] for the row tiles K = 0..T-1 and v = 0..3 -- the
//     instruction's own C/D layout, so J' = A J lands where the Runge-Kutta arithmetic wants it, and the stage input is the
//     B operand as it stands (k-slot g of step (K, v) is row 16 K + 4 v + g);
//   * row M of J is grad_x Delta: row M of A holds -g, so (A J)[M][i] = -sum_p g_p J[p][i] comes out of the same product;
//   * A (assembled from the D x D pair blocks B = eta I + (eta'/r) rho rho^T that the radius lanes leave in per-radius
//     records) and the stage J are read as MFMA operands from LDS (row stride MP + 2 doubles: conflict-free for the
//     (row = lane % 16, column = k0 + lane / 16) operand pattern); S is written over the stage J once every wave has read it.
// Per right-hand side (four workgroup barriers): publish | radii, table rows requested, S = J J^T under the fetch, heads ->
// records | S takes the stage J's place in LDS; own-row sums, A assembled | J' = A J; radius lanes contract their terms with S |
// second-order sums, Dormand-Prince bookkeeping (ff_dp5.h).  fp64 matrix and vector instructions have the same peak on this part (and do not overlap,
// tools/probes/mfma_f64.hip), so the matrix cores buy instruction slots and operand traffic, not flops: the kernel is
// bound by 2 * T * ceil(M/4) matrix instructions of 64 cycles per wave and evaluation.
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include "ff_common.h"
#include "ff_ode.h"
#include "ff_dp5.h"
#include "ff_radial.h"
#include "ff_fwd_args.h"
#include "ff_slater_rows.h"

// FF_STAMPS: diagnostic build only (make variant NAME=stamps EXTRA=-DFF_STAMPS; tools/probes/wide_c5.py): s_memtime shares
// of the phases of the local-energy kernel's right-hand side, added into stats[8..] as 64-bit counters by wave 0.
#ifdef FF_STAMPS
#define FF_STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += t_ - stamp_prev; stamp_prev = t_; } while (0)
#else
#define FF_STAMP(i) do { } while (0)
#endif

#define FF_WIDE_NMAX 24
#define FF_WIDE_MMAX 60
#define FF_WIDE_RMAX (FF_WIDE_NMAX * (FF_WIDE_NMAX + 1) / 2)   // pairs + one-body radii

// radius q of a walker with n particles (pairs a < b in a-major order, then the one-body radii) -> a | b << 5 | q << 10
// (b = 31: one-body); -1 beyond nrad
FF_D int ff_wide_radius_id(int n, int q, int nrad) {
  if (q >= nrad) return -1;
  const int P = n * (n - 1) / 2;
  if (q >= P) return (q - P) | (31 << 5) | (q << 10);
  int a = 0, off = 0;
  while (q >= off + (n - 1 - a)) { off += n - 1 - a; a++; }
  return a | ((a + 1 + q - off) << 5) | (q << 10);
}

// index of the radius between particle a and partner j (j = a: the one-body radius)
FF_D int ff_wide_partner(int n, int P, int a, int j) {
  const int lo = j < a ? j : a, hi = j < a ? a : j;
  return (j == a) ? P + a : ff_pair_index(n, lo, hi);
}

// sum of a per-lane partial over a workgroup of NTHR lanes, identical on every lane: 16 column sums, then their sum
template <int NTHR>
FF_D double ff_wide_sum(double* s_red, double* s_red2, int tid, double part) {
  s_red[tid] = part;
  __syncthreads();
  if (tid < 16) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < NTHR / 16; k++) t += s_red[tid + 16 * k];
    s_red2[tid] = t;
  }
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) t += s_red2[k];
  __syncthreads();
  return t;
}

// =====================================================================================================================
// CNF.generate (MODE 0) / CNF.delta_logp (MODE 1): one wave per walker; lane l owns coordinates l and l + 64 and the
// radii l, l + 64, ... (for MODE 1 also the share of Delta that its radii contribute: a plain quadrature).
template <int D, int MODE, bool TAB>
__global__ void __launch_bounds__(FF_WAVE)
ff_wide_flow_kernel(ff_fwd_args A, int n) {
  constexpr int NH = MODE == 0 ? 1 : 2;
  constexpr int NP = 1;                                  // coordinate slots per lane (M <= 64)
  constexpr int NQ = (FF_WIDE_RMAX + FF_WAVE - 1) / FF_WAVE;
  constexpr int NV = NP + (MODE == 1 ? 1 : 0);
  __shared__ ff_wtab s_w[TAB ? 1 : 2][TAB ? 1 : FF_HPAD];
  __shared__ double s_e2[TAB ? 1 : 64];
  __shared__ double s_z[FF_WAVE], s_f0[FF_WIDE_RMAX], s_red[FF_WAVE], s_red2[16];
  __shared__ int s_st[4];

  const int lane = threadIdx.x;
  const int M = n * D, P = n * (n - 1) / 2;
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
  if (lane < 4) s_st[lane] = 0;
  __syncthreads();
  const int He = A.net.He, Hm = A.net.Hm;
  const bool has_mu = Hm > 0;
  const int nrad = has_mu ? P + n : P;
  const double tab_inv_h = TAB ? rtab[0] : 0.0, tab_h = TAB ? rtab[1] : 0.0;
  int rq_id[NQ];
#pragma unroll
  for (int qk = 0; qk < NQ; qk++) rq_id[qk] = ff_wide_radius_id(n, lane + qk * FF_WAVE, nrad);
  const bool own = lane < M;
  const int ai = own ? lane / D : 0, ci = own ? lane % D : 0;

  for (int64_t bq = blockIdx.x; bq < A.B; bq += gridDim.x) {
    const int64_t b = ff_opt_load(A.order, true, bq, A.y_in, (int32_t)bq);
    double y[NV], c0[NV], c1[NV], c2[NV], c3[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) { y[v] = 0.0; c0[v] = 0.0; c1[v] = 0.0; c2[v] = 0.0; c3[v] = 0.0; }
    if (own) y[0] = A.y_in[b * M + lane];
    ff_stepper S;
    S.begin(A.ta, A.tb, true);
    ff_dp5_ctl C;
    C.rtol = A.rtol; C.atol = A.atol; C.nt_inv = 1.0 / (double)(M + (MODE == 1 ? 1 : 0)); C.max_steps = A.max_steps;
    C.hwarm = ff_open_step(ff_opt_load(A.h_init, true, A.h_scale < 0.0 ? 0 : b, A.y_in, 0.0) * fabs(A.h_scale), A.ta, A.tb, A.h_equal);
    if (!(C.hwarm > 0.0)) C.hwarm = 0.0;
    C.h0v = 0.0; C.d1v = 0.0; C.hmax_acc = 0.0;
    const double hwarm0 = C.hwarm;
    double rmin = 1e300;
    int s = -2, nev = 0;
    auto wgt = [&](int v) -> double { return 1.0; };
    auto gsum = [&](double part) -> double { return ff_wide_sum<FF_WAVE>(s_red, s_red2, lane, part); };

    // one evaluation, instantiated per stage for the table kernels; returns true when the walker has finished
    constexpr bool STATIC_STAGES = TAB;
    auto evaluate = [&](auto stage_tag) -> bool {
      constexpr int SG = decltype(stage_tag)::value;
      if constexpr (SG == FF_STAGE_DYN && STATIC_STAGES) FF_ASSUME(s <= 0);
      const int sv = SG == FF_STAGE_DYN ? s : SG;
      double gy, g0, g1, g2;
      ff_dp5_coeffs(sv, S.h, C.h0v * S.dir, gy, g0, g1, g2);
      __syncthreads();
      if (own) s_z[lane] = fma(g2, c2[0], fma(g1, c1[0], fma(g0, c0[0], gy * y[0])));
      __syncthreads();
      // ---------------------------------------------------------------- radius phase
      double dsum = 0.0;
#pragma unroll
      for (int qk = 0; qk < NQ; qk++) {
        const int id = rq_id[qk];
        if (id < 0) break;
        const int a = id & 31, bb0 = (id >> 5) & 31, pr = id >> 10;
        const bool pair = bb0 != 31;
        double r2 = 0.0;
#pragma unroll
        for (int c = 0; c < D; c++) {
          const double rho = s_z[a * D + c] - (pair ? s_z[bb0 * D + c] : 0.0);
          r2 = fma(rho, rho, r2);
        }
        double r, ri, hd[NH];
        ff_sqrt_rcp(r2, r, ri);
        rmin = fmin(rmin, r);
        if constexpr (TAB) {
          if (!ff_heads_table<NH>(rtab, tab_inv_h, tab_h, pair ? 0 : 1, r, hd)) {
            off_table = true;
#pragma unroll
            for (int m = 0; m < NH; m++) hd[m] = 0.0;
          }
        } else {
          ff_heads<NH, true>(s_w[pair ? 0 : 1], s_e2, pair ? He : Hm, r, hd);
        }
        s_f0[pr] = hd[0];
        if constexpr (MODE == 1) dsum = fma(pair ? 2.0 : 1.0, fma(hd[NH - 1], r, D * hd[0]), dsum);
      }
      __syncthreads();
      nev++;
      // ---------------------------------------------------------------- component phase
      double out[NV];
      {
        double vi = 0.0;
        if (own) {
          const double zc = s_z[lane];
          for (int j = 0; j < n; j++) {
            if (j == ai) continue;
            vi = fma(s_f0[ff_wide_partner(n, P, ai, j)], zc - s_z[j * D + ci], vi);
          }
          if (has_mu) vi = fma(s_f0[P + ai], zc, vi);
        }
        out[0] = vi;
        if constexpr (MODE == 1) out[1] = -dsum;
      }
      s = ff_dp5_consume<NV>(sv, S, C, y, c0, c1, c2, c3, out, wgt, gsum);
      return s == 99;
    };
    if constexpr (STATIC_STAGES) {
#pragma unroll 1
      for (;;) {
        bool fin = false;
#pragma unroll 1
        while (s <= 0 && !fin) fin = evaluate(ff_stage_c<FF_STAGE_DYN>{});
        if (fin) break;
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
    // -------------------------------------------------------------------- results
    const bool failed = S.fail != 0;
    const double bad = failed ? __builtin_nan("") : 0.0;
    if (own) A.y_out[b * M + lane] = y[0] + bad;
    double delta = 0.0;
    if constexpr (MODE == 1) delta = gsum(y[NV - 1]);
    if (A.wcost) {
      s_red[lane] = rmin;
      __syncthreads();
    }
    if (lane == 0) {
      if constexpr (MODE == 1) A.dl_out[b] = delta + bad;
      if (A.h_out) A.h_out[b] = C.hmax_acc > 0.0 ? C.hmax_acc : hwarm0;
      if (A.wcost) {
        double m = 1e300;
        for (int k = 0; k < FF_WAVE; k++) m = fmin(m, s_red[k]);
        A.wcost[b] = ff_cost_class(S.nacc + S.nrej, m);
      }
      if (A.stats) { atomicAdd(&s_st[0], nev); atomicMax(&s_st[1], S.nacc); atomicAdd(&s_st[2], S.nrej); if (failed) atomicMax(&s_st[3], 1); }
    }
    __syncthreads();
  }
  if constexpr (TAB) { if (off_table) *A.evt = A.evt_id; }
  __syncthreads();
  if (A.stats && lane == 0 && (s_st[0] || s_st[3])) {
    atomicAdd(&A.stats[0], s_st[0]);
    atomicMax(&A.stats[1], s_st[1]);
    atomicAdd(&A.stats[2], s_st[2]);
    if (s_st[3]) atomicMax(&A.stats[3], 1);
  }
}

// =====================================================================================================================
// Local-energy sensitivities: one walker per workgroup of T waves, both dense products on the matrix cores (file header).
// Lane roles (NTHR = 64 T lanes):
//   * every lane: its 4 T elements of J (rows 0..M-1: dz/dx; row M: grad_x Delta);
//   * "row lanes" tid = 4 p + s, p < M: the four lanes of coordinate p split the partners of p's particle (b = s, s + 4, ...):
//     own-row sums (v, D_v[kbar], grad div, the diagonal block of A) as quad reductions (DPP), the off-diagonal blocks of row p
//     of A written on the way; lane s = 0 owns z_p and kbar_p;
//   * "radius lanes" tid < nrad (NQ slots): one radius each in R1 / R2, and the shares of Delta / lap_x Delta of its radii.
// Lane state (NV = 4 T + 4 doubles): [0, 4T) J, [4T] z_p, [4T+1] kbar_p, [4T+2] / [4T+3] the Delta / lap Delta shares.
// The error accumulator of the Dormand-Prince step lives in lane-private LDS columns for T >= 3 (registers: the four other
// vectors, the stage J and two sets of MFMA accumulators already fill the file).
template <class TE, int NV, int NTHR, bool IN_LDS>
struct ff_wide_vec;
template <class TE, int NV, int NTHR>
struct ff_wide_vec<TE, NV, NTHR, false> {
  static constexpr bool in_lds = false;
  TE r[NV];
  FF_D ff_wide_vec(TE*, int) {}
  FF_D TE& operator[](int v) { return r[v]; }
  FF_D const TE& operator[](int v) const { return r[v]; }
};
template <class TE, int NV, int NTHR>
struct ff_wide_vec<TE, NV, NTHR, true> {
  static constexpr bool in_lds = true;
  TE* col;
  FF_D ff_wide_vec(TE* base, int tid) : col(base + tid) {}
  FF_D TE& operator[](int v) { return col[v * NTHR]; }
  FF_D const TE& operator[](int v) const { return col[v * NTHR]; }
};

// The matrix instruction per element type of J.  fp64: v_mfma_f64_16x16x4_f64, register v of lane (g, c) holds row 4 v + g of a
// tile; fp32 (the single-precision sensitivity path of BASELINE configs[4]): v_mfma_f32_16x16x4_f32 -- twice the rate -- whose
// register v holds row 4 g + v (measured: tools/probes/wide_probe.hip, profiles/r03_wide_probe_mfma_layouts.txt).
template <class TJ> struct ff_wide_mma;
template <> struct ff_wide_mma<double> {
  typedef ff_d4 acc_t;
  static FF_D int row(int K, int v, int g) { return 16 * K + 4 * v + g; }
  static FF_D int row_lane(int g) { return g; }                       // row = row_lane(g) + row_step(K, v)
  static FF_D constexpr int row_step(int K, int v) { return 16 * K + 4 * v; }
  static FF_D acc_t mma(double a, double b, acc_t c) { return ff_mfma16(a, b, c); }
};
template <> struct ff_wide_mma<float> {
  typedef ff_f4 acc_t;
  static FF_D int row(int K, int v, int g) { return 16 * K + 4 * g + v; }
  static FF_D int row_lane(int g) { return 4 * g; }
  static FF_D constexpr int row_step(int K, int v) { return 16 * K + v; }
  static FF_D acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};

// sum over the four lanes of a quad (lanes 4q .. 4q+3), result on all four
FF_D double ff_quad_sum(double v) {
  v += ff_swap1(v);
  v += ff_swap2(v);
  return v;
}

// Heavy route (one wave per workgroup, grid-stride over the schedule): the wave looks at this workgroup's next 64 entries AT ONCE and
// returns the first one at or behind bq whose walker is of class >= heavy_class (or something >= B).  Walking the entries one by
// one -- two dependent loads each, and 99.6 % of them light -- kept every one of the 1024 workgroups resident for ~100 us with a
// register file to itself (292 registers: no room beside it for a wave of the throughput kernel, launched right behind): the whole
// pass started that much later, 1.00 -> 1.10 ms once it is long enough not to hide it behind its longest chain
// (tools/probes/heavy_neighbour.py, pass_timeline.py).
FF_D int64_t ff_wide_next_heavy(const ff_fwd_args& A, int64_t bq, int lane) {
  while (bq < A.B) {
    const int64_t q = bq + (int64_t)lane * gridDim.x;
    bool hv = false;
    if (q < A.B) hv = A.wclass[A.order ? (int64_t)A.order[q] : q] >= A.heavy_class;
    const unsigned long long m = ff_wave_ballot(hv);
    if (m) return bq + (int64_t)__builtin_ctzll(m) * gridDim.x;
    bq += (int64_t)FF_WAVE * gridDim.x;
  }
  return bq;
}

#ifndef FF_WIDE_T1_WAVES
#define FF_WIDE_T1_WAVES 1
#endif
#ifndef FF_WIDE_C3_FROM
#define FF_WIDE_C3_FROM 2       // waves per walker from which the fp64 kernel keeps its error accumulator in lane-private LDS columns (A/B knob)
#endif
#ifndef FF_WIDE_C3_F32
#define FF_WIDE_C3_F32 0        // ... and the fp32-sensitivity instantiations too (A/B knob)
#endif
#ifndef FF_WIDE_Y_FROM
#define FF_WIDE_Y_FROM 99       // waves per walker from which (up to two) the fp64 kernel keeps the J part of y in LDS as well (A/B knob)
#endif
#ifndef FF_WIDE_T2_WAVES
#define FF_WIDE_T2_WAVES 1      // workgroups per SIMD pair the two-wave instantiations are compiled for (A/B knob: 2 = at most 256 registers)
#endif
// FIN: with the fused finish (ff_ode::compact_finish) as the epilogue of every walker.  A template parameter, not a run-time branch: the
// epilogue's registers and LDS would otherwise weigh on the instantiations that run without it -- the T = 1 kernel of the heavy-walker
// route went from 292 to 380 registers and config 2's pass from 0.809 to 0.832 ms with the branch merely compiled in.
template <int D, int T, bool TAB, class TJ, bool FIN>
__global__ void __launch_bounds__(FF_WAVE * T, T == 1 ? FF_WIDE_T1_WAVES : (T == 2 ? FF_WIDE_T2_WAVES : 1))
ff_wide_eloc_kernel(ff_fwd_args A, int n) {
  typedef ff_wide_mma<TJ> MMA;
  constexpr bool F32 = sizeof(TJ) == 4;
  constexpr int MP = 16 * T, NTHR = FF_WAVE * T, JS = F32 ? MP + 1 : MP + 2;     // (row stride of the LDS operand images)
  constexpr int NVJ = 4 * T, NVS = 4, IZ = 0, IK = 1, IDL = 2, ILP = 3;           // J elements; the four fp64 scalars of a lane
  constexpr int NH = 4;
  constexpr int NCAP = (MP - 4) / D > FF_WIDE_NMAX ? FF_WIDE_NMAX : (MP - 4) / D;   // particles this instantiation can hold
  constexpr int RCAP = NCAP * (NCAP + 1) / 2;
  constexpr int NQ = (RCAP + NTHR - 1) / NTHR;
  constexpr int NPK = (NCAP + 3) / 4;                  // partners per row lane
  constexpr bool C3_LDS = T >= FF_WIDE_C3_FROM && TAB && (!F32 || FF_WIDE_C3_F32);       // (the direct-evaluation variant needs the LDS for its weight tables)
  // record of a radius: written by R1: [0,D) rho  [D] f0 = eta  [D+1] eta'/r  [D+2] gq = c phi'/r  [D+3, 2D+3) D_v[kbar] part
  //   [2D+4] 1/r^2  [2D+5] eta''  [2D+6] c phi''      (c = 2 for pairs, 1 for one-body radii; phi = eta' r + D eta)
  // written by R2: [D+3, 2D+3) the second-order source of kbar (the R1 entry is dead by then)
  constexpr int RW = (2 * D + 8) | 1;     // odd: the row lanes read the same field of many records at a time (an even stride puts them on a few bank groups)
  constexpr int QF0 = D, QF1 = D + 1, QGQ = D + 2, QPW = D + 3, QRI2 = 2 * D + 4, QF2 = 2 * D + 5, QBC = 2 * D + 6;

  __shared__ ff_wtab s_w[TAB ? 1 : 2][TAB ? 1 : FF_HPAD];
  __shared__ double s_e2[TAB ? 1 : 64];
  __shared__ __attribute__((aligned(16))) TJ s_J[MP * JS];   // stage J, [p][i]; then S = J J^T
  __shared__ __attribute__((aligned(16))) TJ s_A[MP * JS];   // A = dv/dz with row M = -grad div
  __shared__ __attribute__((aligned(16))) double s_rec[(RCAP + 1) * RW];   // + one record that stays zero
  __shared__ double s_z[MP], s_kb[MP], s_red[NTHR], s_red2[16];
  constexpr bool Y_LDS = T >= FF_WIDE_Y_FROM && T <= 2 && TAB && !F32;      // the J part of y as well (two waves per walker: LDS to spare)
  __shared__ TJ s_c3[C3_LDS ? NVJ * NTHR : 1];
  __shared__ TJ s_yJ[Y_LDS ? NVJ * NTHR : 1];
  __shared__ double s_c3s[C3_LDS ? NVS * NTHR : 1];
  __shared__ int s_st[4];
  __shared__ long long s_next;

  const int tid = threadIdx.x, w = tid / FF_WAVE, l = tid % FF_WAVE, lg = l / 16, lc = l % 16;
  const int M = n * D, P = n * (n - 1) / 2;
  const double* __restrict__ rtab = A.net.radial_table;
  if constexpr (TAB) {
    if (rtab[3] != 0.0) {
      if (tid == 0 && blockIdx.x == 0) *A.evt = A.evt_id;
      return;
    }
  } else {
    if (A.evt && *A.evt != A.evt_id) return;
    if (tid < FF_WAVE) { ff_fill_exp2_table(s_e2, tid); ff_load_weights(s_w, A.net, tid); }
  }
  // heavy route: a workgroup with no heavy walker among its entries leaves before it has set anything up
  int64_t bq_first = blockIdx.x;
  if constexpr (T == 1) {
    if (A.heavy_mode == 1 && !A.queue) {
      bq_first = ff_wide_next_heavy(A, bq_first, tid);
      if (bq_first >= A.B) return;
    }
  }
  bool off_table = false;
  if (tid < 4) s_st[tid] = 0;
  for (int e = tid; e < MP * JS; e += NTHR) { s_J[e] = (TJ)0; s_A[e] = (TJ)0; }
  for (int e = tid; e < (RCAP + 1) * RW; e += NTHR) s_rec[e] = 0.0;
  if (tid < MP) { s_z[tid] = 0.0; s_kb[tid] = 0.0; }
  __syncthreads();
  const int He = A.net.He, Hm = A.net.Hm;
  const bool has_mu = Hm > 0;
  const int nrad = has_mu ? P + n : P;
  const double tab_inv_h = TAB ? rtab[0] : 0.0, tab_h = TAB ? rtab[1] : 0.0;
  int rq_id[NQ];
#pragma unroll
  for (int qk = 0; qk < NQ; qk++) rq_id[qk] = ff_wide_radius_id(n, tid + qk * NTHR, nrad);
  // row lanes: coordinate p = tid / 4 of particle ra, partners rs, rs + 4, ...; packed per partner:
  // record offset << 3 | (partner < ra) << 2 | (partner == ra) << 1 | valid
  const int rp = tid >> 2, rs = tid & 3;
  const bool rowlane = rp < M;
  const bool own = rowlane && rs == 0;
  const int ra = rowlane ? rp / D : 0, rc = rowlane ? rp % D : 0;
  // per partner slot (branch-free: an absent partner reads the all-zero record behind the last one and "writes" -0.0 where the
  // diagonal block is stored afterwards): record offset | sign bit, and the destination of the off-diagonal block row in A
  int prec[NPK], pdst[NPK];
#pragma unroll
  for (int k = 0; k < NPK; k++) {
    const int bpart = rs + 4 * k;
    const bool valid = rowlane && bpart < n && (bpart != ra || has_mu);
    prec[k] = ((valid ? ff_wide_partner(n, P, ra, bpart) : RCAP) * RW) << 1 | ((valid && bpart < ra) ? 1 : 0);
    // (lanes that own no row -- rp >= M -- park their -0.0 in the padding rows of A, which nobody reads as anything but zero.  They
    // used to write at rp * JS: for rp = M that is A[M][0..D-1], the (grad Delta)' entries of particle 0 which lanes 3, 7 (, 11) store
    // in the same phase -- in program order within one wave, hence harmless in the one-wave kernel this code grew from, but a RACE across
    // waves: now and then the zero won, that evaluation's grad Delta source lost a component and the step was rejected.  Found in
    // round 4 by evaluating every right-hand side three times and comparing (-DFF_WIDE_SELFCHECK).)
    // (each such lane its own D entries of the padding rows M + 1 ..: 4 (MP - M) D <= (MP - M - 1) JS for every shape served, so that
    // not even equal values meet at one address and a ThreadSanitizer run of the host simulator stays silent)
    pdst[k] = rowlane ? rp * JS + ((valid && bpart != ra) ? bpart : ra) * D : (M + 1) * JS + (tid - 4 * M) * D;
  }
#ifdef FF_STAMPS
  unsigned long long stamp_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev = __builtin_amdgcn_s_memtime();
#endif

  bool degrees_known = false;      // (fused finish: the orbitals' Hermite degrees are in LDS)
  for (int64_t bq = bq_first;; bq += gridDim.x) {
    if (A.queue) {   // persistent grid: next walker from the launch's work counter (the order is by schedule key: cost class + 4 x planned steps, costliest first)
      __syncthreads();
      if (tid == 0) s_next = (long long)atomicAdd(A.queue + (TAB ? 0 : 1), 1ULL);
      __syncthreads();
      bq = s_next;
    }
    if constexpr (T == 1) {
      if (A.heavy_mode == 1 && !A.queue) bq = ff_wide_next_heavy(A, bq, tid);
    }
    if (bq >= A.B) break;
    const int64_t b = ff_opt_load(A.order, true, bq, A.y_in, (int32_t)bq);
    if (A.heavy_mode) {      // routing by cost class (workgroup-uniform): this launch takes one side of the threshold only
      const bool heavy = A.wclass[b] >= A.heavy_class;
      if (heavy != (A.heavy_mode == 1)) continue;
    }
    TJ c0J[NVJ], c1J[NVJ], c2J[NVJ];
    ff_wide_vec<TJ, NVJ, NTHR, Y_LDS> yJ(s_yJ, tid);
    ff_wide_vec<TJ, NVJ, NTHR, C3_LDS> c3J(s_c3, tid);
    double y[NVS], c0[NVS], c1[NVS], c2[NVS];
    ff_wide_vec<double, NVS, NTHR, C3_LDS> c3(s_c3s, tid);
#pragma unroll
    for (int v = 0; v < NVS; v++) { y[v] = 0.0; c0[v] = 0.0; c1[v] = 0.0; c2[v] = 0.0; c3[v] = 0.0; }
    y[IZ] = ff_opt_load(A.y_in, own, b * M + rp, A.y_in, 0.0);
#pragma unroll
    for (int K = 0; K < T; K++)
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int p = MMA::row(K, v, lg), i = 16 * w + lc;
        yJ[4 * K + v] = (p == i && p < M) ? (TJ)1 : (TJ)0;
        c0J[4 * K + v] = (TJ)0; c1J[4 * K + v] = (TJ)0; c2J[4 * K + v] = (TJ)0; c3J[4 * K + v] = (TJ)0;
      }
    ff_stepper S;
    S.begin(A.ta, A.tb, true);
    const bool loose = ff_opt_load(A.wclass, true, b, A.y_in, (int32_t)0x7fffffff) <= A.sens_class;
    ff_dp5_ctl C;
    C.rtol = A.rtol; C.atol = A.atol; C.nt_inv = 1.0 / ((double)M * (M + 4) + 1.0); C.max_steps = A.max_steps;
    C.hwarm = ff_opt_load(A.h_init, true, A.h_scale < 0.0 ? 0 : b, A.y_in, 0.0) * (loose ? A.h_scale_loose : fabs(A.h_scale));
    if (!(C.hwarm > 0.0)) C.hwarm = 0.0;
    C.h0v = 0.0; C.d1v = 0.0; C.hmax_acc = 0.0;
    const double hwarm0 = C.hwarm;
    const double sens_w = loose ? A.sens_w : 1.0;
    int s = -2, nev = 0;
    auto wgt = [&](int v) -> double { return v == IZ ? 1.0 : sens_w; };      // (the J elements all weigh sens_w)
    auto gsum = [&](double part) -> double { return ff_wide_sum<NTHR>(s_red, s_red2, tid, part); };

    // One evaluation and what the Dormand-Prince step does with it, instantiated per stage where STATIC_STAGES (with a
    // run-time stage every evaluation ends in a merge through which the compiler moves the whole step state -- here four vectors of
    // 4 T + 4 values per lane plus the stage J, through AGPR copy chains).  Returns true when the walker has finished.
#ifdef FF_WIDE_DYNAMIC_STAGES
    constexpr bool STATIC_STAGES = false;
#else
    // (the direct-evaluation fallbacks keep the loop over a run-time stage; so does the one instantiation in which ROCm 7.2's register
    // allocator answered the new layout with its copy-in-front-of-the-exec-restore bug: tools/check_agpr_spills.py, docs/LOG.md round 2)
    constexpr bool STATIC_STAGES = TAB && !(D == 3 && T == 3 && sizeof(TJ) == 4 && FIN);
#endif
    auto evaluate = [&](auto stage_tag) -> bool {
      constexpr int SG = decltype(stage_tag)::value;
      if constexpr (SG == FF_STAGE_DYN && STATIC_STAGES) FF_ASSUME(s <= 0);
      const int sv = SG == FF_STAGE_DYN ? s : SG;
      TJ outJ[NVJ];
      double out[NVS];
#ifdef FF_WIDE_SELFCHECK   // (experiment of DESIGN.md 4: every right-hand side is evaluated twice from the same registers and compared)
      TJ outJ_first[NVJ], outJ_second[NVJ];
      double out_first[NVS], out_second[NVS];
      for (int rep = 0; rep < 3; rep++) {
      if (rep >= 1) __syncthreads();
#endif
      double gy, g0, g1, g2;
      ff_dp5_coeffs(sv, S.h, C.h0v * S.dir, gy, g0, g1, g2);
      auto form = [&](int v) -> double { return fma(g2, c2[v], fma(g1, c1[v], fma(g0, c0[v], gy * y[v]))); };
      const TJ gyJ = (TJ)gy, g0J = (TJ)g0, g1J = (TJ)g1, g2J = (TJ)g2;
      FF_STAMP(7);
      // ---------------------------------------------------------------- publish z, kbar, J (the stage J stays in registers:
      // it is the B operand of J' = A J)
      auto formJ = [&](int v) -> TJ { return ff_t_fma(g2J, c2J[v], ff_t_fma(g1J, c1J[v], ff_t_fma(g0J, c0J[v], gyJ * yJ[v]))); };
      TJ Jin[NVJ];
      if (!Y_LDS || sv <= 3) {
#pragma unroll
        for (int v = 0; v < NVJ; v++) Jin[v] = formJ(v);
      } else {      // stages 4-6 take their input from c0 / c1 / c2 alone: no LDS read of y's J part (from stage 5 on c0 holds the error accumulator, times 0)
#pragma unroll
        for (int v = 0; v < NVJ; v++) Jin[v] = ff_t_fma(g2J, c2J[v], ff_t_fma(g1J, c1J[v], g0J * c0J[v]));
      }
      const double kb_in = form(IK);
      if (own) { s_z[rp] = form(IZ); s_kb[rp] = kb_in; }
#pragma unroll
      for (int K = 0; K < T; K++)
#pragma unroll
        for (int v = 0; v < 4; v++) s_J[MMA::row(K, v, lg) * JS + 16 * w + lc] = Jin[4 * K + v];
      __syncthreads();
      FF_STAMP(0);
      // ---------------------------------------------------------------- R1, first half: radii, table rows requested
      double rq_rho[NQ][D], rq_dk[NQ][D], rq_r[NQ], rq_ri[NQ], rq_T[NQ][TAB ? NH + 5 : 1], rq_dr[NQ];
      bool rq_ok[NQ];
#pragma unroll
      for (int qk = 0; qk < NQ; qk++) {
        const int id = rq_id[qk];
        const bool act = id >= 0;
        const int a = act ? (id & 31) : 0, bb0 = act ? ((id >> 5) & 31) : 31;
        const bool pair = bb0 != 31;
        double r2 = 0.0;
#pragma unroll
        for (int c = 0; c < D; c++) {
          rq_rho[qk][c] = s_z[a * D + c] - (pair ? s_z[bb0 * D + c] : 0.0);
          rq_dk[qk][c] = s_kb[a * D + c] - (pair ? s_kb[bb0 * D + c] : 0.0);
          r2 = fma(rq_rho[qk][c], rq_rho[qk][c], r2);
        }
        ff_sqrt_rcp(r2, rq_r[qk], rq_ri[qk]);
        rq_dr[qk] = 0.0;
        rq_ok[qk] = true;
        if constexpr (TAB) {
          rq_ok[qk] = ff_table_fetch<NH>(rtab, tab_inv_h, tab_h, pair ? 0 : 1, act ? rq_r[qk] : 1.0, rq_T[qk], rq_dr[qk]);
          if (act && !rq_ok[qk]) off_table = true;
        }
      }
      // ---------------------------------------------------------------- S = J J^T on the matrix cores (under the table fetch):
      // tiles (I, w) = sum_k J[16 I + i][k] J[16 w + j][k]; column blocks beyond M are zero and skipped
      typename MMA::acc_t accS[T], accJ[T];
#pragma unroll
      for (int I = 0; I < T; I++) {
        const typename MMA::acc_t zero = {0, 0, 0, 0};
        accS[I] = zero; accJ[I] = zero;
      }
      {
        const TJ* Jb = &s_J[(16 * w + lc) * JS + lg];
        const TJ* Ja = &s_J[lc * JS + lg];
#pragma unroll
        for (int K = 0; K < T; K++) {
          if (16 * K < M) {      // workgroup-uniform
#pragma unroll
            for (int v = 0; v < 4; v++) {
              const int ks = 4 * K + v;
              const TJ bS = Jb[4 * ks];
              TJ aS[T];
#pragma unroll
              for (int I = 0; I < T; I++) aS[I] = Ja[16 * I * JS + 4 * ks];
#pragma unroll
              for (int I = 0; I < T; I++) accS[I] = MMA::mma(aS[I], bS, accS[I]);
            }
          }
        }
      }
      FF_STAMP(1);
      // ---------------------------------------------------------------- R1, second half: heads, records
      double dsum = 0.0;
#pragma unroll
      for (int qk = 0; qk < NQ; qk++) {
        const int id = rq_id[qk];
        if (id < 0) break;
        const int bb0 = (id >> 5) & 31, pr = id >> 10;
        const bool pair = bb0 != 31;
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
        double rdk = 0.0;
#pragma unroll
        for (int c = 0; c < D; c++) rdk = fma(rho[c], dk[c], rdk);
        const double f1ri = f1 * ri, F1k = f1ri * rdk;
        double* rec = &s_rec[pr * RW];
#pragma unroll
        for (int c = 0; c < D; c++) { rec[c] = rho[c]; rec[QPW + c] = fma(F1k, rho[c], f0 * dk[c]); }
        rec[QF0] = f0; rec[QF1] = f1ri; rec[QGQ] = Ac * ri;
        rec[QRI2] = ri * ri; rec[QF2] = f2; rec[QBC] = Bc;
        dsum = fma(cf, fma(f1, r, D * f0), dsum);           // this radius' share of div v
      }
      __syncthreads();      // records complete; every wave is done reading the stage J ...
      nev++;
      FF_STAMP(2);
#pragma unroll
      for (int I = 0; I < T; I++)
#pragma unroll
        for (int v = 0; v < 4; v++) s_J[MMA::row(I, v, lg) * JS + 16 * w + lc] = accS[I][v];   // ... whose place S takes
      // ---------------------------------------------------------------- row lanes: own-row sums and row p of A
      double vi = 0.0, wk = 0.0, gdi = 0.0, Ad[D];
#pragma unroll
      for (int c = 0; c < D; c++) Ad[c] = 0.0;
#pragma unroll
      for (int k = 0; k < NPK; k++) {
        const double* rec = &s_rec[prec[k] >> 1];
        const double sg = (prec[k] & 1) ? -1.0 : 1.0;
        const double f0 = rec[QF0], rcv = rec[rc], fc = rec[QF1] * rcv;
        vi = fma(sg * f0, rcv, vi);
        wk = fma(sg, rec[QPW + rc], wk);
        gdi = fma(sg * rec[QGQ], rcv, gdi);
        TJ* arow = &s_A[pdst[k]];
#pragma unroll
        for (int c = 0; c < D; c++) {
          const double Bcc = fma(fc, rec[c], rc == c ? f0 : 0.0);
          Ad[c] += Bcc;
          arow[c] = (TJ)(-Bcc);
        }
      }
      vi = ff_quad_sum(vi); wk = ff_quad_sum(wk); gdi = ff_quad_sum(gdi);
#pragma unroll
      for (int c = 0; c < D; c++) Ad[c] = ff_quad_sum(Ad[c]);
      if (rowlane) {
        double adv = Ad[0];
#pragma unroll
        for (int c = 1; c < D; c++) adv = (rs == c) ? Ad[c] : adv;
        if (rs < D) s_A[rp * JS + ra * D + rs] = (TJ)adv;                       // the diagonal block of particle ra, row rc
        if (rs == 3) s_A[M * JS + rp] = (TJ)(-gdi);                              // row M: (grad Delta)' = -g^T J
      }
      __syncthreads();
      FF_STAMP(3);
      // ---------------------------------------------------------------- J' = A J on the matrix cores: tiles (I, w) =
      // sum_k A[16 I + i][k] J[k][16 w + j]; the B operand of k-step (K, v) is the lane's own Jin[4 K + v], i.e. row MMA::row(K, v, g)
      if constexpr (!F32 && T >= 4) {      // fp64 at four tiles: 32 registers are worth more than 64 FMAs -- the stage J is formed again
#pragma unroll
        for (int v = 0; v < NVJ; v++) { Jin[v] = formJ(v); }
      }
      {
        const TJ* Aa = &s_A[lc * JS + MMA::row_lane(lg)];
#pragma unroll
        for (int K = 0; K < T; K++) {
          if (16 * K < M) {      // workgroup-uniform
#pragma unroll
            for (int v = 0; v < 4; v++) {
              const int ks = 4 * K + v;
              TJ aJ[T];
#pragma unroll
              for (int I = 0; I < T; I++) aJ[I] = Aa[16 * I * JS + MMA::row_step(K, v)];
#pragma unroll
              for (int I = 0; I < T; I++) accJ[I] = MMA::mma(aJ[I], Jin[ks], accJ[I]);
            }
          }
        }
      }
      FF_STAMP(4);
      // ---------------------------------------------------------------- R2: radius lanes contract their terms with S
      double qsum = 0.0;
#pragma unroll
      for (int qk = 0; qk < NQ; qk++) {
        const int id = rq_id[qk];
        if (id < 0) break;
        const int a = id & 31, bb0 = (id >> 5) & 31, pr = id >> 10;
        const bool pair = bb0 != 31;
        const int bb = pair ? bb0 : a;
        double* rec = &s_rec[pr * RW];
        double rho[D];
#pragma unroll
        for (int c = 0; c < D; c++) rho[c] = rec[c];
        const double f1ri = rec[QF1], gq = rec[QGQ], ri2 = rec[QRI2], f2 = rec[QF2], Bc = rec[QBC];
        double w1[D], qq = 0.0, tr = 0.0;
#pragma unroll
        for (int c = 0; c < D; c++) {
          double t = 0.0;
#pragma unroll
          for (int c2i = 0; c2i < D; c2i++) {
            double ww = (double)s_J[(a * D + c) * JS + a * D + c2i];
            if (pair) ww += ((double)s_J[(bb * D + c) * JS + bb * D + c2i] - (double)s_J[(a * D + c) * JS + bb * D + c2i]) - (double)s_J[(a * D + c2i) * JS + bb * D + c];
            t = fma(ww, rho[c2i], t);
            if (c2i == c) tr += ww;
          }
          w1[c] = t;
          qq = fma(rho[c], t, qq);
        }
        qq *= ri2;
        const double tq = tr - qq;
        const double F2 = fma(f2, qq, f1ri * tq), F1x2 = 2.0 * f1ri;
#pragma unroll
        for (int c = 0; c < D; c++) rec[QPW + c] = fma(F2, rho[c], F1x2 * w1[c]);
        qsum += fma(Bc, qq, gq * tq);
      }
      __syncthreads();
      FF_STAMP(5);
      // ---------------------------------------------------------------- second-order sums, right-hand side
      double qs = 0.0;
#pragma unroll
      for (int k = 0; k < NPK; k++) qs = fma((prec[k] & 1) ? -1.0 : 1.0, s_rec[(prec[k] >> 1) + QPW + rc], qs);
      qs = ff_quad_sum(qs);
#pragma unroll
      for (int I = 0; I < T; I++)
#pragma unroll
        for (int v = 0; v < 4; v++) outJ[4 * I + v] = accJ[I][v];
      out[IZ] = own ? vi : 0.0;
      out[IK] = own ? wk + qs : 0.0;
      out[IDL] = -dsum;
      out[ILP] = -(qsum + (own ? gdi * kb_in : 0.0));
      FF_STAMP(6);
#ifdef FF_WIDE_SELFCHECK
      if (rep == 0) {
#pragma unroll
        for (int v = 0; v < NVJ; v++) outJ_first[v] = outJ[v];
#pragma unroll
        for (int v = 0; v < NVS; v++) out_first[v] = out[v];
        nev--;
      } else if (rep == 1) {
#pragma unroll
        for (int v = 0; v < NVJ; v++) outJ_second[v] = outJ[v];
#pragma unroll
        for (int v = 0; v < NVS; v++) out_second[v] = out[v];
        nev--;
      } else {
        int bad_idx = -1;
        double va = 0.0, vb = 0.0, vc = 0.0;
#pragma unroll
        for (int v = 0; v < NVJ; v++)
          if (!(outJ_first[v] == outJ[v] && outJ_second[v] == outJ[v]) && bad_idx < 0) { bad_idx = 100 + v; va = (double)outJ_first[v]; vb = (double)outJ_second[v]; vc = (double)outJ[v]; }
#pragma unroll
        for (int v = 0; v < NVS; v++)
          if (!(out_first[v] == out[v] && out_second[v] == out[v]) && bad_idx < 0) { bad_idx = v; va = out_first[v]; vb = out_second[v]; vc = out[v]; }
        if (bad_idx >= 0 && A.stats) {      // stats[8]: count; two records (lane, component, stage, walker, agreement code, three values as floats) behind it
          const int slot = atomicAdd(&A.stats[8], 1);
          if (slot < 2) {
            int* q = A.stats + 9 + 9 * slot;
            q[0] = tid; q[1] = bad_idx; q[2] = s; q[3] = (int)b; q[4] = (va == vb ? 1 : 0) | (vb == vc ? 2 : 0) | (va == vc ? 4 : 0);
            q[5] = __float_as_int((float)va); q[6] = __float_as_int((float)vb); q[7] = __float_as_int((float)vc);
            q[8] = __float_as_int((float)((va - vc) / (fabs(vc) + 1e-300)));
          }
        }
      }
      }      // rep
#endif
      s = ff_dp5_consume2<NVJ, TJ, decltype(c3J), NVS, decltype(c3), decltype(yJ), decltype(wgt), decltype(gsum), STATIC_STAGES>(sv, S, C, yJ, c0J, c1J, c2J, c3J, outJ, sens_w, y, c0, c1, c2, c3, out, wgt, gsum);
      FF_STAMP(8);
      return s == 99;
    };
    if constexpr (STATIC_STAGES) {
#pragma unroll 1
      for (;;) {
        bool fin = false;
#pragma unroll 1
        while (s <= 0 && !fin) fin = evaluate(ff_stage_c<FF_STAGE_DYN>{});
        if (fin) break;
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
    // -------------------------------------------------------------------- results
    const bool failed = S.fail != 0;
    const double bad = failed ? __builtin_nan("") : 0.0;
    const double delta = gsum(y[IDL]);
    const double lapd = gsum(y[ILP]);
    __syncthreads();
#pragma unroll
    for (int K = 0; K < T; K++)
#pragma unroll
      for (int v = 0; v < 4; v++) s_J[MMA::row(K, v, lg) * JS + 16 * w + lc] = yJ[4 * K + v];
    __syncthreads();
    // -------------------------------------------------------------------- fused finish (ff_eloc; workgroup-uniform branch)
    // What ff_eloc_slater_rows_kernel + the contraction kernels did from the workspace -- 8 M^2 bytes per walker written and read
    // back: 4.09 GB per launch at configs[4] -- on the workgroup that still holds J (src/utils.py:56-63, src/VMC.py:48-55):
    //   grad_x logp = J^T g0 - grad Delta          lap_x logp = tr(H0 J J^T) + g0 . kbar - lap Delta
    // The Slater table of z(t0) built by the whole workgroup (ff_slater_table_wg, ff_slater_rows.h) into LDS; S = J J^T of the final J
    // on the matrix cores exactly as every right-hand side forms it (in the precision of the sensitivity matrices).
    if constexpr (FIN) {
      __shared__ ff_slater_rows_smem<2> s_sl;
      constexpr int NH2 = D * (D + 1) / 2;
      // scratch of ff_slater_table_wg: (species, particle, orbital) entries for the largest nup^2 + ndn^2 this instantiation can meet
      constexpr int NSA = NCAP < FF_MAX_NS ? NCAP : FF_MAX_NS, NSQ = NSA * NSA + (NCAP - NSA) * (NCAP - NSA);
      __shared__ double s_orbv[(1 + D + NH2) * NSQ];
      __shared__ int s_odeg[D * NCAP];
      const int nup = A.fin.nup, ndn = A.fin.ndn;
      double* const qs = s_rec;                                   // the walker's Slater slots (layout: ff_slater_rows.h), records are dead
      const int oS = M, oT = M + NH2 * n, oL = oT + D * (nup * nup + ndn * ndn);
      const double xown = own ? A.y_in[b * M + rp] : 0.0;        // (for V(x), requested here: an exposed round trip otherwise)
      if (own) s_z[rp] = y[IZ];
      if (A.fin.wstate != nullptr || !degrees_known) {             // one orbital set for every walker: decoded once per kernel
        ff_slater_wg_degrees<D>(s_odeg, tid, nup, ndn, A.fin.tab_up, A.fin.tab_dn, A.fin.wstate ? A.fin.wstate[b] : 0);
        degrees_known = true;
      }
      __syncthreads();
      ff_slater_table_wg<D, NTHR>(s_sl, s_orbv, s_odeg, tid, nup, ndn, s_z, qs);
      // S = J J^T of the final J on the matrix cores, as every right-hand side forms it (tiles (I, w)) -> s_A (A is dead)
      {
        typename MMA::acc_t accF[T];
#pragma unroll
        for (int I = 0; I < T; I++) { const typename MMA::acc_t zero = {0, 0, 0, 0}; accF[I] = zero; }
        const TJ* Jb = &s_J[(16 * w + lc) * JS + lg];
        const TJ* Ja = &s_J[lc * JS + lg];
#pragma unroll
        for (int K = 0; K < T; K++) {
          if (16 * K < M) {      // workgroup-uniform
#pragma unroll
            for (int v = 0; v < 4; v++) {
              const int ks = 4 * K + v;
              const TJ bS = Jb[4 * ks];
              TJ aS[T];
#pragma unroll
              for (int I = 0; I < T; I++) aS[I] = Ja[16 * I * JS + 4 * ks];
#pragma unroll
              for (int I = 0; I < T; I++) accF[I] = MMA::mma(aS[I], bS, accF[I]);
            }
          }
        }
#pragma unroll
        for (int I = 0; I < T; I++)
#pragma unroll
          for (int v = 0; v < 4; v++) s_A[MMA::row(I, v, lg) * JS + 16 * w + lc] = accF[I][v];
      }
      __syncthreads();
      // tr(H0 S): thread t takes the same-spin particle pairs (a, c) = t, t + NTHR, ... of both species
      double trp = 0.0;
      {
        const int npu = nup * nup, npt = npu + ndn * ndn;
        for (int e = tid; e < npt; e += NTHR) {
          const int sp = e >= npu ? 1 : 0, ee = sp ? e - npu : e, ns = sp ? ndn : nup, off = sp ? nup : 0;
          const int a = ee / ns, c = ee - a * ns;
          const double* Tq = qs + oT + (sp ? D * npu : 0);         // [comp][a][b]
          double Sb[D][D];
#pragma unroll
          for (int u = 0; u < D; u++)
#pragma unroll
            for (int v2 = 0; v2 < D; v2++) Sb[u][v2] = (double)s_A[(D * (off + a) + u) * JS + D * (off + c) + v2];
          double q = 0.0;
#pragma unroll
          for (int u = 0; u < D; u++)
#pragma unroll
            for (int v2 = 0; v2 < D; v2++) q = fma(Tq[u * ns * ns + a * ns + c] * Tq[v2 * ns * ns + c * ns + a], Sb[u][v2], q);
          q = -q;
          if (a == c) {      // the same-particle block: sum_j hess phi_j(r_a) Dinv_ja, upper triangle (xx, xy[, xz], yy[, yz, zz])
            const double* Sa = qs + oS + NH2 * (off + a);
            int k = 0;
#pragma unroll
            for (int u = 0; u < D; u++)
#pragma unroll
              for (int v2 = u; v2 < D; v2++) { q = fma((u == v2 ? 1.0 : 2.0) * Sa[k], Sb[u][v2], q); k++; }
          }
          trp += 2.0 * q;
        }
      }
      // grad_x logp: direction i on thread i
      double gradi = 0.0, gki = 0.0;
      if (tid < M) {
        double t = 0.0;
        for (int k = 0; k < M; k++) t = fma(qs[k], (double)s_J[k * JS + tid], t);
        gradi = t - (double)s_J[M * JS + tid];
      }
      if (own) gki = qs[rp] * y[IK];
      const double lapv = gsum(trp + gki) - lapd;
      const double g2 = gsum(gradi * gradi);
      const double logp0 = qs[oL] + qs[oL + 1];
      const double g0own = own ? qs[rp] : 0.0;
      // V(x) on the radius lanes
      __syncthreads();
      if (own) s_z[rp] = xown;
      __syncthreads();
      double vl = 0.0;
#pragma unroll
      for (int qk = 0; qk < NQ; qk++) {
        const int id = rq_id[qk];
        if (id < 0) break;
        const int a = id & 31, bb0 = (id >> 5) & 31;
        const bool pair = bb0 != 31;
        double r2 = 0.0;
#pragma unroll
        for (int c = 0; c < D; c++) {
          const double dlt = s_z[a * D + c] - (pair ? s_z[bb0 * D + c] : 0.0);
          r2 = fma(dlt, dlt, r2);
        }
        vl += pair ? A.fin.Z / sqrt(r2) : (A.fin.use_ho ? 0.5 * r2 : 0.0);
      }
      if (!has_mu && A.fin.use_ho && own) { const double xv = s_z[rp]; vl = fma(0.5 * xv, xv, vl); }   // (no one-body radii without mu)
      const double Vv = gsum(vl);
      if (tid < M) {
        if (A.fin.grad) A.fin.grad[b * M + tid] = gradi + bad;
      }
      if (own) {
        A.y_out[b * M + rp] = y[IZ] + bad;
        if (A.fin.glogp0) A.fin.glogp0[b * M + rp] = g0own + bad;
      }
      if (tid == 0) {
        A.dl_out[b] = delta + bad;
        if (A.fin.logp) A.fin.logp[b] = logp0 - delta + bad;
        if (A.fin.lap) A.fin.lap[b] = lapv + bad;
        if (A.fin.V) A.fin.V[b] = Vv;
        if (A.fin.eloc) A.fin.eloc[b] = -0.25 * lapv - 0.125 * g2 + Vv + bad;
        if (A.h_out) A.h_out[b] = C.hmax_acc > 0.0 ? C.hmax_acc : hwarm0;
        if (A.wcost) A.wcost[b] = S.nacc + S.nrej;
        if (A.stats) { atomicAdd(&s_st[0], nev); atomicMax(&s_st[1], S.nacc); atomicAdd(&s_st[2], S.nrej); if (failed) atomicMax(&s_st[3], 1); }
      }
      // (the records are rewritten by every evaluation, the all-zero record behind the last one must be zero again -- and so must
      // the padding of A, which S has just overwritten)
      __syncthreads();
      for (int e = tid; e < (RCAP + 1) * RW; e += NTHR) s_rec[e] = 0.0;
      for (int e = tid; e < MP * JS; e += NTHR) s_A[e] = (TJ)0;
      __syncthreads();
      continue;
    }
    // Jt[b][i][p] = dz_p/dx_i: wave w writes the rows i = w, w + T, ... as contiguous spans
    for (int i = w; i < M; i += T) {
      if (l < M) A.Jt[(b * M + i) * M + l] = (double)s_J[l * JS + i];
    }
    if (own) {
      A.y_out[b * M + rp] = y[IZ] + bad;
      A.kbar[b * M + rp] = y[IK];
      A.dD[b * M + rp] = (double)s_J[M * JS + rp];
      A.Lpart[b * M + rp] = rp == 0 ? lapd : 0.0;
    }
    if (tid == 0) {
      A.dl_out[b] = delta + bad;
      if (A.h_out) A.h_out[b] = C.hmax_acc > 0.0 ? C.hmax_acc : hwarm0;
      if (A.wcost) A.wcost[b] = S.nacc + S.nrej;
      if (A.stats) { atomicAdd(&s_st[0], nev); atomicMax(&s_st[1], S.nacc); atomicAdd(&s_st[2], S.nrej); if (failed) atomicMax(&s_st[3], 1); }
    }
    __syncthreads();
  }
#ifdef FF_STAMPS
  if (A.stats && tid == 0)
    for (int q = 0; q < 9; q++) atomicAdd((unsigned long long*)(A.stats + 8) + q, stamp_acc[q]);
#endif
  if constexpr (TAB) { if (off_table) *A.evt = A.evt_id; }
  __syncthreads();
  if (A.stats && tid == 0 && (s_st[0] || s_st[3])) {
    atomicAdd(&A.stats[0], s_st[0]);
    atomicMax(&A.stats[1], s_st[1]);
    atomicAdd(&A.stats[2], s_st[2]);
    if (s_st[3]) atomicMax(&A.stats[3], 1);
  }
}

// =====================================================================================================================
extern void ff_set_error(const char* msg);
#define FF_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { ff_set_error(hipGetErrorString(e_)); return FF_ELAUNCH; } } while (0)

static std::atomic<int> g_family{-1};   // -1: not decided yet (FF_WIDE in the environment), 0: by particle number, 1: wide for all
bool ff_wide_forced() {
  int f = g_family.load();
  if (f < 0) {
    const char* e = getenv("FF_WIDE");
    f = (e && e[0] == '1') ? 1 : 0;
    g_family.store(f);
  }
  return f == 1;
}
static std::atomic<int> g_sens_bits{64};
static bool ff_wide_sens_fp32() { return g_sens_bits.load() == 32; }
extern "C" int ff_set_sens_precision(int bits) {
  const int prev = g_sens_bits.load();
  g_sens_bits.store(bits == 32 ? 32 : 64);
  return prev;
}
extern "C" int ff_set_kernel_family(int family) {
  const int prev = ff_wide_forced() ? 1 : 0;
  g_family.store(family == 1 ? 1 : 0);
  return prev;
}

int ff_wide_supported(int n, int d) {
  return (d == 2 || d == 3) && n >= 1 && n <= FF_WIDE_NMAX && n * d <= FF_WIDE_MMAX;
}

static int64_t wide_cus() {
  static int64_t n = 0;
  if (n == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    n = cus;
  }
  return n;
}

template <int D, int MODE>
static void launch_wide_flow(void* stream, const ff_fwd_args& a, int n) {
  const unsigned grid = (unsigned)(a.B < 64 * wide_cus() ? a.B : 64 * wide_cus());
  if (a.evt) FF_LAUNCH((ff_wide_flow_kernel<D, MODE, true>), grid, FF_WAVE, stream, a, n);
  FF_LAUNCH((ff_wide_flow_kernel<D, MODE, false>), grid, FF_WAVE, stream, a, n);
}

template <int D, int T>
static void launch_wide_eloc(void* stream, const ff_fwd_args& a, int n) {
  // persistent grid when the launch has a work queue: as many workgroups as stay resident (one per CU at T = 4)
  const int64_t per_cu = T >= 3 ? 1 : (T == 2 ? 2 : 4);
  const int64_t cap = a.queue ? per_cu * wide_cus() : ((int64_t)1 << 20);
  const unsigned grid = (unsigned)(a.B < cap ? a.B : cap);
  // single-precision sensitivities (ff_set_sens_precision(32)): the table kernel with J, A, S in fp32; its fallback stays fp64
#ifndef FF_WIDE_NO_FIN
  if (a.fin.on & 2) {      // fused finish (ff_ode::compact_finish)
    if (a.evt && ff_wide_sens_fp32() && T >= 2) FF_LAUNCH((ff_wide_eloc_kernel<D, (T >= 2 ? T : 2), true, float, true>), grid, FF_WAVE * T, stream, a, n);
    else if (a.evt) FF_LAUNCH((ff_wide_eloc_kernel<D, T, true, double, true>), grid, FF_WAVE * T, stream, a, n);
    FF_LAUNCH((ff_wide_eloc_kernel<D, T, false, double, true>), grid, FF_WAVE * T, stream, a, n);
    return;
  }
#endif
  if (a.evt && ff_wide_sens_fp32() && T >= 2) FF_LAUNCH((ff_wide_eloc_kernel<D, (T >= 2 ? T : 2), true, float, false>), grid, FF_WAVE * T, stream, a, n);
  else if (a.evt) FF_LAUNCH((ff_wide_eloc_kernel<D, T, true, double, false>), grid, FF_WAVE * T, stream, a, n);
  FF_LAUNCH((ff_wide_eloc_kernel<D, T, false, double, false>), grid, FF_WAVE * T, stream, a, n);
}

int ff_wide_eloc_heavy(void* stream, int n, int d, const ff_fwd_args& a, int64_t max_groups) {
  if (!ff_wide_supported(n, d) || (n * d + 4 + 15) / 16 != 1 || !a.evt || a.queue) return FF_EUNSUPPORTED;   // (table kernel, T = 1, grid-stride)
#ifdef FF_WIDE_NO_FIN      // (diagnostic build without the fused-finish instantiations: a caller that counts on them must hear it -- ADVICE r04)
  if (a.fin.on & 2) { ff_set_error("this build (FF_WIDE_NO_FIN) has no fused finish: ff_ode.compact_finish is not available"); return FF_EUNSUPPORTED; }
#endif
  const unsigned grid = (unsigned)(a.B < max_groups ? a.B : max_groups);
#ifndef FF_WIDE_NO_FIN
  if (a.fin.on & 2) {
    if (d == 2) FF_LAUNCH((ff_wide_eloc_kernel<2, 1, true, double, true>), grid, FF_WAVE, stream, a, n);
    else FF_LAUNCH((ff_wide_eloc_kernel<3, 1, true, double, true>), grid, FF_WAVE, stream, a, n);
  } else
#endif
  if (d == 2) FF_LAUNCH((ff_wide_eloc_kernel<2, 1, true, double, false>), grid, FF_WAVE, stream, a, n);
  else FF_LAUNCH((ff_wide_eloc_kernel<3, 1, true, double, false>), grid, FF_WAVE, stream, a, n);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_wide_dispatch_fwd(int mode, void* stream, int n, int d, const ff_fwd_args& a) {
  if (!ff_wide_supported(n, d)) {
    ff_set_error("fused CNF kernels serve n <= 24 particles with n*d <= 60 in d = 2, 3");
    return FF_EUNSUPPORTED;
  }
#ifdef FF_WIDE_NO_FIN
  if (mode >= 2 && (a.fin.on & 2)) { ff_set_error("this build (FF_WIDE_NO_FIN) has no fused finish: ff_ode.compact_finish is not available"); return FF_EUNSUPPORTED; }
#endif
  if (mode == 0) { if (d == 2) launch_wide_flow<2, 0>(stream, a, n); else launch_wide_flow<3, 0>(stream, a, n); }
  else if (mode == 1) { if (d == 2) launch_wide_flow<2, 1>(stream, a, n); else launch_wide_flow<3, 1>(stream, a, n); }
  else {
    const int T = (n * d + 4 + 15) / 16;
#define FF_WE(D_, T_) if (d == D_ && T == T_) launch_wide_eloc<D_, T_>(stream, a, n);
    FF_WE(2, 1) FF_WE(2, 2) FF_WE(2, 3) FF_WE(2, 4) FF_WE(3, 1) FF_WE(3, 2) FF_WE(3, 3) FF_WE(3, 4)
#undef FF_WE
  }
  FF_LAUNCH_CHECK();
  return FF_OK;
}
