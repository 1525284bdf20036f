// ff_eloc_rows.h -- local-energy sensitivities, row layout (included by ff_cnf_fwd.hip).
//
// Same system as MODE 2 of ff_ode_fwd_kernel (src/VMC.py:46-49 / src/utils.py:40-65 by forward sensitivities):
//     z' = v(z)            J' = A J   (A = dv/dz, J = dz/dx)         kbar' = A kbar + sum_i D2v[u_i, u_i]
//     Delta' = -div v      (grad Delta)' = -J^T g  (g = grad_z div v)  (lap Delta)' = -(sum_i D2div[u_i, u_i] + g . kbar)
// with u_i the columns of J.  Two things make it ~3x cheaper per right-hand side than the column-sweep kernel:
//
//  1. The quadratic sources only need S = J J^T.  For a pair term eta(|rho|) rho, rho = z_a - z_b, with
//     W = S_aa + S_bb - S_ab - S_ba (the D x D block sum; W = S_aa for a one-body term):
//         sum_i D2[delta_i, delta_i] = (2 eta'/r) W rho + [eta'' q + eta' (tr W - q)/r] rho,   q = rho^T W rho / r^2,
//         sum_i D2div[...]           = phi'' q + phi' (tr W - q)/r                            (phi = c (eta' r + D eta)),
//     i.e. O(1) work per radius instead of O(M) -- done by the lane that owns the radius.
//  2. Lane (g, p) owns ROW p of J (all directions) instead of a column.  Then J' = A J and S = J J^T are the SAME
//     broadcast sweep over the rows of J in LDS: per row q the lane does  dJ_p += A_pq row_q  and  S_pq = J_p . row_q.
//     A is assembled from D x D pair blocks B = eta I + (eta'/r) rho rho^T that the radius lanes leave in LDS.
//
// Lanes: L = M * SPLIT per walker (M = N*D rows, SPLIT lanes per row, each holding MC = ceil(M/SPLIT) columns of its row;
// columns beyond M are padding and stay zero), G = 64 / L walkers per wave, one wave per workgroup.
// Per right-hand side: publish -> R1 (radius lanes: heads -> one 2D+8-double record per radius) -> row sweep (lane (a,c)
// walks the partners j of its particle: block row of B from the record, dJ -= B row_j, S_pq = J_p . row_q, own-row sums;
// then its own particle's rows with the accumulated diagonal block) -> R2 (radius lanes: contraction with S, results
// into the record) -> second-order sums + grad-Delta sweep -> Dormand-Prince bookkeeping.
// State per lane: [0] z_p (h = 0 lanes), [1..MC] J_p,chunk, [MC+1] kbar_p (h = 0), [MC+2] part of dDelta/dx_p,
// [MC+3] part of Delta, [MC+4] part of lap Delta -- the three "parts" are plain quadratures, summed at the end.
#pragma once
#ifndef FF_ROWS_UNROLL_J
#define FF_ROWS_UNROLL_J 1
#endif

template <int V>
struct ff_even { static constexpr int v = (V + 1) & ~1; };

// walker stride (doubles) of a per-walker LDS array whose rows are read both as wave-wide broadcasts (16-byte reads,
// one address per walker) and column-wise (8-byte reads, lane p at column p): stride = M (mod 32) doubles spreads the
// walkers of a wave over the banks for both patterns; always even (16-byte alignment of the rows)
template <int SIZE, int M>
struct ff_wstride {
  static constexpr int want = ((M % 32) + 32 - (SIZE % 32)) % 32;
  static constexpr int v = ff_even<SIZE + want>::v;
};

template <int N, int D, int SPLIT, bool TAB, int WPS = 1>
__global__ void __launch_bounds__(FF_WAVE, WPS)
ff_eloc_rows_kernel(ff_fwd_args A) {
  constexpr int M = N * D, MC = (M + SPLIT - 1) / SPLIT, MCOLS = MC * SPLIT, L = M * SPLIT, G = FF_WAVE / L > 16 ? 16 : FF_WAVE / L;
  static_assert(G >= 1, "walker does not fit a wave");
  static_assert(D >= 2, "the scalar shares ride on coordinates 0 and 1 of a particle");
  constexpr int P = N * (N - 1) / 2, R = P + N;
  constexpr int NH = 4, NV = MC + 5;
  constexpr int IK = MC + 1, IDD = MC + 2, IDL = MC + 3, ILP = MC + 4;

  __shared__ ff_wtab s_w[TAB ? 1 : 2][TAB ? 1 : FF_HPAD];
  __shared__ double s_e2[TAB ? 1 : 64];
  __shared__ double s_z[G][M], s_kb[G][M], s_err[G][L];
  // rows of J: [g][q][MCOLS columns | gd_q] (gd_q = d div v / d z_q, filled by the owner of row q)
  constexpr int JROW = ff_even<MCOLS + 1>::v, JWS = ff_wstride<M * JROW, M>::v;
  __shared__ __attribute__((aligned(16))) double s_J[G * JWS];
  // S = J J^T, one partial per column chunk: [h][g][p][q]
  constexpr int SROW = ff_even<M>::v, SWS = ff_wstride<M * SROW, M>::v;
  __shared__ __attribute__((aligned(16))) double s_S[SPLIT][G * SWS];
  // one record per radius (pairs a < b in a-major order, then the one-body radii):
  //   written by R1: [0,D) rho = z_a - z_b   [D] f0 = eta   [D+1] eta'/r   [D+2] gq = c phi'/r   [D+3, 2D+3) pw = D_v[kbar] part
  //                  [2D+4] 1/r^2   [2D+5] eta''   [2D+6] c phi''   [2D+7] c phi (share of div v)
  //   written by R2: [D+3, 2D+3) quad (second-order source of kbar; pw is dead by then)   [2D+3] second-order source of lap Delta
  // (c = 2 for pairs, 1 for one-body radii; phi = eta' r + D eta).  Records of absent radii (no mu) stay zero.
  constexpr int RW = ff_even<2 * D + 8>::v;
  constexpr int QF0 = D, QF1 = D + 1, QGQ = D + 2, QPW = D + 3, QQD = 2 * D + 3, QRI2 = 2 * D + 4, QF2 = 2 * D + 5, QBC = 2 * D + 6, QDS = 2 * D + 7;
  __shared__ __attribute__((aligned(16))) double s_rec[G * R * RW];
  // the error accumulator of the Dormand-Prince step (touched in four of a step's seven stages) lives in lane-private LDS columns
  __shared__ double s_cv[NV][FF_WAVE];
  __shared__ int s_pa[R], s_pb[R], s_any;
  __shared__ int s_st[4];
  __shared__ long long s_next;

  const int lane = threadIdx.x;
  const int g = lane / L, idx = lane % L, h = idx / M, p = idx % M;
  const bool ingrp = g < G;
  const int gg = ingrp ? g : 0;
  const int ai = p / D, ci = p % D;
  const bool owner = (h == 0);
  const int col0 = h * MC;   // first column of this lane's chunk
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
  if (lane == 0) {
    int q = 0;
    for (int a = 0; a < N; a++)
      for (int b = a + 1; b < N; b++) { s_pa[q] = a; s_pb[q] = b; q++; }
    for (int a = 0; a < N; a++) { s_pa[P + a] = a; s_pb[P + a] = -1; }
  }
  // padding columns of the J rows, the gd slots and the records of absent radii start (and stay) at zero
  for (int e = lane; e < G * JWS; e += FF_WAVE) s_J[e] = 0.0;
  for (int e = lane; e < G * R * RW; e += FF_WAVE) s_rec[e] = 0.0;
  FF_WG1_SYNC();
  const int He = A.net.He, Hm = A.net.Hm;
  const bool has_mu = Hm > 0;
  const int nrad = has_mu ? R : P;
  const double tab_inv_h = TAB ? rtab[0] : 0.0, tab_h = TAB ? rtab[1] : 0.0;
  const double rtol = A.rtol, atol = A.atol;
  constexpr double NT = (double)M * M + 3.0 * M + L + N;   // z, J, kbar, the grad-Delta parts, Delta parts, lap parts
  const int64_t ngroups = (A.B + G - 1) / G;
  // radii this lane evaluates (slot qk: radius lane + 64 qk of the wave's G*nrad): walker | a << 4 | b (15: none) << 8 | index << 12
  constexpr int NQ = (G * R + FF_WAVE - 1) / FF_WAVE;
  int rq_id[NQ];
#pragma unroll
  for (int qk = 0; qk < NQ; qk++) {
    const int q = lane + qk * FF_WAVE;
    const bool act = q < G * nrad;
    const int qg = act ? q / nrad : 0, pr = act ? q - qg * nrad : 0;
    rq_id[qk] = act ? (qg | (s_pa[pr] << 4) | ((s_pb[pr] < 0 ? 15 : s_pb[pr]) << 8) | (pr << 12)) : -1;
  }
  // the record of (my particle, partner j) -- j = my particle: its one-body radius -- and the sign of rho = z_mine - z_j in it
  int roff[N];
#pragma unroll
  for (int j = 0; j < N; j++) {
    const int lo = j < ai ? j : ai, hi = j < ai ? ai : j;
    const int pr = (j == ai) ? P + ai : ff_pair_index(N, lo, hi);
    roff[j] = (gg * R + pr) * RW;
  }

#ifdef FF_STAMPS
  unsigned long long stamp_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev = __builtin_amdgcn_s_memtime();
#endif
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
    // Dormand-Prince storage as in ff_ode_fwd_kernel: y, c0..c2 (k0..k2, then the inputs of stages 4, 5 and y_new), c3 (error)
    double y[NV], c0[NV], c1[NV], c2[NV];
    ff_lane_vec<NV, true> c3(&s_cv[0][0], lane);
#pragma unroll
    for (int v = 0; v < NV; v++) { y[v] = 0.0; c0[v] = 0.0; c1[v] = 0.0; c2[v] = 0.0; c3[v] = 0.0; }
    {
      const double y0 = ff_opt_load(A.y_in, valid && owner, b * M + p, A.y_in, 0.25 * (p + 1) + 0.125 * ((p * 7) % 5));   // idle rows: finite, distinct
      y[0] = owner ? y0 : y[0];
    }
#pragma unroll
    for (int k = 0; k < MC; k++) y[1 + k] = (col0 + k == p) ? 1.0 : 0.0;
    ff_stepper S;
    S.begin(A.ta, A.tb, valid);
    // walkers of a low cost class: looser tolerance for the sensitivity components, larger first step (ff_ode.walker_class)
    const bool loose = ff_opt_load(A.wclass, valid, b, A.y_in, (int32_t)0x7fffffff) <= A.sens_class;
    const double hwarm = ff_opt_load(A.h_init, valid, A.h_scale < 0.0 ? 0 : b, A.y_in, 0.0) * (loose ? A.h_scale_loose : fabs(A.h_scale));
    const bool warm = hwarm > 0.0;
    // tolerance of the sensitivity components relative to the coordinates' (ff_ode.sens_tol): weight in the error norm
    const double sens_w = loose ? A.sens_w : 1.0;
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

    // One evaluation and what the step does with it, instantiated per stage.  Returns true when every walker has finished.
    auto evaluate = [&](auto stage_tag) -> bool {
      constexpr int SG = decltype(stage_tag)::value;
      if constexpr (SG == FF_STAGE_DYN) FF_ASSUME(s <= 0);
      const int sv = SG == FF_STAGE_DYN ? s : SG;
      // ------------------------------------------------------------------ stage input
      // lane indices, laundered once per evaluation (FF_OPAQUE: nothing derived from them is hoisted out of this loop)
      int l_g = g, l_gg = gg, l_p = p, l_col0 = col0, l_h = h, l_ai = ai, l_ci = ci, l_idx = idx;
      FF_OPAQUE(l_g); FF_OPAQUE(l_gg); FF_OPAQUE(l_p); FF_OPAQUE(l_col0); FF_OPAQUE(l_h); FF_OPAQUE(l_ai); FF_OPAQUE(l_ci); FF_OPAQUE(l_idx);
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
        default: break;   // -2, 0: the state itself
      }
      // the stage input lives in LDS from here on (z, kbar, the row chunks): the registers are needed for the sweeps; only
      // the accept/reject stage forms it again (for y_new and the error scale)
      auto form = [&](int v) -> double { return fma(g2, c2[v], fma(g1, c1[v], fma(g0, c0[v], gy * y[v]))); };
      double out[NV];
      FF_STAMP(0);
      // ------------------------------------------------------------------ publish z, kbar (owners) and the row chunks
      FF_WG1_SYNC();
      if (ingrp) {
        if (owner) { s_z[l_g][l_p] = form(0); s_kb[l_g][l_p] = form(IK); }
        double* row = &s_J[l_g * JWS + l_p * JROW + l_col0];
#pragma unroll
        for (int k = 0; k < MC; k++) row[k] = form(1 + k);
      }
      FF_WG1_SYNC();
      FF_STAMP(1);
      // ------------------------------------------------------------------ R1: radius lanes
      {
        double rq_rho[NQ][D], rq_dk[NQ][D], rq_r[NQ], rq_ri[NQ], rq_T[NQ][TAB ? NH + 5 : 1], rq_dr[NQ];
        bool rq_ok[NQ];
#pragma unroll
        for (int qk = 0; qk < NQ; qk++) {
          int id = rq_id[qk];
          FF_OPAQUE(id);
          const bool act = id >= 0;
          const int qg = act ? (id & 15) : 0, a = act ? ((id >> 4) & 15) : 0, bb0 = act ? ((id >> 8) & 15) : 15;
          const bool pair = bb0 != 15;
          const int bb = pair ? bb0 : a;
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
          int id = rq_id[qk];
          FF_OPAQUE(id);
          if (id < 0) break;
          const int qg = id & 15, bb0 = (id >> 8) & 15, pr = id >> 12;
          const bool pair = bb0 != 15;
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
          double* rec = &s_rec[(qg * R + pr) * RW];
#pragma unroll
          for (int c = 0; c < D; c++) { rec[c] = rho[c]; rec[QPW + c] = fma(F1k, rho[c], f0 * dk[c]); }
          rec[QF0] = f0; rec[QF1] = f1ri; rec[QGQ] = Ac * ri;
          rec[QRI2] = ri * ri; rec[QF2] = f2; rec[QBC] = Bc;
          rec[QDS] = cf * fma(f1, r, D * f0);                 // this radius' share of div v
        }
      }
      FF_WG1_SYNC();
      nev++;
      FF_STAMP(2);
      // ------------------------------------------------------------------ row sweep: dJ = A J, S = J J^T, own-row sums
      double vi = 0.0, wk = 0.0, gdi = 0.0, Ad[D];
#pragma unroll
      for (int c2i = 0; c2i < D; c2i++) Ad[c2i] = 0.0;
#pragma unroll
      for (int k = 0; k < MC; k++) out[1 + k] = 0.0;
      {
        double Jp[MC];   // this lane's own row chunk, back from LDS
#pragma unroll
        for (int k = 0; k < MC; k++) Jp[k] = s_J[l_gg * JWS + l_p * JROW + l_col0 + k];
#pragma unroll FF_ROWS_UNROLL_J
        for (int j = 0; j < N; j++) {
          // row l_ci of the block B = f0 I + (f1/r) rho rho^T of (my particle, partner j) and my coordinate's own-row terms
          const double* rec = &s_rec[roff[j]];
          const double sg = j < l_ai ? -1.0 : 1.0;
          const double f0 = rec[QF0], fc = rec[QF1] * rec[l_ci];
          vi = fma(sg * f0, rec[l_ci], vi);
          wk = fma(sg, rec[QPW + l_ci], wk);
          gdi = fma(sg * rec[QGQ], rec[l_ci], gdi);
          double Ab[D], sq[D];
#pragma unroll
          for (int c2i = 0; c2i < D; c2i++) {
            const double Bcc = fma(fc, rec[c2i], l_ci == c2i ? f0 : 0.0);
            Ad[c2i] += Bcc;
            Ab[c2i] = (j == l_ai) ? 0.0 : -Bcc;     // my own particle's rows come last, with the accumulated diagonal block
          }
#pragma unroll
          for (int c2i = 0; c2i < D; c2i++) {
            const double* row = &s_J[l_gg * JWS + (j * D + c2i) * JROW + l_col0];
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < MC; k++) {
              const double rk = row[k];
              out[1 + k] = fma(Ab[c2i], rk, out[1 + k]);
              acc = fma(Jp[k], rk, acc);
            }
            sq[c2i] = acc;
          }
          if (ingrp) {
#pragma unroll
            for (int c2i = 0; c2i < D; c2i++) s_S[l_h][l_g * SWS + l_p * SROW + j * D + c2i] = sq[c2i];
          }
        }
#pragma unroll
        for (int c2i = 0; c2i < D; c2i++) {
          const double* row = &s_J[l_gg * JWS + (l_ai * D + c2i) * JROW + l_col0];
#pragma unroll
          for (int k = 0; k < MC; k++) out[1 + k] = fma(Ad[c2i], row[k], out[1 + k]);
        }
      }
      if (ingrp && owner) s_J[l_g * JWS + l_p * JROW + MCOLS] = gdi;   // read by the grad-Delta sweep, two barriers from here
      FF_WG1_SYNC();
      FF_STAMP(3);
      // ------------------------------------------------------------------ R2: radius lanes contract their terms with S
#pragma unroll
      for (int qk = 0; qk < NQ; qk++) {
        int id = rq_id[qk];
        FF_OPAQUE(id);
        if (id < 0) break;
        const int qg = id & 15, a = (id >> 4) & 15, bb0 = (id >> 8) & 15, pr = id >> 12;
        const bool pair = bb0 != 15;
        const int bb = pair ? bb0 : a;
        double* rec = &s_rec[(qg * R + pr) * RW];
        double rho[D];
#pragma unroll
        for (int c = 0; c < D; c++) rho[c] = rec[c];
        const double f1ri = rec[QF1], gq = rec[QGQ], ri2 = rec[QRI2], f2 = rec[QF2], Bc = rec[QBC];
        double W[D][D];
#pragma unroll
        for (int c = 0; c < D; c++)
#pragma unroll
          for (int c2i = 0; c2i < D; c2i++) {
            double w = 0.0;
#pragma unroll
            for (int hh = 0; hh < SPLIT; hh++) {
              const double* Sg = &s_S[hh][qg * SWS];
              w += Sg[(a * D + c) * SROW + a * D + c2i];
              if (pair) w += Sg[(bb * D + c) * SROW + bb * D + c2i] - Sg[(a * D + c) * SROW + bb * D + c2i] - Sg[(a * D + c2i) * SROW + bb * D + c];
            }
            W[c][c2i] = w;
          }
        double w1[D], qq = 0.0, tr = 0.0;
#pragma unroll
        for (int c = 0; c < D; c++) {
          double t = 0.0;
#pragma unroll
          for (int c2i = 0; c2i < D; c2i++) t = fma(W[c][c2i], rho[c2i], t);
          w1[c] = t;
          qq = fma(rho[c], t, qq);
          tr += W[c][c];
        }
        qq *= ri2;
        const double tq = tr - qq;                       // sum_i (|delta_i|^2 - r1_i^2)
        const double F2 = fma(f2, qq, f1ri * tq), F1x2 = 2.0 * f1ri;
#pragma unroll
        for (int c = 0; c < D; c++) rec[QPW + c] = fma(F2, rho[c], F1x2 * w1[c]);
        rec[QQD] = fma(Bc, qq, gq * tq);
      }
      FF_WG1_SYNC();
      FF_STAMP(4);
      // ------------------------------------------------------------------ second-order sources, grad-Delta sweep
      double qs = 0.0, shq = 0.0, shd = 0.0;
#pragma unroll
      for (int j = 0; j < N; j++) {
        const double* rec = &s_rec[roff[j]];
        qs = fma(j < l_ai ? -1.0 : 1.0, rec[QPW + l_ci], qs);
        if (j >= l_ai) { shq += rec[QQD]; shd += rec[QDS]; }   // scalar shares: the radius belongs to its first particle
      }
      double dd = 0.0;
      {
        const double* Jg = &s_J[l_gg * JWS];
#pragma unroll
        for (int k = 0; k < MC; k++) {
          const int q = l_col0 + k;     // this lane's share of the rows
          if (q < M) dd = fma(Jg[q * JROW + MCOLS], Jg[q * JROW + l_p], dd);
        }
      }
      out[0] = owner ? vi : 0.0;
      out[IK] = owner ? wk + qs : 0.0;
      out[IDD] = -dd;
      out[IDL] = (owner && l_ci == 1) ? -shd : 0.0;
      out[ILP] = owner ? -(((l_ci == 0) ? shq : 0.0) + gdi * s_kb[l_gg][l_p]) : 0.0;
      FF_STAMP(5);
      // ------------------------------------------------------------------ consume (Dormand-Prince bookkeeping)
      if (sv == -2) {
#pragma unroll
        for (int v = 0; v < NV; v++) c0[v] = out[v];
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double isc = (v >= 1 ? sens_w : 1.0) * ff_rcp(fma(fabs(y[v]), rtol, atol));
          p0 = fma(y[v] * isc, y[v] * isc, p0);
          p1 = fma(c0[v] * isc, c0[v] * isc, p1);
        }
        const double d0 = sqrt(group_sum(p0) * (1.0 / NT));
        d1v = sqrt(group_sum(p1) * (1.0 / NT));
        h0v = S.h0(d0, d1v);
        s = -1;
        if (!ff_wave_or(&s_any, lane, (!S.done && !warm) ? 1 : 0)) {   // every walker of the wave brings its own first step
          S.habs = fmin(hwarm, S.interval);
          S.plan();
          s = 1;
        }
      } else if (sv == -1) {
        double p2 = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double t = (out[v] - c0[v]) * (v >= 1 ? sens_w : 1.0) * ff_rcp(fma(fabs(y[v]), rtol, atol));
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
          const double t = e * (v >= 1 ? sens_w : 1.0) * ff_rcp(fma(fmax(fabs(y[v]), fabs(c2[v])), rtol, atol));   // c2 = the candidate y_new
          pe = fma(t, t, pe);
        }
        const double err = sqrt(group_sum(pe) * (1.0 / NT));
        const bool was_active = !S.done;
        const bool acc = S.decide(err, A.max_steps);
        if (acc) hmax_acc = fmax(hmax_acc, fabs(hs));
        if (acc) {
#pragma unroll
          for (int v = 0; v < NV; v++) { y[v] = c2[v]; c0[v] = out[v]; }
        }
        S.plan();
        const int any = ff_wave_or(&s_any, lane, S.done ? 0 : ((was_active && !acc) ? 3 : 1));
        s = (any & 2) ? 0 : 1;
        if (!any) { FF_STAMP(6); return true; }
      }
      FF_STAMP(6);
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
    // ---------------------------------------------------------------------- results: combine the per-lane parts
    const double delta = group_sum(y[IDL]);
    double dD_p = y[IDD];
    if constexpr (SPLIT > 1) {
      if (ingrp) s_err[g][idx] = y[IDD];
      FF_WG1_SYNC();
      dD_p = 0.0;
#pragma unroll
      for (int hh = 0; hh < SPLIT; hh++) dD_p += s_err[gg][hh * M + p];
      FF_WG1_SYNC();
    }
    if (valid) {
      const bool failed = S.fail != 0;
      const double bad = failed ? __builtin_nan("") : 0.0;   // failed integration -> NaN results (see ff_ode_fwd_kernel)
      if (owner) {
        A.y_out[b * M + p] = y[0] + bad;
        A.kbar[b * M + p] = y[IK];
        A.dD[b * M + p] = dD_p;
        A.Lpart[b * M + p] = y[ILP];
      }
#pragma unroll
      for (int k = 0; k < MC; k++) {
        if (col0 + k < M) A.Jt[(b * M + col0 + k) * M + p] = y[1 + k];   // Jt[b][i][k] = dz_k/dx_i
      }
      if (idx == 0) {
        A.dl_out[b] = delta + bad;
        if (A.h_out) A.h_out[b] = hmax_acc > 0.0 ? hmax_acc : hwarm;
        if (A.wcost) A.wcost[b] = S.nacc + S.nrej;
        if (A.stats) { atomicAdd(&s_st[0], nev); atomicMax(&s_st[1], S.nacc); atomicAdd(&s_st[2], S.nrej); if (failed) atomicMax(&s_st[3], 1); }
      }
    }
    FF_WG1_SYNC();
  }
#ifdef FF_STAMPS
  FF_STAMP(7);
  if (A.stats && lane == 0)
    for (int q = 0; q < 9; q++) atomicAdd((unsigned long long*)(A.stats + 8) + q, stamp_acc[q]);
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
