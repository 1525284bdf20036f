// ff_adj_wide.h -- adjoint of CNF.delta_logp (SolveIVP.backward, src/NeuralODE/nnModule.py:76-133) for walkers that do not fit
// the one-wave-per-walker-group kernels of ff_cnf_adj.hip: ONE WALKER PER WORKGROUP of W waves (W = 2 up to 128 radii, 4
// beyond), particle number at run time (n <= 24, n d <= 60; ff_wide.hip has the forward passes).  Included by ff_cnf_adj.hip.
//
// Same augmented system (z, a_z, theta-quadrature), same two variants, exactly one of which runs (decided on the device
// from the radial-table header, as for the narrow kernels):
//   ff_wide_adjtab_kernel<D, W, NQ>  derivative heads from the radial table; per (stage, radius) a 4-number record, added with
//                                    the Runge-Kutta weights into the workgroup-private deposit table when the step is accepted
//                                    (wave after wave, lanes in order: a fixed summation order, bit-reproducible);
//   ff_wide_adj_kernel<D, W, NQ>     direct evaluation: radius lanes evaluate all hidden units for the heads, and every lane
//                                    integrates the parameter gradient of ITS hidden units (tid + 64 W j) over all radii.
// Lane tid owns coordinate tid (z, a_z; tid < M) and the radii tid, tid + 64 W, ... (NQ slots: one radius per lane up to 22
// particles -- a one-wave version with five radii per lane took 414 registers and 38 ms per 131 072 walkers at 20 particles).
#pragma once
#include "ff_dp5.h"

#define FF_WADJ_NMAX 24
#define FF_WADJ_RMAX (FF_WADJ_NMAX * (FF_WADJ_NMAX + 1) / 2)

FF_D int ff_wadj_radius_id(int n, int q, int nrad) {   // a | b << 5 | q << 10 (b = 31: one-body); -1 beyond nrad
  if (q >= nrad) return -1;
  const int P = n * (n - 1) / 2;
  if (q >= P) return (q - P) | (31 << 5) | (q << 10);
  int a = 0, off = 0;
  while (q >= off + (n - 1 - a)) { off += n - 1 - a; a++; }
  return a | ((a + 1 + q - off) << 5) | (q << 10);
}
FF_D int ff_wadj_partner(int n, int P, int a, int j) {
  const int lo = j < a ? j : a, hi = j < a ? a : j;
  return (j == a) ? P + a : ff_pair_index(n, lo, hi);
}
template <int NTHR>
FF_D double ff_wadj_sum(double* s_red, double* s_red2, int lane, double part) {
  s_red[lane] = part;
  __syncthreads();
  if (lane < 16) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < NTHR / 16; k++) t += s_red[lane + 16 * k];
    s_red2[lane] = t;
  }
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) t += s_red2[k];
  __syncthreads();
  return t;
}

// seeds of walker b: a_z (this lane's coordinate) and a_Delta
FF_D void ff_wadj_seeds(const ff_adj_args& A, int64_t b, int M, int lane, bool own, double& az, double& ad) {
  const bool ws = A.w_e != nullptr;
  const int wi = ff_opt_load(A.w_index, ws, b, A.z_in, (int32_t)0);
  const double wb = (ff_opt_load(A.w_e, true, b, A.z_in, 0.0) - ff_opt_load(A.w_mean, ws, wi, A.z_in, 0.0)) * A.w_scale;
  const double az0 = ff_opt_load(A.az_in, own, b * M + lane, A.z_in, 0.0), ad0 = ff_opt_load(A.ad_in, true, b, A.z_in, 0.0);
  az = ws ? wb * az0 : az0;
  ad = ws ? -wb : ad0;
}

#ifndef FF_WADJ_WPS
#define FF_WADJ_WPS 1      // waves per SIMD the tabulated kernel is compiled for
#endif
template <int D, int W, int NQ>
__global__ void __launch_bounds__(FF_WAVE * W, FF_WADJ_WPS)
ff_wide_adjtab_kernel(ff_adj_args A, int n) {
  constexpr int NV = 2, NTHR = FF_WAVE * W;
  const double* __restrict__ rtab = A.net.radial_table;
  if (!(rtab && rtab[3] == 0.0 && rtab[4] == 0.0)) return;   // the direct-evaluation kernel serves this call

  __shared__ double s_z[FF_WAVE], s_kb[FF_WAVE], s_red[NTHR], s_red2[16];
  __shared__ double s_T[NQ * NTHR][3 * D];                     // per radius: its contribution to particle a's rows of v, Dv^T[lambda], grad div
  __shared__ double s_W[2][FF_DEP_NLDS][FF_DEP_LROW];
  __shared__ int s_st[4];

  const int lane = threadIdx.x;
  const int M = n * D, P = n * (n - 1) / 2;
  for (int e = lane; e < 2 * FF_DEP_NLDS * FF_DEP_LROW; e += NTHR) (&s_W[0][0][0])[e] = 0.0;
  for (int e = lane; e < NQ * NTHR * 3 * D; e += NTHR) (&s_T[0][0])[e] = 0.0;
  if (lane < 4) s_st[lane] = 0;
  __syncthreads();
  const bool has_mu = A.net.Hm > 0;
  const int nrad = has_mu ? P + n : P;
  const double tab_inv_h = rtab[0], tab_h = rtab[1];
  int rq_id[NQ];
#pragma unroll
  for (int sl = 0; sl < NQ; sl++) rq_id[sl] = ff_wadj_radius_id(n, lane + sl * NTHR, nrad);
  const bool own = lane < M;
  const int ai = own ? lane / D : 0, ci = own ? lane % D : 0;
  bool off_any = false;
  double* const ovf = A.trows + (size_t)gridDim.x * 2 * FF_DEP_NLDS * FF_DEP_ROW;   // Wtot region: [2][NTOT][ROW]

  for (int64_t bq = blockIdx.x; bq < A.B; bq += gridDim.x) {
    const int64_t b = ff_opt_load(A.order, true, bq, A.z_in, (int32_t)bq);
    double y[NV] = {0.0, 0.0}, c0[NV] = {0.0, 0.0}, c1[NV] = {0.0, 0.0}, c2[NV] = {0.0, 0.0}, c3[NV] = {0.0, 0.0};
    double ad;
    y[0] = ff_opt_load(A.z_in, own, b * M + lane, A.z_in, 0.0);
    ff_wadj_seeds(A, b, M, lane, own, y[1], ad);
    ff_rec r0[NQ], r2[NQ], r3[NQ], r4[NQ], r5[NQ];
    ff_stepper S;
    S.begin(A.ta, A.tb, true);
    ff_dp5_ctl C;
    C.rtol = A.rtol; C.atol = A.atol; C.nt_inv = 1.0 / (2.0 * M); C.max_steps = A.max_steps;
    C.hwarm = ff_opt_load(A.h_init, true, A.h_scale < 0.0 ? 0 : b, A.z_in, 0.0) * fabs(A.h_scale);
    if (!(C.hwarm > 0.0)) C.hwarm = 0.0;
    C.h0v = 0.0; C.d1v = 0.0; C.hmax_acc = 0.0;
    const double hwarm0 = C.hwarm;
    int s = -2, nev = 0;
    auto wgt = [&](int v) -> double { return 1.0; };
    auto gsum = [&](double part) -> double { return ff_wadj_sum<NTHR>(s_red, s_red2, lane, part); };

#pragma unroll 1
    for (;;) {
      double gy, g0, g1, g2;
      ff_dp5_coeffs(s, S.h, C.h0v * S.dir, gy, g0, g1, g2);
      __syncthreads();
      if (own) {
        s_z[lane] = fma(g2, c2[0], fma(g1, c1[0], fma(g0, c0[0], gy * y[0])));
        s_kb[lane] = fma(g2, c2[1], fma(g1, c1[1], fma(g0, c0[1], gy * y[1])));
      }
      __syncthreads();
      // ------------------------------------------------------------------ radius phase (branch-free: a slot without a radius
      // computes on particle 0 and writes row lane + 64 sl of s_T, which no coordinate reads -- see ff_opt_load, ff_common.h)
      ff_rec cur[NQ];
#pragma unroll
      for (int sl = 0; sl < NQ; sl++) {
        const int id = rq_id[sl];
        const bool act = id >= 0;
        const int a = act ? (id & 31) : 0, bq0 = act ? ((id >> 5) & 31) : 31, pr = act ? (id >> 10) : lane + sl * NTHR;
        const bool pair = bq0 != 31;
        double rho[D], dl[D], r2 = 0.0, al = 0.0;
#pragma unroll
        for (int c = 0; c < D; c++) {
          rho[c] = s_z[a * D + c] - (pair ? s_z[bq0 * D + c] : 0.0);
          dl[c] = s_kb[a * D + c] - (pair ? s_kb[bq0 * D + c] : 0.0);
          r2 = fma(rho[c], rho[c], r2);
          al = fma(dl[c], rho[c], al);
        }
        double r, ri, hd[3] = {0.0, 0.0, 0.0};
        ff_sqrt_rcp(r2, r, ri);
        const bool ok = ff_heads_table<3>(rtab, tab_inv_h, tab_h, pair ? 0 : 1, r, hd);
        hd[0] = ok ? hd[0] : 0.0; hd[1] = ok ? hd[1] : 0.0; hd[2] = ok ? hd[2] : 0.0;
        double jf = rint(r * FF_DEP_INVH);
        if (act && (!ok || !(jf <= (double)(FF_DEP_NTOT - 1)))) off_any = true;   // beyond either table (or NaN): the direct kernel redoes the call
        jf = fmin(jf, (double)(FF_DEP_NTOT - 1));
        cur[sl].j = (act && r == r) ? (int)jf : 0;
        cur[sl].dr = act ? fma(-jf, 1.0 / FF_DEP_INVH, r) : 0.0;
        cur[sl].ca = act ? (pair ? -(al - 2.0 * D * ad) : -(al - D * ad)) : 0.0;
        cur[sl].cb = act ? (pair ? 2.0 * ad * r : ad * r) : 0.0;
        const double f0 = hd[0], f1 = hd[1], f2 = hd[2];
        const double F1 = f1 * (al * ri), gq = (pair ? 2.0 : 1.0) * fma(f2, r, (1.0 + D) * f1) * ri;
        double* Tq = &s_T[pr][0];
#pragma unroll
        for (int c = 0; c < D; c++) { Tq[c] = f0 * rho[c]; Tq[D + c] = fma(F1, rho[c], f0 * dl[c]); Tq[2 * D + c] = gq * rho[c]; }
      }
      __syncthreads();
      nev++;
      // ------------------------------------------------------------------ component phase
      double out[NV] = {0.0, 0.0};
      if (own) {
        double vi = 0.0, dvk = 0.0, gdi = 0.0;
        for (int j = 0; j < n; j++) {
          if (j == ai && !has_mu) continue;
          const double* Tq = &s_T[ff_wadj_partner(n, P, ai, j)][ci];
          const double sg = j < ai ? -1.0 : 1.0;
          vi = fma(sg, Tq[0], vi); dvk = fma(sg, Tq[D], dvk); gdi = fma(sg, Tq[2 * D], gdi);
        }
        out[0] = vi;
        out[1] = fma(ad, gdi, -dvk);
      }
      // ------------------------------------------------------------------ records of the stage, then the stage machine
      const int s_was = s;
      const double h_was = S.h;
      const int nacc_was = S.nacc;
      if (s_was == -2 || s_was == 0) {
#pragma unroll
        for (int sl = 0; sl < NQ; sl++) r0[sl] = cur[sl];
      } else if (s_was == 2) {
#pragma unroll
        for (int sl = 0; sl < NQ; sl++) r2[sl] = cur[sl];
      } else if (s_was == 3) {
#pragma unroll
        for (int sl = 0; sl < NQ; sl++) r3[sl] = cur[sl];
      } else if (s_was == 4) {
#pragma unroll
        for (int sl = 0; sl < NQ; sl++) r4[sl] = cur[sl];
      } else if (s_was == 5) {
#pragma unroll
        for (int sl = 0; sl < NQ; sl++) r5[sl] = cur[sl];
      }
      s = ff_dp5_consume<NV>(s, S, C, y, c0, c1, c2, c3, out, wgt, gsum);
      if (s_was == 6 && S.nacc != nacc_was) {     // accepted (workgroup-uniform): deposit the step, the record of k6 opens the next one
#pragma unroll 1
        for (int ww = 0; ww < W; ww++) {          // one wave at a time, lanes in order: a fixed summation order
          if (lane / FF_WAVE == ww) {
#pragma unroll
            for (int sl = 0; sl < NQ; sl++) {
              if (rq_id[sl] >= 0) {
                const int t = ((rq_id[sl] >> 5) & 31) != 31 ? 0 : 1;
                ff_deposit5(s_W, ovf, t, r0[sl], r2[sl], r3[sl], r4[sl], r5[sl], h_was);
              }
            }
          }
          __syncthreads();
        }
#pragma unroll
        for (int sl = 0; sl < NQ; sl++) r0[sl] = cur[sl];
      }
      if (s == 99) break;
    }
    const double bad = S.fail ? __builtin_nan("") : 0.0;   // failed integration -> NaN gradients
    if (own && A.gx_out) A.gx_out[b * M + lane] = y[1] + bad;
    if (lane == 0) {
      if (S.fail) atomicAdd(&s_W[0][0][0], bad);
      if (A.h_out) A.h_out[b] = C.hmax_acc > 0.0 ? C.hmax_acc : hwarm0;
      if (A.wcost) A.wcost[b] = S.nacc + S.nrej;
      if (A.stats) { atomicAdd(&s_st[0], nev); atomicMax(&s_st[1], S.nacc); atomicAdd(&s_st[2], S.nrej); if (S.fail) atomicMax(&s_st[3], 1); }
    }
    __syncthreads();
  }
  if (off_any) *A.off_table = 1.0;
  {   // flush the workgroup-private coefficient table
    double* row = A.trows + (size_t)blockIdx.x * 2 * FF_DEP_NLDS * FF_DEP_ROW;
    for (int e = lane; e < 2 * FF_DEP_NLDS * FF_DEP_ROW; e += NTHR) row[e] = (&s_W[0][0][0])[(e / FF_DEP_ROW) * FF_DEP_LROW + e % FF_DEP_ROW];
  }
  __syncthreads();
  if (A.stats && lane == 0 && (s_st[0] || s_st[3])) {
    atomicAdd(&A.stats[0], s_st[0]);
    atomicMax(&A.stats[1], s_st[1]);
    atomicAdd(&A.stats[2], s_st[2]);
    if (s_st[3]) atomicMax(&A.stats[3], 1);
  }
}

// Direct evaluation.  Parameter gradient: lane l integrates, for its hidden units u = l + 64 j of eta and mu, the quadrature
//   dtheta*/dt = ca df(r)/dtheta + cb df'(r)/dtheta   summed over the radii of that net,   (ca, cb) as in the tabulated kernel,
// with the Runge-Kutta weights of accepted steps, and adds it to the workgroup's row of A.rows (its own entries only: plain
// read-modify-writes); ff_rows_reduce_kernel sums the rows.
template <int D, int W, int NQ>
__global__ void __launch_bounds__(FF_WAVE * W)
ff_wide_adj_kernel(ff_adj_args A, int n) {
  constexpr int NV = 2, NTHR = FF_WAVE * W, MAXU = (FF_HMAX + NTHR - 1) / NTHR;
  {
    const double* rt = A.net.radial_table;
    if (rt && rt[3] == 0.0 && rt[4] == 0.0 && *A.off_table == 0.0) return;   // the tabulated kernel served this call
  }
  __shared__ ff_wtab s_w[2][FF_HPAD];
  __shared__ double s_e2[64];
  __shared__ double s_z[FF_WAVE], s_kb[FF_WAVE], s_red[NTHR], s_red2[16];
  __shared__ double s_T[NQ * NTHR][3 * D];
  __shared__ double s_q[NQ * NTHR][3];        // r, ca, cb of every radius of the stage
  __shared__ int s_st[4];

  const int lane = threadIdx.x;
  const int M = n * D, P = n * (n - 1) / 2;
  if (lane < FF_WAVE) { ff_load_weights(s_w, A.net, lane); ff_fill_exp2_table(s_e2, lane); }
  for (int e = lane; e < NQ * NTHR * 3 * D; e += NTHR) (&s_T[0][0])[e] = 0.0;
  if (lane < 4) s_st[lane] = 0;
  __syncthreads();
  const int He = A.net.He, Hm = A.net.Hm;
  const bool has_mu = Hm > 0;
  const int nrad = has_mu ? P + n : P;
  int rq_id[NQ];
#pragma unroll
  for (int sl = 0; sl < NQ; sl++) rq_id[sl] = ff_wadj_radius_id(n, lane + sl * NTHR, nrad);
  const bool own = lane < M;
  const int ai = own ? lane / D : 0, ci = own ? lane % D : 0;
  double* const myrow = A.rows + (int64_t)blockIdx.x * (3 * He + 3 * Hm);

  for (int64_t bq = blockIdx.x; bq < A.B; bq += gridDim.x) {
    const int64_t b = ff_opt_load(A.order, true, bq, A.z_in, (int32_t)bq);
    double y[NV] = {0.0, 0.0}, c0[NV] = {0.0, 0.0}, c1[NV] = {0.0, 0.0}, c2[NV] = {0.0, 0.0}, c3[NV] = {0.0, 0.0};
    double ad;
    y[0] = ff_opt_load(A.z_in, own, b * M + lane, A.z_in, 0.0);
    ff_wadj_seeds(A, b, M, lane, own, y[1], ad);
    // parameter integrands of this lane's units: k0 (the stage-0 value of the step under way) and the tentative sum
    double gk0[2][MAXU][3], tent[2][MAXU][3];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int j = 0; j < MAXU; j++)
#pragma unroll
        for (int c = 0; c < 3; c++) { gk0[t][j][c] = 0.0; tent[t][j][c] = 0.0; }
    ff_stepper S;
    S.begin(A.ta, A.tb, true);
    ff_dp5_ctl C;
    C.rtol = A.rtol; C.atol = A.atol; C.nt_inv = 1.0 / (2.0 * M); C.max_steps = A.max_steps;
    C.hwarm = ff_opt_load(A.h_init, true, A.h_scale < 0.0 ? 0 : b, A.z_in, 0.0) * fabs(A.h_scale);
    if (!(C.hwarm > 0.0)) C.hwarm = 0.0;
    C.h0v = 0.0; C.d1v = 0.0; C.hmax_acc = 0.0;
    const double hwarm0 = C.hwarm;
    int s = -2, nev = 0;
    auto wgt = [&](int v) -> double { return 1.0; };
    auto gsum = [&](double part) -> double { return ff_wadj_sum<NTHR>(s_red, s_red2, lane, part); };

#pragma unroll 1
    for (;;) {
      double gy, g0, g1, g2;
      ff_dp5_coeffs(s, S.h, C.h0v * S.dir, gy, g0, g1, g2);
      __syncthreads();
      if (own) {
        s_z[lane] = fma(g2, c2[0], fma(g1, c1[0], fma(g0, c0[0], gy * y[0])));
        s_kb[lane] = fma(g2, c2[1], fma(g1, c1[1], fma(g0, c0[1], gy * y[1])));
      }
      __syncthreads();
      // ------------------------------------------------------------------ radius phase: heads by direct evaluation
#pragma unroll 1
      for (int sl = 0; sl < NQ; sl++) {
        const int id = rq_id[sl];
        if (id < 0) break;
        const int a = id & 31, bq0 = (id >> 5) & 31, pr = id >> 10;
        const bool pair = bq0 != 31;
        double rho[D], dl[D], r2 = 0.0, al = 0.0;
#pragma unroll
        for (int c = 0; c < D; c++) {
          rho[c] = s_z[a * D + c] - (pair ? s_z[bq0 * D + c] : 0.0);
          dl[c] = s_kb[a * D + c] - (pair ? s_kb[bq0 * D + c] : 0.0);
          r2 = fma(rho[c], rho[c], r2);
          al = fma(dl[c], rho[c], al);
        }
        double r, ri, hd[3];
        ff_sqrt_rcp(r2, r, ri);
        ff_heads<3, true>(s_w[pair ? 0 : 1], s_e2, pair ? He : Hm, r, hd);
        s_q[pr][0] = r;
        s_q[pr][1] = pair ? -(al - 2.0 * D * ad) : -(al - D * ad);
        s_q[pr][2] = pair ? 2.0 * ad * r : ad * r;
        const double f0 = hd[0], f1 = hd[1], f2 = hd[2];
        const double F1 = f1 * (al * ri), gq = (pair ? 2.0 : 1.0) * fma(f2, r, (1.0 + D) * f1) * ri;
        double* Tq = &s_T[pr][0];
#pragma unroll
        for (int c = 0; c < D; c++) { Tq[c] = f0 * rho[c]; Tq[D + c] = fma(F1, rho[c], f0 * dl[c]); Tq[2 * D + c] = gq * rho[c]; }
      }
      __syncthreads();
      nev++;
      // ------------------------------------------------------------------ component phase
      double out[NV] = {0.0, 0.0};
      if (own) {
        double vi = 0.0, dvk = 0.0, gdi = 0.0;
        for (int j = 0; j < n; j++) {
          if (j == ai && !has_mu) continue;
          const double* Tq = &s_T[ff_wadj_partner(n, P, ai, j)][ci];
          const double sg = j < ai ? -1.0 : 1.0;
          vi = fma(sg, Tq[0], vi); dvk = fma(sg, Tq[D], dvk); gdi = fma(sg, Tq[2 * D], gdi);
        }
        out[0] = vi;
        out[1] = fma(ad, gdi, -dvk);
      }
      // ------------------------------------------------------------------ unit phase: parameter integrands of this stage
      double gcur[2][MAXU][3];
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int H = t ? Hm : He, q0 = t ? P : 0, q1 = t ? nrad : P;
#pragma unroll
        for (int j = 0; j < MAXU; j++) {
          double aw1 = 0.0, ab1 = 0.0, aw2 = 0.0;
          const int u = lane + NTHR * j;
          if (j * NTHR < H) {      // workgroup-uniform
            const ff_wtab wt = s_w[t][u < H ? u : 0];
            for (int q = q0; q < q1; q++) {
              const double r = s_q[q][0], ca = s_q[q][1], cb = s_q[q][2];
              const double sg = ff_sigmoid_sel<true>(fma(wt.w1, r, wt.b1), s_e2);
              const double s1 = sg * (1.0 - sg), s2 = s1 * fma(-2.0, sg, 1.0);
              aw2 = fma(ca, sg, fma(cb * wt.w1, s1, aw2));
              ab1 = fma(ca, s1, fma(cb * wt.w1, s2, ab1));
              aw1 = fma(ca * r, s1, fma(cb, fma(wt.w1 * r, s2, s1), aw1));
            }
            ab1 *= wt.w2; aw1 *= wt.w2;
          }
          gcur[t][j][0] = aw1; gcur[t][j][1] = ab1; gcur[t][j][2] = aw2;
        }
      }
      const int s_was = s;
      const double h_was = S.h;
      const int nacc_was = S.nacc;
      {
        // tentative quadrature of the step: h (B0 k0 + B2 k2 + B3 k3 + B4 k4 + B5 k5); k1 and k6 carry no weight
        const double bw = s_was == 2 ? FF_B2 : (s_was == 3 ? FF_B3 : (s_was == 4 ? FF_B4 : (s_was == 5 ? FF_B5 : 0.0)));
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int j = 0; j < MAXU; j++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
              if (s_was == -2 || s_was == 0) gk0[t][j][c] = gcur[t][j][c];
              if (s_was == 1) tent[t][j][c] = FF_B0 * gk0[t][j][c];       // a new attempt starts: drop what a rejected one left
              if (s_was >= 2 && s_was <= 5) tent[t][j][c] = fma(bw, gcur[t][j][c], tent[t][j][c]);
            }
      }
      s = ff_dp5_consume<NV>(s, S, C, y, c0, c1, c2, c3, out, wgt, gsum);
      if (s_was == 6 && S.nacc != nacc_was) {     // accepted: add the step's quadrature, k6 is the next k0
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int j = 0; j < MAXU; j++) {
            const int u = lane + NTHR * j, H = t ? Hm : He;
            if (u < H) {
#pragma unroll
              for (int c = 0; c < 3; c++) myrow[(t ? 3 * He : 0) + c * H + u] += h_was * tent[t][j][c];
            }
#pragma unroll
            for (int c = 0; c < 3; c++) gk0[t][j][c] = gcur[t][j][c];
          }
      }
      if (s == 99) break;
    }
    const double bad = S.fail ? __builtin_nan("") : 0.0;
    if (own && A.gx_out) A.gx_out[b * M + lane] = y[1] + bad;
    if (lane == 0) {
      if (S.fail) myrow[0] += bad;
      if (A.h_out) A.h_out[b] = C.hmax_acc > 0.0 ? C.hmax_acc : hwarm0;
      if (A.wcost) A.wcost[b] = S.nacc + S.nrej;
      if (A.stats) { atomicAdd(&s_st[0], nev); atomicMax(&s_st[1], S.nacc); atomicAdd(&s_st[2], S.nrej); if (S.fail) atomicMax(&s_st[3], 1); }
    }
    __syncthreads();
  }
  if (A.stats && lane == 0 && (s_st[0] || s_st[3])) {
    atomicAdd(&A.stats[0], s_st[0]);
    atomicMax(&A.stats[1], s_st[1]);
    atomicAdd(&A.stats[2], s_st[2]);
    if (s_st[3]) atomicMax(&A.stats[3], 1);
  }
}
