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

// Tabulated kernel.  Everything a radius leaves behind per stage lives in LDS, not in registers:
//   s_T[q]      its contributions to particle a's rows of v, (dv/dz)^T a_z and grad div (partner b reads them with the sign flipped);
//   s_dep[.][q] its deposit record (dr, ca, cb, node | net): buffer `cur` of the stage just evaluated, buffer `k0` of the
//               step's first stage (FSAL: the stage-6 record of an accepted step);
//   s_Tt        the TENTATIVE coefficient table of the step under way: stage s adds its records with weight h B_s as soon as it is
//               evaluated (B_1 = 0; the k0 records go in at stage 1, when h is known); an accepted step adds s_Tt to the
//               workgroup's permanent table -- a private region of A.trows in global memory (L2-resident: plain read-modify-writes
//               of the owning workgroup) -- a rejected one just clears it.
// Deposits are partitioned by NODE: wave w adds the records whose node j has j % W == w, every wave scanning all radii
// -- no two waves touch the same row, lanes of a wave add in lane order: a fixed summation order, bit-reproducible.
// A deposit beyond the LDS table (r >= 8: a few per sweep at 131 072 walkers of 20 particles) is parked in a short LDS list and
// goes to the launch's global overflow table (atomics, as in the narrow kernel) when its step is accepted; r >= 32 raises off_table.
// ~100 registers and 66 KB of LDS: two workgroups per CU (the register-resident version: 416 registers, one workgroup, 52 ms
// per 131 072 walkers of 20 particles).
struct __attribute__((aligned(16))) ff_wdep { double dr, ca, cb; int j, t; };
#define FF_WOVL 48      // deposits of one step on nodes beyond the LDS table that the overflow list holds (more: the direct kernel redoes the call)

template <int D, int W, int NQ>
__global__ void __launch_bounds__(FF_WAVE * W, 2)
ff_wide_adjtab_kernel(ff_adj_args A, int n) {
  constexpr int NV = 2, NTHR = FF_WAVE * W, RCAP = NQ * NTHR;
  constexpr int NE = 2 * FF_DEP_NLDS * FF_DEP_ROW;          // entries of a coefficient table
  const double* __restrict__ rtab = A.net.radial_table;
  if (!(rtab && rtab[3] == 0.0 && rtab[4] == 0.0)) return;   // the direct-evaluation kernel serves this call

  __shared__ double s_z[FF_WAVE], s_kb[FF_WAVE], s_red[NTHR], s_red2[16];
  __shared__ double s_T[RCAP][3 * D];
  __shared__ ff_wdep s_dep[2][RCAP];
  __shared__ double s_Tt[2][FF_DEP_NLDS][FF_DEP_LROW];
  __shared__ ff_wdep s_ovl[FF_WOVL];       // the step's deposits on nodes beyond the LDS table (r >= 8), weights folded in
  __shared__ int s_novl;
  __shared__ int s_st[4];

  const int lane = threadIdx.x, wv = lane / FF_WAVE;
  const int M = n * D, P = n * (n - 1) / 2;
  for (int e = lane; e < 2 * FF_DEP_NLDS * FF_DEP_LROW; e += NTHR) (&s_Tt[0][0][0])[e] = 0.0;
  for (int e = lane; e < RCAP * 3 * D; e += NTHR) (&s_T[0][0])[e] = 0.0;
  for (int e = lane; e < 2 * RCAP; e += NTHR) { ff_wdep z = {0.0, 0.0, 0.0, 0, -1}; (&s_dep[0][0])[e] = z; }
  if (lane < 4) s_st[lane] = 0;
  if (lane == 0) s_novl = 0;
  __syncthreads();
  const bool has_mu = A.net.Hm > 0;
  const int nrad = has_mu ? P + n : P;
  const double tab_inv_h = rtab[0], tab_h = rtab[1];
  const int dep_nrow = FF_UNIFORM(rtab[5] >= 6.0 && rtab[5] <= (double)FF_DEP_ROW ? (int)rtab[5] : FF_DEP_ROW);   // coefficients per deposit row
  int rq_id[NQ];
#pragma unroll
  for (int sl = 0; sl < NQ; sl++) rq_id[sl] = ff_wadj_radius_id(n, lane + sl * NTHR, nrad);
  double* const ovf = A.trows + (size_t)gridDim.x * NE;      // nodes beyond the LDS tables: one global table [2][NTOT][ROW], atomics
  // coordinate p = lane / 4: its four lanes split the partners of p's particle (quad reduction); lane 4 p owns z_p, a_p
  const int rp = lane >> 2, rs = lane & 3;
  const bool rowlane = rp < M;
  const bool own = rowlane && rs == 0;
  const int ai = rowlane ? rp / D : 0, ci = rowlane ? rp % D : 0;
  bool off_any = false;
  double* const Wg = A.trows + (size_t)blockIdx.x * NE;      // this workgroup's permanent table [2][NLDS][ROW]

  for (int64_t bq = blockIdx.x; bq < A.B; bq += gridDim.x) {
    const int64_t b = ff_opt_load(A.order, true, bq, A.z_in, (int32_t)bq);
    double y[NV] = {0.0, 0.0}, c0[NV] = {0.0, 0.0}, c1[NV] = {0.0, 0.0}, c2[NV] = {0.0, 0.0}, c3[NV] = {0.0, 0.0};
    double ad;
    y[0] = ff_opt_load(A.z_in, own, b * M + rp, A.z_in, 0.0);
    ff_wadj_seeds(A, b, M, rp, own, y[1], ad);
    ff_stepper S;
    S.begin(A.ta, A.tb, true);
    ff_dp5_ctl C;
    C.rtol = A.rtol; C.atol = A.atol; C.nt_inv = 1.0 / (2.0 * M); C.max_steps = A.max_steps;
    C.hwarm = ff_opt_load(A.h_init, true, A.h_scale < 0.0 ? 0 : b, A.z_in, 0.0) * fabs(A.h_scale);      // (ff_ode.walker_h_equal: rounded by ff_open_steps_kernel in front of the launch)
    if (!(C.hwarm > 0.0)) C.hwarm = 0.0;
    C.h0v = 0.0; C.d1v = 0.0; C.hmax_acc = 0.0;
    const double hwarm0 = C.hwarm;
    int s = -2, nev = 0, cur = 0;          // s_dep[cur]: records of the stage being evaluated; s_dep[cur ^ 1]: the k0 records
    auto wgt = [&](int v) -> double { return 1.0; };
    auto gsum = [&](double part) -> double { return ff_wadj_sum<NTHR>(s_red, s_red2, lane, part); };
    // adds the records of buffer `buf` into the tentative table with weight w: this wave takes the nodes j % W == wv
    auto deposit = [&](int buf, double w) {
      for (int q0 = 0; q0 < nrad; q0 += FF_WAVE) {
        const int q = q0 + (lane & (FF_WAVE - 1));
        if (q < nrad) {
          const ff_wdep rc = s_dep[buf][q];
          if (rc.t >= 0 && (rc.j % W) == wv) {
            if (rc.j >= FF_DEP_NLDS) {      // beyond the LDS table (rare): parked until the step is accepted
              const int pos = atomicAdd(&s_novl, 1);
              if (pos < FF_WOVL) { ff_wdep o = rc; o.ca *= w; o.cb *= w; s_ovl[pos] = o; }
            } else {
              // (dep_nrow coefficients: what the launch's weights need -- header slot 5 of the radial table, ff_radial.h)
              double pk = 1.0, pm = 0.0;
              double* row = &s_Tt[rc.t][rc.j][0];
#pragma unroll
              for (int k = 0; k < FF_DEP_ROW; k++) {
                if (k < 6 || k < dep_nrow) atomicAdd(row + k, w * fma(rc.ca, pk, rc.cb * pm));
                pm = pk;
                pk = pk * rc.dr * (1.0 / (k + 1));
              }
            }
          }
        }
      }
    };

    // one evaluation, instantiated per stage; returns true when the walker has finished
    auto evaluate = [&](auto stage_tag) -> bool {
      constexpr int SG = decltype(stage_tag)::value;
      if constexpr (SG == FF_STAGE_DYN) FF_ASSUME(s <= 0);
      const int sv = SG == FF_STAGE_DYN ? s : SG;
      double gy, g0, g1, g2;
      ff_dp5_coeffs(sv, S.h, C.h0v * S.dir, gy, g0, g1, g2);
      __syncthreads();
      if (own) {
        s_z[rp] = fma(g2, c2[0], fma(g1, c1[0], fma(g0, c0[0], gy * y[0])));
        s_kb[rp] = fma(g2, c2[1], fma(g1, c1[1], fma(g0, c0[1], gy * y[1])));
      }
      __syncthreads();
      // ------------------------------------------------------------------ radius phase (branch-free; a slot without a radius
      // works on particle 0 and writes row lane + NTHR sl, which nobody reads)
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
        const bool on = jf <= (double)(FF_DEP_NTOT - 1);           // (false for NaN too)
        if (act && (!ok || !on)) off_any = true;                    // beyond either table: the direct kernel redoes the call
        ff_wdep rc;
        rc.j = on ? (int)jf : 0;
        rc.t = (act && on) ? (pair ? 0 : 1) : -1;
        rc.dr = fma(-(double)rc.j, 1.0 / FF_DEP_INVH, r);
        rc.ca = pair ? -(al - 2.0 * D * ad) : -(al - D * ad);
        rc.cb = pair ? 2.0 * ad * r : ad * r;
        s_dep[cur][pr] = rc;
        const double f0 = hd[0], f1 = hd[1], f2 = hd[2];
        const double F1 = f1 * (al * ri), gq = (pair ? 2.0 : 1.0) * fma(f2, r, (1.0 + D) * f1) * ri;
        double* Tq = &s_T[pr][0];
#pragma unroll
        for (int c = 0; c < D; c++) { Tq[c] = f0 * rho[c]; Tq[D + c] = fma(F1, rho[c], f0 * dl[c]); Tq[2 * D + c] = gq * rho[c]; }
      }
      __syncthreads();
      nev++;
      // ------------------------------------------------------------------ component phase: four lanes per coordinate
      double vi = 0.0, dvk = 0.0, gdi = 0.0;
      if (rowlane) {
        for (int j = rs; j < n; j += 4) {
          if (j == ai && !has_mu) continue;
          const double* Tq = &s_T[ff_wadj_partner(n, P, ai, j)][ci];
          const double sg = j < ai ? -1.0 : 1.0;
          vi = fma(sg, Tq[0], vi); dvk = fma(sg, Tq[D], dvk); gdi = fma(sg, Tq[2 * D], gdi);
        }
      }
      vi += ff_swap1(vi); vi += ff_swap2(vi);
      dvk += ff_swap1(dvk); dvk += ff_swap2(dvk);
      gdi += ff_swap1(gdi); gdi += ff_swap2(gdi);
      double out[NV] = {0.0, 0.0};
      if (own) { out[0] = vi; out[1] = fma(ad, gdi, -dvk); }
      // ------------------------------------------------------------------ this stage's share of the step's quadrature
      const int s_was = sv;
      const double h_was = S.h;
      const int nacc_was = S.nacc;
      if (s_was == 1) deposit(cur ^ 1, h_was * FF_B0);            // the step starts: its k0 records, now that h is known
      else if (s_was == 2) deposit(cur, h_was * FF_B2);
      else if (s_was == 3) deposit(cur, h_was * FF_B3);
      else if (s_was == 4) deposit(cur, h_was * FF_B4);
      else if (s_was == 5) deposit(cur, h_was * FF_B5);
      s = ff_dp5_consume<NV>(sv, S, C, y, c0, c1, c2, c3, out, wgt, gsum);      // (its reductions are workgroup barriers)
      if (s_was == -2 || s_was == 0) cur ^= 1;                      // f(y): these records are the step's k0
      if (s_was == 6) {
        const bool acc = S.nacc != nacc_was;                         // workgroup-uniform
        for (int e = lane; e < NE; e += NTHR) {
          double* tt = &(&s_Tt[0][0][0])[(e / FF_DEP_ROW) * FF_DEP_LROW + e % FF_DEP_ROW];
          if (acc) Wg[e] += *tt;
          *tt = 0.0;
        }
        __syncthreads();
        const int novl = s_novl;
        if (novl > FF_WOVL) off_any = true;                          // more far deposits than the list holds: redo by the direct kernel
        if (acc && lane < novl && lane < FF_WOVL) {
          const ff_wdep rc = s_ovl[lane];
          double pk = 1.0, pm = 0.0;
          double* row = ovf + ((size_t)rc.t * FF_DEP_NTOT + rc.j) * FF_DEP_ROW;
#pragma unroll
          for (int k = 0; k < FF_DEP_ROW; k++) {
            atomicAdd(row + k, fma(rc.ca, pk, rc.cb * pm));
            pm = pk;
            pk = pk * rc.dr * (1.0 / (k + 1));
          }
        }
        __syncthreads();
        if (lane == 0) s_novl = 0;
        if (acc) cur ^= 1;                                           // FSAL: the stage-6 records open the next step
      }
      return s == 99;
    };
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
    const double bad = S.fail ? __builtin_nan("") : 0.0;   // failed integration -> NaN gradients
    if (own && A.gx_out) A.gx_out[b * M + rp] = y[1] + bad;
    if (lane == 0) {
      if (S.fail) Wg[0] += bad;
      if (A.h_out) A.h_out[b] = C.hmax_acc > 0.0 ? C.hmax_acc : hwarm0;
      if (A.wcost) A.wcost[b] = S.nacc + S.nrej;
      if (A.stats) { atomicAdd(&s_st[0], nev); atomicMax(&s_st[1], S.nacc); atomicAdd(&s_st[2], S.nrej); if (S.fail) atomicMax(&s_st[3], 1); }
    }
    __syncthreads();
  }
  if (off_any) *A.off_table = 1.0;
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
// read-modify-writes); ff_adj_reduce_kernel sums the rows.
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
    C.hwarm = ff_opt_load(A.h_init, true, A.h_scale < 0.0 ? 0 : b, A.z_in, 0.0) * fabs(A.h_scale);      // (ff_ode.walker_h_equal: rounded by ff_open_steps_kernel in front of the launch)
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
