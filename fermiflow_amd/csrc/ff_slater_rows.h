// ff_slater_rows.h -- HO3D orbital helpers and the SIXTEEN-LANES-PER-DETERMINANT Slater table of the local-energy finish, as a
// device routine shared by its stand-alone kernel (ff_ho3d.hip: ff_eloc_slater_rows_kernel) and the fused epilogue of the
// one-walker-per-workgroup local-energy kernel (ff_wide.hip).  Reference: src/slater.py:13-62 (D_ij = phi_j(r_i), d log|det| =
// tr(D^-1 dD)); three dimensions have no upstream counterpart (src/orbitals.py:56).
#pragma once
#include "ff_common.h"
#include "ff_slater.h"

#define FF_HO3D_NORB 120        // shells 0..7 (Hermite degrees 0..7)
#define FF_PI_M34 0.42377720812375763   // pi^(-3/4)

FF_D void ff_ho3d_decode(int k, int& nx, int& ny, int& nz) {
  int shell = 0;
  while ((shell + 1) * (shell + 2) * (shell + 3) / 6 <= k) shell++;
  int idx = k - shell * (shell + 1) * (shell + 2) / 6;
  nx = ny = nz = 0;
  for (int a = 0; a <= shell; a++) {
    const int cnt = shell - a + 1;
    if (idx < cnt) { nx = a; ny = idx; nz = shell - a - idx; return; }
    idx -= cnt;
  }
}

// phi_k at r and optionally its gradient (3) and Laplacian
template <bool DERIV>
FF_D void ff_orbital3d(int k, const double* r, double gauss /* pi^-3/4 exp(-r^2/2) */, double& v, double* g, double& lap) {
  int n[3];
  ff_ho3d_decode(k, n[0], n[1], n[2]);
  double h[3], h1[3], h2[3];
#pragma unroll
  for (int c = 0; c < 3; c++) ff_herm<DERIV>(n[c], r[c], h[c], h1[c], h2[c]);
  v = gauss * h[0] * h[1] * h[2];
  if (DERIV) {
    double p1[3], p2[3];   // (e^{-x^2/2} h)' / e^{-x^2/2}, (e^{-x^2/2} h)'' / e^{-x^2/2}
#pragma unroll
    for (int c = 0; c < 3; c++) {
      p1[c] = h1[c] - r[c] * h[c];
      p2[c] = h2[c] - 2.0 * r[c] * h1[c] + (r[c] * r[c] - 1.0) * h[c];
    }
    g[0] = gauss * p1[0] * h[1] * h[2]; g[1] = gauss * h[0] * p1[1] * h[2]; g[2] = gauss * h[0] * h[1] * p1[2];
    lap = gauss * (p2[0] * h[1] * h[2] + h[0] * p2[1] * h[2] + h[0] * h[1] * p2[2]);
  }
}

FF_D double ff_gauss3d(const double* r) { return FF_PI_M34 * exp(-0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2])); }

template <bool DERIV>
FF_D void ff_orbital3d_hess(int k, const double* r, double gauss, double& v, double* g, double* hs) {
  int n[3];
  ff_ho3d_decode(k, n[0], n[1], n[2]);
  double h[3], h1[3], h2[3];
#pragma unroll
  for (int c = 0; c < 3; c++) ff_herm<true>(n[c], r[c], h[c], h1[c], h2[c]);
  double p1[3], p2[3];
#pragma unroll
  for (int c = 0; c < 3; c++) { p1[c] = h1[c] - r[c] * h[c]; p2[c] = h2[c] - 2.0 * r[c] * h1[c] + (r[c] * r[c] - 1.0) * h[c]; }
  v = gauss * h[0] * h[1] * h[2];
  g[0] = gauss * p1[0] * h[1] * h[2]; g[1] = gauss * h[0] * p1[1] * h[2]; g[2] = gauss * h[0] * h[1] * p1[2];
  hs[0] = gauss * p2[0] * h[1] * h[2]; hs[1] = gauss * p1[0] * p1[1] * h[2]; hs[2] = gauss * p1[0] * h[1] * p1[2];
  hs[3] = gauss * h[0] * p2[1] * h[2]; hs[4] = gauss * h[0] * p1[1] * p1[2]; hs[5] = gauss * h[0] * h[1] * p2[2];
}

// Scratch of the routine for NG groups of sixteen lanes
template <int NG>
struct ff_slater_rows_smem {
  double row[NG][2 * FF_MAX_NS];            // the pivot row of the step: [A | Inv]
  double inv[NG][FF_MAX_NS][FF_MAX_NS + 1];  // D^-1 by rows
  int orb[NG][FF_MAX_NS];
};

// Lane r = lane & 15 of group grp owns particle r of the group's species -- row r of D_ij = phi_j(r_i) and of the unit matrix beside
// it.  Gauss-Jordan with partial pivoting, without moving rows: per column the unused lane with the largest entry becomes the pivot
// (16-lane butterfly), normalises its row, publishes it through LDS, everyone else eliminates; the lane that was the pivot of column
// c ends up holding row c of D^-1.  Then lane a forms its particle's rows of T[comp][a][b] = sum_j d_comp phi_j(r_a) Dinv[j][b]
// and the same-particle Hessian sums S.  Every thread of the workgroup must call it (it contains workgroup barriers);
// threads with live = false (and whole waves beyond the groups) take part in the barriers and lane exchanges only.
// zw: the walker's coordinates (n, D); q: its slots [0,M) g0 | [M, M + NH2 n) S | T_up (D nup^2, component-major) | T_dn | 2 log|det|
// per spin -- both may point to global memory or LDS.  `lane`: the hardware lane (0..63).
template <int D, int NG>
FF_D void ff_slater_rows_body(ff_slater_rows_smem<NG>& sm, int lane, int grp, bool live, int sp, int nup, int ndn,
                              const int* __restrict__ tab_up, const int* __restrict__ tab_dn, int st, const double* zw, double* q) {
  constexpr int NS = FF_MAX_NS, NH2 = D * (D + 1) / 2;
  auto& s_row = sm.row; auto& s_inv = sm.inv; auto& s_orb = sm.orb;
  const int r = lane & 15;
  const int n = nup + ndn, M = D * n;
  const int ns = live ? (sp ? ndn : nup) : 0, off = sp ? nup : 0;
  const int64_t nq = M + NH2 * n + D * (nup * nup + ndn * ndn) + 2;
  const bool mine = r < ns;
  if (mine) s_orb[grp][r] = ((sp ? tab_dn : tab_up) + st * ns)[r];
  __syncthreads();
  double x[D];
#pragma unroll
  for (int c = 0; c < D; c++) x[c] = mine ? zw[D * (off + r) + c] : 0.25 * (c + 1 + r);
  double gs;
  if constexpr (D == 2) gs = ff_gauss2d(x[0], x[1]); else gs = ff_gauss3d(x);
  double A[NS], Inv[NS];
#pragma unroll
  for (int j = 0; j < NS; j++) {
    double v = (j == r) ? 1.0 : 0.0;      // rows / columns beyond ns: the unit matrix (never chosen as pivots, eliminated with f = 0)
    if (j < ns && mine) {
      if constexpr (D == 2) ff_orbital<false>(s_orb[grp][j], x[0], x[1], gs, v, nullptr, nullptr);
      else { double lp; ff_orbital3d<false>(s_orb[grp][j], x, gs, v, nullptr, lp); }
    }
    A[j] = v;
    Inv[j] = (j == r) ? 1.0 : 0.0;
  }
  double acc = 0.0;
  bool used = !mine;
  int myrow = -1;
  const int nsmax = nup > ndn ? nup : ndn;      // (kernel-uniform: columns beyond both determinants are skipped by a scalar branch)
#pragma unroll
  for (int c = 0; c < NS; c++) {
    if (c >= nsmax) break;
    const bool act = c < ns;      // (uniform within the group)
    // pivot: the unused lane with the largest |A[c]| (ties: the lower lane)
    double best = (!used && act) ? fabs(A[c]) : -1.0;
    int who = r;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
      const double ob = ff_lane_read(best, lane ^ m);
      const int ow = __builtin_amdgcn_ds_bpermute((lane ^ m) << 2, who);
      const bool take = ob > best || (ob == best && ow < who);
      best = take ? ob : best;
      who = take ? ow : who;
    }
    const bool ispiv = act && who == r && !used;
    if (ispiv) {
      const double ip = 1.0 / A[c];
      acc = log(fabs(A[c]));
#pragma unroll
      for (int j = 0; j < NS; j++) { A[j] *= ip; Inv[j] *= ip; s_row[grp][j] = A[j]; s_row[grp][NS + j] = Inv[j]; }
      used = true;
      myrow = c;
    }
    __syncthreads();
    if (act && !ispiv) {
      const double f = A[c];
#pragma unroll
      for (int j = 0; j < NS; j++) { A[j] = fma(-f, s_row[grp][j], A[j]); Inv[j] = fma(-f, s_row[grp][NS + j], Inv[j]); }
    }
    __syncthreads();
  }
  if (myrow >= 0) {
#pragma unroll
    for (int j = 0; j < NS; j++) s_inv[grp][myrow][j] = Inv[j];
  }
  // log|det| = sum of the pivots' logs (each pivot lane holds one of them)
  double lsum = (myrow >= 0) ? acc : 0.0;
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) lsum += ff_lane_read(lsum, lane ^ m);
  __syncthreads();
  if (live && ns == 0 && r == 0) q[nq - 2 + sp] = 0.0;
  if (mine) {
    double T[D][NS], S[NH2], gd[D];
#pragma unroll
    for (int c = 0; c < D; c++) {
      gd[c] = 0.0;
#pragma unroll
      for (int j = 0; j < NS; j++) T[c][j] = 0.0;
    }
#pragma unroll
    for (int e = 0; e < NH2; e++) S[e] = 0.0;
    for (int j = 0; j < ns; j++) {
      double v, g[D], hs[NH2];
      if constexpr (D == 2) ff_orbital<true>(s_orb[grp][j], x[0], x[1], gs, v, g, hs);
      else ff_orbital3d_hess<true>(s_orb[grp][j], x, gs, v, g, hs);
      const double da = s_inv[grp][j][r];
#pragma unroll
      for (int bb = 0; bb < NS; bb++) {
        const double di = s_inv[grp][j][bb];
#pragma unroll
        for (int c = 0; c < D; c++) T[c][bb] = fma(g[c], di, T[c][bb]);
      }
#pragma unroll
      for (int c = 0; c < D; c++) gd[c] = fma(g[c], da, gd[c]);
#pragma unroll
      for (int e = 0; e < NH2; e++) S[e] = fma(hs[e], da, S[e]);
    }
#pragma unroll
    for (int c = 0; c < D; c++) q[D * (off + r) + c] = 2.0 * gd[c];        // g0 = 2 grad log|det|
#pragma unroll
    for (int e = 0; e < NH2; e++) q[M + NH2 * (off + r) + e] = S[e];
    double* Tq = q + M + NH2 * n + (sp ? D * nup * nup : 0);                // [comp][a][b]
#pragma unroll
    for (int c = 0; c < D; c++)
#pragma unroll
      for (int bb = 0; bb < NS; bb++) {
        if (bb < ns) Tq[c * ns * ns + r * ns + bb] = T[c][bb];
      }
    if (r == 0) q[nq - 2 + sp] = 2.0 * lsum;
  }
}

// =====================================================================================================================
// The same Slater table built by a WHOLE WORKGROUP for ONE walker (the fused finish of the one-walker-per-workgroup local-energy
// kernel, ff_wide.hip): inside that kernel the sixteen-lane routine above is a serial 36 k-cycle chain with three of four waves
// idle (14 ms per 131 072 walkers at configs[4]); here every (particle, orbital) pair, every entry of T and S is its own thread's
// task, Hermite functions with both derivatives come from the recurrence (no coefficient-table loads), and only the
// Gauss-Jordan elimination itself stays on sixteen lanes per species (wave 0).
// LDS scratch: orbital values, gradients and Hessian entries of every (species, particle, orbital): (1 + D + NH2) NSQ doubles.
template <int D>
struct ff_slater_wg_dims { static constexpr int NH2 = D * (D + 1) / 2, PER = 1 + D + NH2; };

// zw: the walker's coordinates in LDS; q: its Slater slots in LDS (layout as above); so: scratch of PER * (nup^2 + ndn^2) doubles;
// sdeg: D * (nup + ndn) ints; sm: the elimination's scratch.  Every thread of the workgroup calls it.
// Hermite degrees of every orbital of many-body state st -> sdeg (D (nup + ndn) ints); the caller synchronises.  With one orbital set
// for all walkers (walker_state NULL: the ground-state runs) once per kernel is enough.
template <int D>
FF_D void ff_slater_wg_degrees(int* __restrict__ sdeg, int tid, int nup, int ndn, const int* __restrict__ tab_up,
                               const int* __restrict__ tab_dn, int st) {
  if (tid < nup + ndn) {
    const int sp = tid >= nup ? 1 : 0, j = sp ? tid - nup : tid, ns = sp ? ndn : nup;
    const int k = ((sp ? tab_dn : tab_up) + st * ns)[j];
    int dg[3] = {0, 0, 0};
    if constexpr (D == 2) ff_orb_decode(k, dg[0], dg[1]); else ff_ho3d_decode(k, dg[0], dg[1], dg[2]);
#pragma unroll
    for (int c = 0; c < D; c++) sdeg[D * tid + c] = dg[c];
  }
}

template <int D, int NTHR>
FF_D void ff_slater_table_wg(ff_slater_rows_smem<2>& sm, double* __restrict__ so, const int* __restrict__ sdeg, int tid, int nup, int ndn,
                             const double* zw, double* q) {
  constexpr int NS = FF_MAX_NS, NH2 = D * (D + 1) / 2, PER = 1 + D + NH2;
  const int n = nup + ndn, M = D * n, npu = nup * nup, npt = npu + ndn * ndn;
  const int oS = M, oT = M + NH2 * n, oL = oT + D * npt;
  // (1) orbital j at particle a: value, gradient, Hessian (upper triangle) -- one task per (species, a, j)
  for (int e = tid; e < npt; e += NTHR) {
    const int sp = e >= npu ? 1 : 0, ee = sp ? e - npu : e, ns = sp ? ndn : nup, off = sp ? nup : 0;
    const int a = ee / ns, j = ee - a * ns;
    double h[D], p1[D], p2[D], r2 = 0.0;
#pragma unroll
    for (int c = 0; c < D; c++) {
      const double xc = zw[D * (off + a) + c];
      const int dg = sdeg[D * (off + j) + c];
      double hv, h1, h2;
      ff_herm_one_d2(dg, xc, hv, h1, h2);
      h[c] = hv;
      p1[c] = fma(-xc, hv, h1);                                            // (e^{-x^2/2} h)' / e^{-x^2/2}
      p2[c] = fma(fma(xc, xc, -1.0), hv, fma(-2.0 * xc, h1, h2));           // second derivative likewise
      r2 = fma(xc, xc, r2);
    }
    const double gs = (D == 2 ? FF_PI_SQRT_INV : FF_PI_M34) * ff_exp(fmax(-0.5 * r2, -708.0));
    double* o = so + (size_t)e * PER;
    if constexpr (D == 2) {
      o[0] = gs * h[0] * h[1];
      o[1] = gs * p1[0] * h[1]; o[2] = gs * h[0] * p1[1];
      o[3] = gs * p2[0] * h[1]; o[4] = gs * p1[0] * p1[1]; o[5] = gs * h[0] * p2[1];
    } else {
      o[0] = gs * h[0] * h[1] * h[2];
      o[1] = gs * p1[0] * h[1] * h[2]; o[2] = gs * h[0] * p1[1] * h[2]; o[3] = gs * h[0] * h[1] * p1[2];
      o[4] = gs * p2[0] * h[1] * h[2]; o[5] = gs * p1[0] * p1[1] * h[2]; o[6] = gs * p1[0] * h[1] * p1[2];
      o[7] = gs * h[0] * p2[1] * h[2]; o[8] = gs * h[0] * p1[1] * p1[2]; o[9] = gs * h[0] * h[1] * p2[2];
    }
  }
  __syncthreads();
  // (2) Gauss-Jordan with partial pivoting on sixteen lanes per species, rows from LDS; D^-1 by rows -> sm.inv.  Both species live
  // in wave 0, so the pivot row's trip through LDS needs the wave's own ordering only (FF_WAVE_SYNC) -- the other waves skip the
  // elimination and its twenty barriers altogether and wait at the one below.
  if (tid < 64) {
    const int lane = tid & 63, grp = tid < 32 ? (tid >> 4) : 0, r = lane & 15;
    const bool live = tid < 32;
    const int sp = (tid >> 4) & 1;
    const int ns = live ? (sp ? ndn : nup) : 0;
    const bool mine = r < ns;
    double A[NS], Inv[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
      A[j] = (j < ns && mine) ? so[(size_t)((sp ? npu : 0) + r * ns + j) * PER] : ((j == r) ? 1.0 : 0.0);
      Inv[j] = (j == r) ? 1.0 : 0.0;
    }
    double acc = 0.0;
    bool used = !mine;
    int myrow = -1;
    const int nsmax = nup > ndn ? nup : ndn;
#pragma unroll
    for (int c = 0; c < NS; c++) {
      if (c >= nsmax) break;
      const bool act = c < ns;
      // pivot: the unused lane with the largest |A[c]| -- ONE 32-bit DPP maximum over the group's 16 lanes: |entry| rounded to
      // float with the lane index in its low four bits (equal keys: the lower lane; a pivot within 2^-19 of the largest entry
      // instead of the largest changes the rounding of the inverse, not its value -- as in ff_mcmc_rows_kernel)
      unsigned key = (!used && act) ? ((__float_as_uint((float)fabs(A[c])) & ~15u) | (unsigned)(15 - r)) : 0u;
      {
        unsigned o = (unsigned)__builtin_amdgcn_mov_dpp((int)key, 0xB1, 0xF, 0xF, true); key = o > key ? o : key;      // lane ^ 1
        o = (unsigned)__builtin_amdgcn_mov_dpp((int)key, 0x4E, 0xF, 0xF, true); key = o > key ? o : key;               // lane ^ 2
        o = (unsigned)__builtin_amdgcn_mov_dpp((int)key, 0x124, 0xF, 0xF, true); key = o > key ? o : key;              // row_ror:4
        o = (unsigned)__builtin_amdgcn_mov_dpp((int)key, 0x128, 0xF, 0xF, true); key = o > key ? o : key;              // row_ror:8
      }
      const int who = 15 - (int)(key & 15u);
      const bool ispiv = act && who == r && !used;
      if (ispiv) {
        const double ip = 1.0 / A[c];
        acc = log(fabs(A[c]));
#pragma unroll
        for (int j = 0; j < NS; j++) { A[j] *= ip; Inv[j] *= ip; sm.row[grp][j] = A[j]; sm.row[grp][NS + j] = Inv[j]; }
        used = true;
        myrow = c;
      }
      FF_WAVE_SYNC();
      if (act && !ispiv) {
        const double f = A[c];
#pragma unroll
        for (int j = 0; j < NS; j++) { A[j] = fma(-f, sm.row[grp][j], A[j]); Inv[j] = fma(-f, sm.row[grp][NS + j], Inv[j]); }
      }
      FF_WAVE_SYNC();
    }
    if (myrow >= 0) {
#pragma unroll
      for (int j = 0; j < NS; j++) sm.inv[grp][myrow][j] = Inv[j];
    }
    double lsum = (myrow >= 0) ? acc : 0.0;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) lsum += ff_lane_read(lsum, lane ^ m);
    if (live && r == 0) q[oL + sp] = ns ? 2.0 * lsum : 0.0;
  }
  __syncthreads();
  // (3) T[comp][a][b] = sum_j d_comp phi_j(r_a) Dinv[j][b]  and  S[a][e] = sum_j hess_e phi_j(r_a) Dinv[j][a]: one task per entry
  for (int e = tid; e < (D + 0) * npt; e += NTHR) {
    const int c = e / npt, e2 = e - c * npt;
    const int sp = e2 >= npu ? 1 : 0, ee = sp ? e2 - npu : e2, ns = sp ? ndn : nup, off = sp ? nup : 0;
    const int a = ee / ns, bb = ee - a * ns;
    const double* oa = so + (size_t)((sp ? npu : 0) + a * ns) * PER + 1 + c;
    double t = 0.0;
    for (int j = 0; j < ns; j++) t = fma(oa[(size_t)j * PER], sm.inv[sp][j][bb], t);
    q[oT + (sp ? D * npu : 0) + c * ns * ns + a * ns + bb] = t;
    if (a == bb) q[D * (off + a) + c] = 2.0 * t;                            // g0 = 2 grad log|det|
  }
  for (int e = tid; e < NH2 * n; e += NTHR) {
    const int ag = e / NH2, he = e - ag * NH2;
    const int sp = ag >= nup ? 1 : 0, a = sp ? ag - nup : ag, ns = sp ? ndn : nup;
    const double* oa = so + (size_t)((sp ? npu : 0) + a * ns) * PER + 1 + D + he;
    double t = 0.0;
    for (int j = 0; j < ns; j++) t = fma(oa[(size_t)j * PER], sm.inv[sp][j][a], t);
    q[oS + NH2 * ag + he] = t;
  }
  __syncthreads();
}
