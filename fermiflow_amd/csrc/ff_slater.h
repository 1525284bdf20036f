// ff_slater.h -- HO2D orbitals and Slater-determinant device routines.
//
// Reference semantics: HO2D orbitals  src/orbitals.py:65-82; LogAbsSlaterDet  src/slater.py:13-62;
// FreeFermion.log_prob  src/base_dist.py:49-56.  Derivatives are closed-form (Hermite polynomials are
// differentiated analytically; d log|det D| = tr(D^-1 dD)) instead of the reference's autograd passes.
#pragma once
#include "ff_common.h"

// h_n(x) = FF_HERM_NORM[n] * sum_k FF_HERM_A[n][k] x^k, the polynomials of src/orbitals.py:66-73
__constant__ double FF_HERM_A[8][8] = {
    {1, 0, 0, 0, 0, 0, 0, 0},
    {0, 1, 0, 0, 0, 0, 0, 0},
    {-1, 0, 2, 0, 0, 0, 0, 0},
    {0, -3, 0, 2, 0, 0, 0, 0},
    {1.5, 0, -6, 0, 2, 0, 0, 0},
    {0, 7.5, 0, -10, 0, 2, 0, 0},
    {-1.25, 0, 7.5, 0, -5, 0, 2.0 / 3.0, 0},
    {0, -17.5, 0, 35, 0, -14, 0, 4.0 / 3.0}};
__constant__ double FF_HERM_NORM[8] = {
    1.0, 1.4142135623730951 /* sqrt(2) */, 0.70710678118654746 /* 1/sqrt(2) */, 0.57735026918962584 /* 1/sqrt(3) */,
    0.40824829046386307 /* 1/sqrt(6) */, 0.25819888974716110 /* 1/sqrt(15) */, 0.44721359549995793 /* 1/sqrt(5) */,
    0.11952286093343936 /* 1/sqrt(70) */};
#define FF_PI_SQRT_INV 0.56418958354775628

// orbital index k -> (nx, ny): list order "for n in range(8) for nx in range(n+1): (nx, n-nx)"
FF_D void ff_orb_decode(int k, int& nx, int& ny) {
  int shell = 0;
  while ((shell + 1) * (shell + 2) / 2 <= k) shell++;
  nx = k - shell * (shell + 1) / 2;
  ny = shell - nx;
}

// h_n(x) and (optionally) its first two derivatives, Horner on the coefficient table
template <bool DERIV>
FF_D void ff_herm(int n, double x, double& h, double& h1, double& h2) {
  double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
  for (int k = 7; k >= 0; k--) {
    double ck = FF_HERM_A[n][k];
    a = fma(a, x, ck);
    if (DERIV) {
      if (k >= 1) b = fma(b, x, ck * k);
      if (k >= 2) c = fma(c, x, ck * (k * (k - 1)));
    }
  }
  double nm = FF_HERM_NORM[n];
  h = nm * a;
  if (DERIV) { h1 = nm * b; h2 = nm * c; }
}

// phi_k at (x,y) and optionally gradient (2) and Hessian (xx, xy, yy)
template <bool DERIV>
FF_D void ff_orbital(int k, double x, double y, double gauss /* pi^-1/2 exp(-r^2/2) */, double& v, double* g, double* hs) {
  int nx, ny;
  ff_orb_decode(k, nx, ny);
  double hx, hx1, hx2, hy, hy1, hy2;
  ff_herm<DERIV>(nx, x, hx, hx1, hx2);
  ff_herm<DERIV>(ny, y, hy, hy1, hy2);
  v = gauss * hx * hy;
  if (DERIV) {
    double px1 = hx1 - x * hx, py1 = hy1 - y * hy;
    double px2 = hx2 - 2.0 * x * hx1 + (x * x - 1.0) * hx, py2 = hy2 - 2.0 * y * hy1 + (y * y - 1.0) * hy;
    g[0] = gauss * px1 * hy; g[1] = gauss * hx * py1;
    hs[0] = gauss * px2 * hy; hs[1] = gauss * px1 * py1; hs[2] = gauss * hx * py2;
  }
}

FF_D double ff_gauss2d(double x, double y) { return FF_PI_SQRT_INV * exp(-0.5 * (x * x + y * y)); }
// same, with the in-house exp (argument clamped: exp(-708) is already a denormal-free 3e-308)
FF_D double ff_gauss2d_fast(double x, double y) { return FF_PI_SQRT_INV * ff_exp(fmax(-0.5 * (x * x + y * y), -708.0)); }

// --------------------------------------------------------------------------------------------------
// Register-resident log|det| for compile-time NS (MCMC hot loop).  The normalised Hermite function of degree n
// comes from the three-term recurrence
//   h_0 = 1, h_1 = sqrt(2) x, h_{m+1} = sqrt(2/(m+1)) x h_m - sqrt(m/(m+1)) h_{m-1}
// (same polynomials as src/orbitals.py:66-73).  LU with partial pivoting; all indices static, row exchange by
// predicated swaps, so the determinant stays in VGPRs.
// (no table of all degrees + select: hipcc turns such a select chain into a runtime-indexed private array,
//  i.e. scratch traffic on every Metropolis step; the short recurrence loop keeps two live values instead)
__constant__ double FF_REC_A[7] = {1.4142135623730951, 1.0, 0.81649658092772603, 0.70710678118654752, 0.63245553203367588,
                                   0.57735026918962576, 0.53452248382484879};  // sqrt(2/(m+1)), m = 0..6
__constant__ double FF_REC_B[7] = {0.0, 0.70710678118654752, 0.81649658092772603, 0.86602540378443865, 0.89442719099991588,
                                   0.91287092917527686, 0.92582009977255146};  // sqrt(m/(m+1))
// degree n <= md, md WAVE-UNIFORM (scalar trip count: no exec-masked loop, two live values, result by select)
FF_D double ff_herm_rec(int n, double x, int md) {
  double hm = 1.0, h = FF_REC_A[0] * x;
  double res = (n == 0) ? 1.0 : h;
  for (int m = 1; m < md; m++) {
    const double hn = fma(FF_REC_A[m] * x, h, -FF_REC_B[m] * hm);
    hm = h;
    h = hn;
    res = (n == m + 1) ? h : res;
  }
  return res;
}

// h_n, h_n', h_n'' of the normalised Hermite polynomials of degrees n[0..NS) at x from ONE pass of the recurrence, with
//   h_n' = sqrt(2 n) h_{n-1},   h_n'' = 2 sqrt(n (n - 1)) h_{n-2}
// -- no coefficient-table loads (ff_herm's Horner fetches 8 coefficients per polynomial from constant memory by a per-lane index),
// no private arrays.  md: wave-uniform bound of the degrees (scalar trip count).
__constant__ double FF_HD1[8] = {0.0, 1.4142135623730951, 2.0, 2.4494897427831779, 2.8284271247461903, 3.1622776601683795,
                                 3.4641016151377544, 3.7416573867739413};                         // sqrt(2 n)
__constant__ double FF_HD2[8] = {0.0, 0.0, 2.8284271247461903, 4.8989794855663558, 6.9282032302755088, 8.9442719099991592,
                                 10.954451150103322, 12.961481396815721};                          // 2 sqrt(n (n - 1))
template <int NS>
FF_D void ff_herm_rec_d2(const int* n, double x, int md, double* h, double* h1, double* h2) {
  double hm2 = 0.0, hm1 = 1.0, hc = FF_REC_A[0] * x;      // h_{m-1}, h_m, h_{m+1} at m = 0
#pragma unroll
  for (int j = 0; j < NS; j++) {
    h[j] = (n[j] == 0) ? 1.0 : hc;
    h1[j] = (n[j] == 0) ? 0.0 : FF_HD1[1];
    h2[j] = 0.0;
  }
  for (int m = 1; m < md; m++) {
    const double hn = fma(FF_REC_A[m] * x, hc, -FF_REC_B[m] * hm1);
    hm2 = hm1; hm1 = hc; hc = hn;
    const double d1 = FF_HD1[m + 1] * hm1, d2 = FF_HD2[m + 1] * hm2;
#pragma unroll
    for (int j = 0; j < NS; j++) {
      const bool sel = n[j] == m + 1;
      h[j] = sel ? hc : h[j]; h1[j] = sel ? d1 : h1[j]; h2[j] = sel ? d2 : h2[j];
    }
  }
}

// h_n(x), h_n'(x), h_n''(x) for ONE degree n <= 7, the recurrence fully unrolled over literal coefficients: no constant-memory
// loads and no loop (a lane of the fused finish of ff_wide.hip evaluates one orbital; every table load there is an exposed
// round trip).  Same numbers as ff_herm_rec_d2 (the same recurrence in the same order).
FF_D void ff_herm_one_d2(int n, double x, double& h, double& h1, double& h2) {
  constexpr double RA[7] = {1.4142135623730951, 1.0, 0.81649658092772603, 0.70710678118654752, 0.63245553203367588,
                            0.57735026918962576, 0.53452248382484879};
  constexpr double RB[7] = {0.0, 0.70710678118654752, 0.81649658092772603, 0.86602540378443865, 0.89442719099991588,
                            0.91287092917527686, 0.92582009977255146};
  constexpr double HD1[8] = {0.0, 1.4142135623730951, 2.0, 2.4494897427831779, 2.8284271247461903, 3.1622776601683795,
                             3.4641016151377544, 3.7416573867739413};
  constexpr double HD2[8] = {0.0, 0.0, 2.8284271247461903, 4.8989794855663558, 6.9282032302755088, 8.9442719099991592,
                             10.954451150103322, 12.961481396815721};
  double hm2 = 0.0, hm1 = 1.0, hc = RA[0] * x;
  h = (n == 0) ? 1.0 : hc;
  h1 = (n == 0) ? 0.0 : HD1[1];
  h2 = 0.0;
#pragma unroll
  for (int m = 1; m < 7; m++) {
    const double hn = fma(RA[m] * x, hc, -RB[m] * hm1);
    hm2 = hm1; hm1 = hc; hc = hn;
    const bool sel = n == m + 1;
    h = sel ? hc : h; h1 = sel ? HD1[m + 1] * hm1 : h1; h2 = sel ? HD2[m + 1] * hm2 : h2;
  }
}

// inverse (by the adjugate) and determinant of a register-resident matrix up to 3 x 3: Ainv[j][b], no pivot search, one reciprocal
template <int NS>
FF_D double ff_inv_small(const double (&A)[NS][NS], double (&Ai)[NS][NS]) {
  static_assert(NS >= 1 && NS <= 3, "closed-form inverse up to 3 x 3");
  if constexpr (NS == 1) {
    Ai[0][0] = 1.0 / A[0][0];
    return A[0][0];
  } else if constexpr (NS == 2) {
    const double det = fma(A[0][0], A[1][1], -A[0][1] * A[1][0]), id = 1.0 / det;
    Ai[0][0] = A[1][1] * id; Ai[0][1] = -A[0][1] * id; Ai[1][0] = -A[1][0] * id; Ai[1][1] = A[0][0] * id;
    return det;
  } else {
    const double c00 = fma(A[1][1], A[2][2], -A[1][2] * A[2][1]), c01 = fma(A[1][2], A[2][0], -A[1][0] * A[2][2]),
                 c02 = fma(A[1][0], A[2][1], -A[1][1] * A[2][0]);
    const double det = fma(A[0][0], c00, fma(A[0][1], c01, A[0][2] * c02)), id = 1.0 / det;
    Ai[0][0] = c00 * id; Ai[1][0] = c01 * id; Ai[2][0] = c02 * id;
    Ai[0][1] = fma(A[0][2], A[2][1], -A[0][1] * A[2][2]) * id;
    Ai[1][1] = fma(A[0][0], A[2][2], -A[0][2] * A[2][0]) * id;
    Ai[2][1] = fma(A[0][1], A[2][0], -A[0][0] * A[2][1]) * id;
    Ai[0][2] = fma(A[0][1], A[1][2], -A[0][2] * A[1][1]) * id;
    Ai[1][2] = fma(A[0][2], A[1][0], -A[0][0] * A[1][2]) * id;
    Ai[2][2] = fma(A[0][0], A[1][1], -A[0][1] * A[1][0]) * id;
    return det;
  }
}

// nx/ny: the orbitals' Hermite degrees, decoded once by the caller (ff_orb_decode) outside its step loop;
// md: a wave-uniform upper bound of those degrees.
// |det D| of a register-resident NS x NS matrix (destroyed): LU with partial pivoting, the product of the pivots
template <int NS>
FF_D double ff_lu_absdet_reg(double (&D)[NS][NS]) {
  double prod = 1.0;
#pragma unroll
  for (int c = 0; c < NS; c++) {
    int p = c;
    double best = fabs(D[c][c]);
#pragma unroll
    for (int r = c + 1; r < NS; r++) {
      double a = fabs(D[r][c]);
      if (a > best) { best = a; p = r; }
    }
#pragma unroll
    for (int r = c + 1; r < NS; r++) {
      bool sw = (p == r);
#pragma unroll
      for (int j = c; j < NS; j++) {
        double a = D[c][j], b = D[r][j];
        D[c][j] = sw ? b : a;
        D[r][j] = sw ? a : b;
      }
    }
    double piv = D[c][c];
    prod *= fabs(piv);
    double ip = 1.0 / piv;
#pragma unroll
    for (int r = c + 1; r < NS; r++) {
      double f = D[r][c] * ip;
#pragma unroll
      for (int j = c + 1; j < NS; j++) D[r][j] = fma(-f, D[c][j], D[r][j]);
    }
  }
  return prod;
}
// log|det D|: one log per determinant (NS <= 6: no over/underflow of the product)
template <int NS>
FF_D double ff_lu_logabsdet_reg(double (&D)[NS][NS]) { return ff_log(ff_lu_absdet_reg<NS>(D)); }

// det D up to its sign, for the ratio test of the Philox-fed Metropolis kernels: closed form up to 3 x 3, pivoted LU beyond
template <int NS>
FF_D double ff_det_reg(double (&D)[NS][NS]) {
  if constexpr (NS == 1) return D[0][0];
  else if constexpr (NS == 2) return fma(D[0][0], D[1][1], -D[0][1] * D[1][0]);
  else if constexpr (NS == 3) {
    const double m0 = fma(D[1][1], D[2][2], -D[1][2] * D[2][1]);
    const double m1 = fma(D[1][0], D[2][2], -D[1][2] * D[2][0]);
    const double m2 = fma(D[1][0], D[2][1], -D[1][1] * D[2][0]);
    return fma(D[0][0], m0, fma(-D[0][1], m1, D[0][2] * m2));
  } else return ff_lu_absdet_reg<NS>(D);
}

// the polynomial part of one row of the Slater matrix: h_nx_j(x) h_ny_j(y) -- phi_j(x, y) without pi^-1/2 exp(-r^2/2), which is
// the same for every orbital of the row: det[phi_j(r_i)] = prod_i (pi^-1/2 exp(-r_i^2/2)) det[h_nx_j(x_i) h_ny_j(y_i)]
// (one pass of the recurrence per coordinate serves the NS orbitals: NS selects per degree instead of NS recurrences)
template <int NS>
FF_D void ff_herm_rec_n(const int* n, double x, int md, double* res) {
  double hm = 1.0, h = FF_REC_A[0] * x;
#pragma unroll
  for (int j = 0; j < NS; j++) res[j] = (n[j] == 0) ? 1.0 : h;
  for (int m = 1; m < md; m++) {
    const double hn = fma(FF_REC_A[m] * x, h, -FF_REC_B[m] * hm);
    hm = h;
    h = hn;
#pragma unroll
    for (int j = 0; j < NS; j++) res[j] = (n[j] == m + 1) ? h : res[j];
  }
}
template <int NS>
FF_D void ff_poly_row_reg(const int* nx, const int* ny, double x, double y, int md, double* row) {
  double hy[NS];
  ff_herm_rec_n<NS>(nx, x, md, row);
  ff_herm_rec_n<NS>(ny, y, md, hy);
#pragma unroll
  for (int j = 0; j < NS; j++) row[j] *= hy[j];
}
template <int NS>
FF_D double ff_slater_polydet_reg(const int* nx, const int* ny, const double* x, int md) {
  double D[NS][NS];
#pragma unroll
  for (int i = 0; i < NS; i++) ff_poly_row_reg<NS>(nx, ny, x[2 * i], x[2 * i + 1], md, D[i]);
  return ff_det_reg<NS>(D);
}

// one row of the Slater matrix: phi_j(x, y), j = 0..NS-1
template <int NS>
FF_D void ff_slater_row_reg(const int* nx, const int* ny, double x, double y, int md, double* row) {
  double gs = ff_gauss2d_fast(x, y);
#pragma unroll
  for (int j = 0; j < NS; j++) row[j] = gs * ff_herm_rec(nx[j], x, md) * ff_herm_rec(ny[j], y, md);
}

template <int NS>
FF_D double ff_slater_logabsdet_reg(const int* nx, const int* ny, const double* x, int md) {
  double D[NS][NS];
#pragma unroll
  for (int i = 0; i < NS; i++) ff_slater_row_reg<NS>(nx, ny, x[2 * i], x[2 * i + 1], md, D[i]);
  return ff_lu_logabsdet_reg<NS>(D);
}

// --------------------------------------------------------------------------------------------------
// General routine (runtime ns <= FF_MAX_NS, arrays in private memory).  Builds D, inverts it by
// Gauss-Jordan with partial pivoting and returns log|det D|.  If T/S are given also forms
//   T[c][a][b] = sum_j d_c phi_j(r_a) Dinv[j][b]         (c = 0,1)
//   S[a][0..2] = sum_j (d_xx, d_xy, d_yy) phi_j(r_a) Dinv[j][a]
// from which  grad_a,c log|det| = T[c][a][a],  Hessian[(a,c),(b,e)] = delta_ab S_a^{ce} - T[c][a][b] T[e][b][a].
FF_D double ff_slater_general(int ns, const int* __restrict__ orb, const double* x,
                              double* T /* [2*ns*ns] or null */, double* S /* [3*ns] or null */) {
  double A[FF_MAX_NS * FF_MAX_NS], Inv[FF_MAX_NS * FF_MAX_NS];
  for (int i = 0; i < ns; i++) {
    double gs = ff_gauss2d(x[2 * i], x[2 * i + 1]);
    for (int j = 0; j < ns; j++) {
      double v;
      ff_orbital<false>(orb[j], x[2 * i], x[2 * i + 1], gs, v, nullptr, nullptr);
      A[i * ns + j] = v;
      Inv[i * ns + j] = (i == j) ? 1.0 : 0.0;
    }
  }
  double acc = 0.0;
  for (int c = 0; c < ns; c++) {
    int p = c;
    double best = fabs(A[c * ns + c]);
    for (int r = c + 1; r < ns; r++) {
      double a = fabs(A[r * ns + c]);
      if (a > best) { best = a; p = r; }
    }
    if (p != c)
      for (int j = 0; j < ns; j++) {
        double t = A[c * ns + j]; A[c * ns + j] = A[p * ns + j]; A[p * ns + j] = t;
        t = Inv[c * ns + j]; Inv[c * ns + j] = Inv[p * ns + j]; Inv[p * ns + j] = t;
      }
    double piv = A[c * ns + c];
    acc += log(fabs(piv));
    if (!T) {  // value only: plain elimination below the pivot
      double ip = 1.0 / piv;
      for (int r = c + 1; r < ns; r++) {
        double f = A[r * ns + c] * ip;
        for (int j = c + 1; j < ns; j++) A[r * ns + j] = fma(-f, A[c * ns + j], A[r * ns + j]);
      }
      continue;
    }
    double ip = 1.0 / piv;
    for (int j = 0; j < ns; j++) { A[c * ns + j] *= ip; Inv[c * ns + j] *= ip; }
    for (int r = 0; r < ns; r++) {
      if (r == c) continue;
      double f = A[r * ns + c];
      for (int j = 0; j < ns; j++) {
        A[r * ns + j] = fma(-f, A[c * ns + j], A[r * ns + j]);
        Inv[r * ns + j] = fma(-f, Inv[c * ns + j], Inv[r * ns + j]);
      }
    }
  }
  if (T) {
    for (int a = 0; a < ns; a++) {
      double gs = ff_gauss2d(x[2 * a], x[2 * a + 1]);
      for (int b = 0; b < ns; b++) { T[a * ns + b] = 0.0; T[ns * ns + a * ns + b] = 0.0; }
      double s0 = 0.0, s1 = 0.0, s2 = 0.0;
      for (int j = 0; j < ns; j++) {
        double v, g[2], hs[3];
        ff_orbital<true>(orb[j], x[2 * a], x[2 * a + 1], gs, v, g, hs);
        for (int b = 0; b < ns; b++) {
          double di = Inv[j * ns + b];
          T[a * ns + b] = fma(g[0], di, T[a * ns + b]);
          T[ns * ns + a * ns + b] = fma(g[1], di, T[ns * ns + a * ns + b]);
        }
        double da = Inv[j * ns + a];
        s0 = fma(hs[0], da, s0); s1 = fma(hs[1], da, s1); s2 = fma(hs[2], da, s2);
      }
      if (S) { S[3 * a] = s0; S[3 * a + 1] = s1; S[3 * a + 2] = s2; }
    }
  }
  return acc;
}

// The same routine for a compile-time determinant size: every loop unrolls, the row exchange of the pivoting is a chain
// of selects, and A, Inv, T live in registers instead of private (scratch) memory.  Same operations in the same order as
// ff_slater_general with T and S requested: bit-identical results.
template <int NS, bool DERIV = true>
FF_D double ff_slater_fixed(const int* __restrict__ orb, const double* x, double* T /* [2*NS*NS] */, double* S /* [3*NS] */) {
  double A[NS * NS], Inv[DERIV ? NS * NS : 1];
#pragma unroll
  for (int i = 0; i < NS; i++) {
    const double gs = ff_gauss2d(x[2 * i], x[2 * i + 1]);
#pragma unroll
    for (int j = 0; j < NS; j++) {
      double v;
      ff_orbital<false>(orb[j], x[2 * i], x[2 * i + 1], gs, v, nullptr, nullptr);
      A[i * NS + j] = v;
      if constexpr (DERIV) Inv[i * NS + j] = (i == j) ? 1.0 : 0.0;
    }
  }
  double acc = 0.0;
#pragma unroll
  for (int c = 0; c < NS; c++) {
    int p = c;
    double best = fabs(A[c * NS + c]);
#pragma unroll
    for (int r = c + 1; r < NS; r++) {
      const double a = fabs(A[r * NS + c]);
      if (a > best) { best = a; p = r; }
    }
#pragma unroll
    for (int r = c + 1; r < NS; r++) {
      const bool sw = (r == p);
#pragma unroll
      for (int j = 0; j < NS; j++) {
        const double ac = A[c * NS + j], ar = A[r * NS + j];
        A[c * NS + j] = sw ? ar : ac; A[r * NS + j] = sw ? ac : ar;
        if constexpr (DERIV) {
          const double ic = Inv[c * NS + j], ir = Inv[r * NS + j];
          Inv[c * NS + j] = sw ? ir : ic; Inv[r * NS + j] = sw ? ic : ir;
        }
      }
    }
    const double piv = A[c * NS + c];
    acc += log(fabs(piv));
    const double ip = 1.0 / piv;
    if constexpr (!DERIV) {   // value only: plain elimination below the pivot (as ff_slater_general without T)
#pragma unroll
      for (int r = c + 1; r < NS; r++) {
        const double f = A[r * NS + c] * ip;
#pragma unroll
        for (int j = c + 1; j < NS; j++) A[r * NS + j] = fma(-f, A[c * NS + j], A[r * NS + j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NS; j++) { A[c * NS + j] *= ip; Inv[c * NS + j] *= ip; }
#pragma unroll
      for (int r = 0; r < NS; r++) {
        if (r == c) continue;
        const double f = A[r * NS + c];
#pragma unroll
        for (int j = 0; j < NS; j++) {
          A[r * NS + j] = fma(-f, A[c * NS + j], A[r * NS + j]);
          Inv[r * NS + j] = fma(-f, Inv[c * NS + j], Inv[r * NS + j]);
        }
      }
    }
  }
  if constexpr (DERIV) {
#pragma unroll
    for (int a = 0; a < NS; a++) {
      const double gs = ff_gauss2d(x[2 * a], x[2 * a + 1]);
#pragma unroll
      for (int b = 0; b < NS; b++) { T[a * NS + b] = 0.0; T[NS * NS + a * NS + b] = 0.0; }
      double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int j = 0; j < NS; j++) {
        double v, g[2], hs[3];
        ff_orbital<true>(orb[j], x[2 * a], x[2 * a + 1], gs, v, g, hs);
#pragma unroll
        for (int b = 0; b < NS; b++) {
          const double di = Inv[j * NS + b];
          T[a * NS + b] = fma(g[0], di, T[a * NS + b]);
          T[NS * NS + a * NS + b] = fma(g[1], di, T[NS * NS + a * NS + b]);
        }
        const double da = Inv[j * NS + a];
        s0 = fma(hs[0], da, s0); s1 = fma(hs[1], da, s1); s2 = fma(hs[2], da, s2);
      }
      if (S) { S[3 * a] = s0; S[3 * a + 1] = s1; S[3 * a + 2] = s2; }
    }
  }
  return acc;
}
