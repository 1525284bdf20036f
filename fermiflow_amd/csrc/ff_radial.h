// ff_radial.h -- tabulated radial functions eta(r), mu(r) and their derivatives for the fused ODE kernels.
//
// Inside one launch the MLP weights are fixed, so eta and mu (src/MLP.py:30-45) are fixed analytic functions of one
// variable -- exactly the situation in which QMC codes tabulate their radial functions.  A small kernel evaluates
// eta^(0..8) and mu^(0..8) on a uniform grid (spacing h = 2^-k chosen from max|w1| so that the truncation error of the
// expansion below stays ~1e-16 relative to the function's own scale); the ODE kernels then evaluate the NH derivative
// heads of a radius as 5th-order Taylor expansions about the nearest node:
//     f^(m)(r) = sum_{k=0..5} T[j][m+k] dr^k / k!,   dr = r - j h,  |dr| <= h/2,  remainder <= (h/2)^6/720 |f^(m+6)|.
// Cost per radius: ~30 fp64 instructions instead of H (=50) exp/rcp chains (~2000).  Radii beyond the table
// (r >= FF_TAB_RMAX), non-finite radii, or weights too stiff for the largest grid fall back to the direct evaluation.
// The table is rebuilt by every call that receives new weights (a few microseconds).
#pragma once
#include "ff_common.h"

#define FF_TAB_RMAX 32.0
#ifndef FF_TAB_ROW
#define FF_TAB_ROW 10                       // doubles per node (9 used; 80-byte rows keep 16-byte alignment)
#endif
#define FF_TAB_MAXLOG 9                     // finest grid: h = 2^-9
#define FF_TAB_NMAX (32 * (1 << FF_TAB_MAXLOG) + 1)
#define FF_TAB_HDR 16                       // [0] 1/h, [1] h, [2] nodes, [3] 1.0 if the table must not be used,
                                            // [4] 1.0 if the adjoint's deposit grid must not be used,
                                            // [5] coefficients per deposit row these weights need (6, 8, 10 or 12),
                                            // [8..15] event slots written by the kernels that read the table: a launch
                                            // that met a radius beyond the table leaves its id in slot (id mod 8)
#define FF_TAB_EVT0 8
#define FF_TAB_NEVT 8
#define FF_TAB_DOUBLES (FF_TAB_HDR + 2 * FF_TAB_NMAX * FF_TAB_ROW)

// sigma^(n)(a) as a polynomial in s = sigma(a): P_0 = s, P_{n+1} = P_n'(s) s (1 - s)
// (a macro, expanded into a function-local constexpr array: with the fully unrolled loops below every coefficient becomes an instruction
// literal.  As a __constant__ array it cost the small kernels a chain of COLD scalar loads -- ff_dep_contract_kernel spent 35 of its
// 40 us waiting for eighteen of them, one round trip to HBM each)
#define FF_SIGPOLY_INIT { \
    {0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, \
    {0, 1, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, \
    {0, 1, -3, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, \
    {0, 1, -7, 12, -6, 0, 0, 0, 0, 0, 0, 0, 0, 0}, \
    {0, 1, -15, 50, -60, 24, 0, 0, 0, 0, 0, 0, 0, 0}, \
    {0, 1, -31, 180, -390, 360, -120, 0, 0, 0, 0, 0, 0, 0}, \
    {0, 1, -63, 602, -2100, 3360, -2520, 720, 0, 0, 0, 0, 0, 0}, \
    {0, 1, -127, 1932, -10206, 25200, -31920, 20160, -5040, 0, 0, 0, 0, 0}, \
    {0, 1, -255, 6050, -46620, 166824, -317520, 332640, -181440, 40320, 0, 0, 0, 0}, \
    {0, 1, -511, 18660, -204630, 1020600, -2739240, 4233600, -3780000, 1814400, -362880, 0, 0, 0}, \
    {0, 1, -1023, 57002, -874500, 5921520, -21538440, 46070640, -59875200, 46569600, -19958400, 3628800, 0, 0}, \
    {0, 1, -2047, 173052, -3669006, 33105600, -158838240, 451725120, -801496080, 898128000, -618710400, 239500800, -39916800, 0}, \
    {0, 1, -4095, 523250, -15195180, 180204024, -1118557440, 4115105280, -9574044480, 14495120640, -14270256000, 8821612800, -3113510400, 479001600}}


// sigma^(0..NMAXD)(a) from s = sigma(a)
template <int NMAXD>
FF_D void ff_sigma_derivs(double s, double* out) {
  constexpr double FF_SIGPOLY[13][14] = FF_SIGPOLY_INIT;
#pragma unroll
  for (int n = 0; n <= NMAXD; n++) {
    double p = FF_SIGPOLY[n][n + 1];
#pragma unroll
    for (int c = n; c >= 1; c--) p = fma(p, s, FF_SIGPOLY[n][c]);
    out[n] = p * s;
  }
}

// ---- "deposit" grid of the tabulated adjoint (ff_cnf_adj.hip): the parameter gradient of
//   sum_records w [ca f(r) + cb f'(r)]  with  f(r) = sum_k T[j][k] dr^k/k!  is  sum_{j,k} Wacc[j][k] dT[j][k]/dtheta,
// so the kernel only accumulates Wacc (coefficients ca dr^k/k! + cb dr^(k-1)/(k-1)!) on a coarse grid, h_d = 1/16,
// expansion order 11, and one small kernel contracts Wacc with dT/dtheta at the end.  Usable while max|w1| h_d <= 0.4
// (the kernels expand up to 1.5 h_d from a node: ff_deposit5).
#define FF_DEP_INVH 16.0
#ifndef FF_DEP_NLDS
#define FF_DEP_NLDS 128                     // nodes kept in LDS (r < 8); the rest (r < 32) goes to a global table
#endif
#define FF_DEP_NTOT 512
#define FF_DEP_ROW 12                       // T_0 .. T_11

#ifdef FF_RADIAL_BUILD_KERNELS   // defined by the one translation unit that owns ff_radial_table_build
// The whole table in ONE launch (round 3: a single-workgroup header kernel + a one-lane-per-node build kernel, 5 + 18 us in front of
// every sweep).  Every workgroup derives the header -- grid spacing from the stiffest first-layer weight -- itself (the same
// hundred loads and the same arithmetic everywhere), workgroup 0 stores it; FOUR lanes share a node, each summing every fourth
// hidden unit, joined by a quad reduction in a fixed order: the sequential chain of H sigmoid evaluations per lane was the kernel.
#define FF_TAB_NODES_PER_WG 32
#ifndef FF_TAB_GRID
#define FF_TAB_GRID 256                     // one workgroup per CU: 2 x 2049 nodes (h = 1/64) in one round, 2 x 16385 in four
#endif
__global__ void __launch_bounds__(128) ff_table_kernel(ff_net net, double* __restrict__ tab) {
  FF_SETPRIO();
  __shared__ double sm[128];
  const int tid = threadIdx.x;
  double w = 0.0;
  for (int h = tid; h < net.He; h += 128) w = fmax(w, fabs(net.ew1[h]));
  for (int h = tid; h < net.Hm; h += 128) w = fmax(w, fabs(net.mw1[h]));
  if (!(w == w)) w = __builtin_inf();      // fmax drops NaN: a NaN weight must make the table unusable
  for (int h = tid; h < net.He; h += 128) if (!(net.ew1[h] == net.ew1[h])) w = __builtin_inf();
  for (int h = tid; h < net.Hm; h += 128) if (!(net.mw1[h] == net.mw1[h])) w = __builtin_inf();
  sm[tid] = w;
  __syncthreads();
  for (int q = 64; q > 0; q >>= 1) {
    if (tid < q) sm[tid] = fmax(sm[tid], sm[tid + q]);
    __syncthreads();
  }
  w = sm[0];
  // (w h/2)^6/720 <~ 1e-15  <=>  w h <= 0.06
#ifndef FF_TAB_MINLOG
#define FF_TAB_MINLOG 6
#endif
  int lg = FF_TAB_MINLOG;
  while (lg < FF_TAB_MAXLOG && w * ldexp(1.0, -lg) > 0.06) lg++;
  const bool bad = !(w * ldexp(1.0, -lg) <= 0.06);
  const double hstep = ldexp(1.0, -lg);
  const int nodes = 32 * (1 << lg) + 1;
  if (blockIdx.x == 0 && tid == 0) {
    tab[0] = ldexp(1.0, lg);
    tab[1] = hstep;
    tab[2] = (double)nodes;
    tab[3] = bad ? 1.0 : 0.0;
    tab[4] = (w * (1.0 / FF_DEP_INVH) <= 0.4) ? 0.0 : 1.0;   // 1.0: the coarse deposit grid is not accurate enough
    // [5]: coefficients of a deposit row these weights need.  A deposit is expanded up to 1.5 h_d from its node; the remainder of n
    // coefficients, (1.5 w h_d)^n / n!, is held to what the full row leaves at the validity bound above: 0.6^12 / 12! = 4.5e-12.
    {
      const double xd = 1.5 * w * (1.0 / FF_DEP_INVH);
      double nrow = (double)FF_DEP_ROW, rem = xd * xd * xd * xd * (1.0 / 24.0);      // xd^4 / 4!
      for (int n = 6; n < FF_DEP_ROW; n += 2) {
        rem *= xd * xd / (double)(n * (n - 1));
        if (rem <= 4.5e-12) { nrow = (double)n; break; }
      }
      tab[5] = nrow;
    }
    for (int q = 6; q < FF_TAB_HDR; q++) tab[q] = 0.0;
  }
  if (bad) return;
  // f^(0..8)(r_j) = sum_h w2 w1^n sigma^(n)(w1 r_j + b1): quad q of the launch takes the nodes q, q + (quads of the launch), ... of
  // the 2 x nodes (net, node) pairs; its lanes the hidden units h = tid % 4, + 4, ...
  const int part = tid & 3;
  const int nquads = (int)gridDim.x * FF_TAB_NODES_PER_WG;
  for (int idx = blockIdx.x * FF_TAB_NODES_PER_WG + (tid >> 2); idx - (tid >> 2) < 2 * nodes; idx += nquads) {     // (workgroup-uniform trip count)
    const int t = idx >= nodes ? 1 : 0, j = idx - t * nodes;
    const bool live = idx < 2 * nodes;         // (whole quads are live or not: the quad reduction below is uniform per quad)
    const int H = live ? (t ? net.Hm : net.He) : 0;
    const double* w1 = t ? net.mw1 : net.ew1;
    const double* b1 = t ? net.mb1 : net.eb1;
    const double* w2 = t ? net.mw2 : net.ew2;
    const double r = (double)j * hstep;
    double acc[9];
#pragma unroll
    for (int n = 0; n < 9; n++) acc[n] = 0.0;
    for (int h = part; h < H; h += 4) {
      const double s = ff_sigmoid(fma(w1[h], r, b1[h]));
      double sd[9], wp = w2[h];
      ff_sigma_derivs<8>(s, sd);
#pragma unroll
      for (int n = 0; n < 9; n++) {
        acc[n] = fma(wp, sd[n], acc[n]);
        wp *= w1[h];
      }
    }
#pragma unroll
    for (int n = 0; n < 9; n++) {
      acc[n] += ff_swap1(acc[n]);
      acc[n] += ff_swap2(acc[n]);
    }
    if (live && part == 0) {
      double* row = tab + FF_TAB_HDR + ((size_t)t * FF_TAB_NMAX + j) * FF_TAB_ROW;
#pragma unroll
      for (int n = 0; n < 9 && n < FF_TAB_ROW; n++) row[n] = acc[n];
#pragma unroll
      for (int n = 9; n < FF_TAB_ROW; n++) row[n] = 0.0;
    }
  }
}

#endif  // FF_RADIAL_BUILD_KERNELS

// NH derivative heads of net t (0 eta, 1 mu) at radius r from the table, in two halves so that a caller can put other
// work (or a second fetch) between the loads and their use.  fetch returns false if r is off the table.
template <int NH>
FF_D bool ff_table_fetch(const double* __restrict__ tab, double inv_h, double h, int t, double r, double* T, double& dr) {
  if (!(r < FF_TAB_RMAX)) return false;
  const double jf = rint(r * inv_h);
  dr = fma(-jf, h, r);
#ifdef FF_DIAG_TAB_ROW0      // (timing diagnostic: every lane reads ONE row -- what the table's cache misses cost; the numbers are then wrong)
  const double* __restrict__ row = tab + FF_TAB_HDR + ((size_t)t * FF_TAB_NMAX + 100) * FF_TAB_ROW;
#else
  const double* __restrict__ row = tab + FF_TAB_HDR + ((size_t)t * FF_TAB_NMAX + (int)jf) * FF_TAB_ROW;
#endif
  constexpr int NT = NH + 5 < FF_TAB_ROW ? NH + 5 : FF_TAB_ROW;      // (a row holds FF_TAB_ROW derivatives: the highest heads expand to fewer orders)
#pragma unroll
  for (int e = 0; e < NT; e++) T[e] = row[e];
  return true;
}

template <int NH>
FF_D void ff_table_eval(const double* T, double dr, double* hd) {
  constexpr int NT = NH + 5 < FF_TAB_ROW ? NH + 5 : FF_TAB_ROW;
  const double dk[5] = {dr, dr * 0.5, dr * (1.0 / 3.0), dr * 0.25, dr * 0.2};
#pragma unroll
  for (int m = 0; m < NH; m++) {
    const int top = m + 5 < NT ? m + 5 : NT - 1;      // (compile-time after unrolling)
    double v = T[top];
#pragma unroll
    for (int k = 4; k >= 0; k--)
      if (m + k < top) v = fma(v, dk[k], T[m + k]);
    hd[m] = v;
  }
}

template <int NH>
FF_D bool ff_heads_table(const double* __restrict__ tab, double inv_h, double h, int t, double r, double* hd) {
  double T[NH + 5], dr;
  if (!ff_table_fetch<NH>(tab, inv_h, h, t, r, T, dr)) return false;
  ff_table_eval<NH>(T, dr, hd);
  return true;
}
