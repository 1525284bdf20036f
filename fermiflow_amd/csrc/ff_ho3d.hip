// ff_ho3d.hip -- groundwork for BASELINE.json configs[4] (3-D harmonic trap, fp32 MLP path), SURVEY.md 8(f).4.
//
// The reference has no three-dimensional code (src/orbitals.py:56 and src/base_dist.py:62 hard-code d = 2), so there is no
// upstream file to cite beyond the two-dimensional ones this generalises:
//   HO3D orbitals        <- HO2D, src/orbitals.py:65-82          (one more Hermite factor, E = shell + 3/2)
//   ff_logprob3d         <- FreeFermion.log_prob, src/base_dist.py:49-56, with the gradient / Laplacian y_grad_laplacian extracts
//   ff_mcmc_sample*3d    <- FreeFermion.sample, src/base_dist.py:58-71 (walkers (B, n, 3))
//   ff_backflow_v_div_f32<- Backflow.forward / .divergence, src/equivariant_funs.py:83-102, sigmoid sums in fp32
// One lane per walker and runtime determinant sizes (private arrays): correct and simple, not yet tuned -- the fused ODE
// kernels are templated on d but only instantiated for d = 2.
// Known-answer test: E_loc == sum of orbital energies at every point (tests/test_basedist.py:5-60 one dimension up).
#include "ff_common.h"
#include "ff_slater.h"
#include "ff_rng.h"
#include "ff_eloc_ws.h"

#include "ff_slater_rows.h"

// log|det D| (D_ij = phi_j(r_i)) of ns particles; with grad: d/dr_i = sum_j grad phi_j(r_i) Dinv_ji (src/slater.py:51-61),
// and the Laplacian sum_i [ sum_j lap phi_j(r_i) Dinv_ji - |grad_i|^2 ] (rank-one update of the determinant).
// Value only: LU with partial pivoting, the operations of ff_slater_general; with derivatives: Gauss-Jordan inverse.
FF_D double ff_ho3d_logabsdet(int ns, const int* __restrict__ orb, const double* x, double* grad, double* lap) {
  double A[FF_MAX_NS * FF_MAX_NS], Inv[FF_MAX_NS * FF_MAX_NS];
  const bool deriv = grad != nullptr;
  for (int i = 0; i < ns; i++) {
    const double gs = ff_gauss3d(x + 3 * i);
    for (int j = 0; j < ns; j++) {
      double v, lp;
      ff_orbital3d<false>(orb[j], x + 3 * i, gs, v, nullptr, lp);
      A[i * ns + j] = v;
      if (deriv) Inv[i * ns + j] = (i == j) ? 1.0 : 0.0;
    }
  }
  double acc = 0.0;
  for (int c = 0; c < ns; c++) {
    int p = c;
    double best = fabs(A[c * ns + c]);
    for (int r = c + 1; r < ns; r++) {
      const double a = fabs(A[r * ns + c]);
      if (a > best) { best = a; p = r; }
    }
    if (p != c)
      for (int j = 0; j < ns; j++) {
        double t = A[c * ns + j]; A[c * ns + j] = A[p * ns + j]; A[p * ns + j] = t;
        if (deriv) { t = Inv[c * ns + j]; Inv[c * ns + j] = Inv[p * ns + j]; Inv[p * ns + j] = t; }
      }
    const double piv = A[c * ns + c];
    acc += log(fabs(piv));
    if (!deriv) {
      for (int r = c + 1; r < ns; r++) {
        const double f = A[r * ns + c] / piv;
        for (int j = c + 1; j < ns; j++) A[r * ns + j] = A[r * ns + j] - f * A[c * ns + j];
      }
    } else {
      const double ip = 1.0 / piv;
      for (int j = 0; j < ns; j++) { A[c * ns + j] *= ip; Inv[c * ns + j] *= ip; }
      for (int r = 0; r < ns; r++) {
        if (r == c) continue;
        const double f = A[r * ns + c];
        for (int j = 0; j < ns; j++) {
          A[r * ns + j] = fma(-f, A[c * ns + j], A[r * ns + j]);
          Inv[r * ns + j] = fma(-f, Inv[c * ns + j], Inv[r * ns + j]);
        }
      }
    }
  }
  if (deriv) {
    double l = 0.0;
    for (int a = 0; a < ns; a++) {
      const double gs = ff_gauss3d(x + 3 * a);
      double g[3] = {0.0, 0.0, 0.0}, s = 0.0;
      for (int j = 0; j < ns; j++) {
        double v, gj[3], lj;
        ff_orbital3d<true>(orb[j], x + 3 * a, gs, v, gj, lj);
        const double da = Inv[j * ns + a];
        g[0] = fma(gj[0], da, g[0]); g[1] = fma(gj[1], da, g[1]); g[2] = fma(gj[2], da, g[2]);
        s = fma(lj, da, s);
      }
      grad[3 * a] = g[0]; grad[3 * a + 1] = g[1]; grad[3 * a + 2] = g[2];
      l += s - (g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    }
    *lap = l;
  }
  return acc;
}

// logp = 2 (log|det up| + log|det dn|) and, optionally, its gradient and Laplacian wrt all coordinates
__global__ void __launch_bounds__(64)
ff_logprob3d_kernel(int64_t B, int nup, int ndn, const int* __restrict__ tab_up, const int* __restrict__ tab_dn,
                    const int* __restrict__ wstate, const double* __restrict__ x, double* __restrict__ logp,
                    double* __restrict__ grad, double* __restrict__ lap) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int n = nup + ndn, st = wstate ? wstate[b] : 0;
  double xl[3 * FF_MAX_NS], gl[3 * FF_MAX_NS];
  double lp = 0.0, lsum = 0.0;
  for (int sp = 0; sp < 2; sp++) {
    const int ns = sp ? ndn : nup, off = sp ? nup : 0;
    if (!ns) continue;
    for (int k = 0; k < 3 * ns; k++) xl[k] = x[b * 3 * n + 3 * off + k];
    double l = 0.0;
    lp += ff_ho3d_logabsdet(ns, (sp ? tab_dn : tab_up) + st * ns, xl, grad ? gl : nullptr, &l);
    if (grad) {
      for (int k = 0; k < 3 * ns; k++) grad[b * 3 * n + 3 * off + k] = 2.0 * gl[k];
      lsum += 2.0 * l;
    }
  }
  logp[b] = 2.0 * lp;
  if (lap) lap[b] = lsum;
}

// two roundings, as torch evaluates x + tau * g (the backend must not contract them into one fma)
FF_D double ff3_mul_rn(double a, double b) { double r = a * b; FF_OPAQUE(r); return r; }
FF_D double ff3_add_rn(double a, double b) { double r = a + b; FF_OPAQUE(r); return r; }

// FreeFermion.sample with SIXTEEN LANES PER DETERMINANT (two walkers x two spin species per wave), for every shape the
// register-resident samplers of ff_walkers.hip do not cover: d = 3 (BASELINE configs[4]: two 10 x 10 determinants per step) and the
// d = 2 systems beyond their template list.  Lane r of a group owns particle r of its species: it draws that particle's
// proposal, evaluates its row of D_ij = phi_j(r_i), and the LU with partial pivoting runs without moving rows -- per column the
// unused lane with the largest entry is the pivot (16-lane butterfly), publishes its row through LDS, the other unused lanes
// eliminate.  Every arithmetic operation is the one the one-lane-per-walker kernels perform (ff_slater_general in d = 2,
// ff_ho3d_logabsdet in d = 3: same pivots, same multipliers, log|det| summed column by column), so chains, accept masks and
// log-densities are bit-identical to theirs and to the oracle's -- only who computes what changed.  (One lane per walker kept two
// 12 x 12 matrices in scratch: 85 ms per 131 072 walkers x 100 steps at 20 particles.)
// NOISE: explicit g0 (B,n,D), g (S,B,n,D), u (S,B); otherwise Philox counters (seed, woff + b) with the quad / uniform slots of
// the respective one-lane kernel.  g0 without NOISE: the walkers to continue from (ff_mcmc_continue).
// NS: compile-time capacity of a species' determinant (8, 10 or 12: the matrix row of a lane and every elimination loop have that many
// entries -- at configs[4], 10 x 10, a sixth less than with FF_MAX_NS)
#ifndef FF_ROWS_WAVES
#define FF_ROWS_WAVES 4      // waves per SIMD the sixteen-lane sampler is compiled for (115 registers; A/B knob)
#endif
template <int D, bool NOISE, int NS>
__global__ void __launch_bounds__(FF_WAVE, FF_ROWS_WAVES)
ff_mcmc_rows_kernel(int64_t B, int nup, int ndn, const int* __restrict__ tab_up, const int* __restrict__ tab_dn,
                    const int* __restrict__ wstate, int steps, double tau, const double* __restrict__ g0,
                    const double* __restrict__ g, const double* __restrict__ u, uint64_t seed, int64_t woff,
                    double* __restrict__ x_out, double* __restrict__ logp_out, uint8_t* __restrict__ accept, int* __restrict__ acc_count) {
  __shared__ double s_row[2][4][NS];            // the pivot row of a column, double-buffered by column parity: ONE barrier per column
  __shared__ int s_deg[4][NS][D];               // Hermite degrees of the group's orbitals
  // per lane: h_0 .. h_{MDL-1} of each coordinate of its particle.  Degrees beyond (orbitals past the fourth shell: 10 per species in
  // d = 2, 20 in d = 3 -- excited many-body states of BetaVMC) are re-run through the recurrence where they are needed (same operations,
  // same bits).  With all eight degrees tabulated this array alone was 12 of the kernel's 14.6 KB of LDS: ten workgroups per CU, 2.5
  // waves per SIMD for a kernel whose every column of the elimination waits for a pivot.
  constexpr int MDL = 4;
  __shared__ double s_h[FF_WAVE][D][MDL];
  __shared__ int s_md;
  const int lane = threadIdx.x, grp = lane >> 4, r = lane & 15;
  const int64_t gid = (int64_t)blockIdx.x * 4 + grp;
  int64_t b = gid >> 1;
  const int sp = (int)(gid & 1);
  const bool live = b < B;
  if (!live) b = B - 1;          // idle groups shadow the last walker (they take part in the barriers and lane exchanges)
  const int n = nup + ndn, M = D * n;
  const int ns = sp ? ndn : nup, off = sp ? nup : 0;
  const int st = wstate ? wstate[b] : 0;
  const bool mine = r < ns;
  if (lane == 0) s_md = 0;
  __syncthreads();
  if (mine) {
    const int k = ((sp ? tab_dn : tab_up) + st * ns)[r];
    int dg[3] = {0, 0, 0};
    if constexpr (D == 2) ff_orb_decode(k, dg[0], dg[1]); else ff_ho3d_decode(k, dg[0], dg[1], dg[2]);
    int mx = 0;
#pragma unroll
    for (int c = 0; c < D; c++) { s_deg[grp][r][c] = dg[c]; mx = dg[c] > mx ? dg[c] : mx; }
    atomicMax(&s_md, mx);
  }
  __syncthreads();
  const int md = FF_UNIFORM(s_md);               // largest Hermite degree in the workgroup (scalar loop bound)
  const uint64_t wid = (uint64_t)(woff + b);
  const int i0 = D * (off + (mine ? r : 0));      // this lane's first coordinate
  const int nsmax = nup > ndn ? nup : ndn;

  // log|det| of the species' matrix at the positions xx (this lane's particle); identical on all lanes of the group.
  // Orbitals: the normalised Hermite functions of every coordinate once per call (three-term recurrence, as the register-resident
  // samplers of ff_walkers.hip), phi_j = gauss * prod_c h_{deg_j,c}(x_c).  LU with partial pivoting without moving rows; the
  // pivot of a column is found by ONE 32-bit maximum over the group's 16 lanes (DPP): key = |entry| rounded to float with the lane
  // index in the low four bits (equal keys: the lower lane) -- a pivot that is within 2^-19 of the largest entry instead of the
  // largest changes the rounding of log|det|, not its value.  |det| is the product of the pivots (one log per determinant).
  // detprod(xx, true): the product of the pivots of the species' Slater matrix, the entries with their Gaussians; (xx, false): of its
  // polynomial part h_nx(x) h_ny(y) [h_nz(z)] alone (the Gaussian of a row factors out of the determinant: the Philox-fed chain)
  auto detprod = [&](const double* xx, bool with_gauss) -> double {
    double A[NS];
    double gs = 1.0;
    if (with_gauss) { if constexpr (D == 2) gs = ff_gauss2d(xx[0], xx[1]); else gs = ff_gauss3d(xx); }
#pragma unroll
    for (int c = 0; c < D; c++) {
      double hm = 1.0, h = FF_REC_A[0] * xx[c];
      s_h[lane][c][0] = 1.0;
      s_h[lane][c][1] = h;
      const int mtab = md < MDL - 1 ? md : MDL - 1;
      for (int m = 1; m < mtab; m++) {
        const double hn = fma(FF_REC_A[m] * xx[c], h, -FF_REC_B[m] * hm);
        hm = h;
        h = hn;
        s_h[lane][c][m + 1] = h;
      }
    }
    auto hdeg = [&](int c, int dg) -> double {      // (dg is uniform within the group)
      if (dg < MDL) return s_h[lane][c][dg];
      double hm = 1.0, h = FF_REC_A[0] * xx[c];
      for (int m = 1; m < dg; m++) {
        const double hn = fma(FF_REC_A[m] * xx[c], h, -FF_REC_B[m] * hm);
        hm = h;
        h = hn;
      }
      return h;
    };
#pragma unroll
    for (int j = 0; j < NS; j++) {
      double v = 0.0;
      if (j < ns) {
        v = gs;
#pragma unroll
        for (int c = 0; c < D; c++) v *= hdeg(c, s_deg[grp][j][c]);
      }
      A[j] = v;
    }
    double prod = 1.0;
    bool used = !mine;
#pragma unroll
    for (int c = 0; c < NS; c++) {
      if (c >= nsmax) break;        // (kernel-uniform)
      const bool act = c < ns;      // (uniform within the group)
      unsigned key = (!used && act) ? ((__float_as_uint((float)fabs(A[c])) & ~15u) | (unsigned)(15 - r)) : 0u;
      {
        unsigned o = (unsigned)__builtin_amdgcn_mov_dpp((int)key, 0xB1, 0xF, 0xF, true); key = o > key ? o : key;      // lane ^ 1
        o = (unsigned)__builtin_amdgcn_mov_dpp((int)key, 0x4E, 0xF, 0xF, true); key = o > key ? o : key;               // lane ^ 2
        o = (unsigned)__builtin_amdgcn_mov_dpp((int)key, 0x124, 0xF, 0xF, true); key = o > key ? o : key;              // row_ror:4
        o = (unsigned)__builtin_amdgcn_mov_dpp((int)key, 0x128, 0xF, 0xF, true); key = o > key ? o : key;              // row_ror:8
      }
      const int who = 15 - (int)(key & 15u);
      const double piv = ff_lane_read(A[c], (lane & ~15) | who);
      if (act) prod *= piv;
      const bool ispiv = act && who == r && !used;
      if (ispiv) {
#pragma unroll
        for (int j = 0; j < NS; j++) s_row[c & 1][grp][j] = A[j];
        used = true;
      }
      __syncthreads();
      if (act && !used) {
        const double f = A[c] * ff_rcp(piv);
#pragma unroll
        for (int j = 0; j < NS; j++) { if (j > c) A[j] = fma(-f, s_row[c & 1][grp][j], A[j]); }
      }
      // (no second barrier: column c + 1 publishes into the other buffer, and whoever writes this one again at column c + 2 has passed
      // the barrier of column c + 1, which every reader of column c reaches only after its reads)
    }
    __syncthreads();      // (the next call's first column writes buffer 0 again)
    return prod;
  };
  auto logabsdet = [&](const double* xx) -> double { return log(fabs(detprod(xx, true))); };
  // log p of the walker = 2 (log|det up| + log|det down|): the two groups of a walker exchange their sums
  auto logprob = [&](const double* xx) -> double {
    const double mysum = logabsdet(xx);
    const double other = ff_lane_read(mysum, lane ^ 16);
    const double up = sp ? other : mysum, dn = sp ? mysum : other;
    double sacc = 0.0;
    if (nup) sacc += up;
    if (ndn) sacc += dn;
    return 2.0 * sacc;
  };
  // this lane's D normals of (walker, step)
  auto normals = [&](uint32_t step, double* z) {
    double z4[4] = {0.0, 0.0, 0.0, 0.0};
    int have = -1;
#pragma unroll
    for (int c = 0; c < D; c++) {
      const int q = (i0 + c) >> 2;
      if (q != have) { ff_normal_quad(seed, wid, step, (uint32_t)q, z4); have = q; }
      const int e = (i0 + c) & 3;
      z[c] = e == 0 ? z4[0] : (e == 1 ? z4[1] : (e == 2 ? z4[2] : z4[3]));
    }
  };

  double x[D], nx[D];
  if constexpr (!NOISE) {
    // Philox-fed chain (round 4), the same stream and the same walkers as ff_rng_fill + the noise-fed branch below:
    //  * the walker's Philox blocks -- ceil(M / 4) quads of normals and the block of the uniform -- are dealt over its 32 lanes, one
    //    each, and meet in LDS (every lane used to evaluate the blocks of its own coordinates and the uniform's: up to three per
    //    lane and step, six times the walker's sixteen);
    //  * the decision u < |psi(x')|^2 / |psi(x)|^2 is taken as u e^{R' - R} (P_up P_dn)^2 < (P'_up P'_dn)^2 on the determinants of the
    //    polynomial parts of the orbitals, R = sum r_i^2 (ff_mcmc_spin_philox_kernel): no exp per particle, no log per determinant.
    // (capacity: the largest walker the entry points admit -- D * 2 FF_MAX_NS coordinates, rounded up to whole quads; 64 was too small
    //  for more than 21 particles in d = 3: ADVICE r04)
    constexpr int NRM = 4 * ((D * 2 * FF_MAX_NS + 3) / 4);
    __shared__ double s_nrm[2][NRM], s_uu[2];
    const int wk = lane >> 5, L = lane & 31, nq = (M + 3) >> 2;
    auto draw = [&](uint32_t step) {
      if (L < nq) {
        const ff_u4 rw = ff_philox(seed, wid, step, (uint32_t)L);
        double z4[4];
        ff_normal_pair32(rw.x, rw.y, z4[0], z4[1]);
        ff_normal_pair32(rw.z, rw.w, z4[2], z4[3]);
#pragma unroll
        for (int k = 0; k < 4; k++) s_nrm[wk][4 * L + k] = z4[k];
      } else if (L == nq) {
        s_uu[wk] = ff_uniform(seed, wid, step, D == 2 ? (uint32_t)n : 0xffffu);
      }
      __syncthreads();
    };
    // sum over the walker's 32 lanes, taken from its first lane so that every lane holds the same bits
    auto wsum = [&](double v) -> double {
#define FF_DPP_ADD(a, ctrl) ((a) + __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(a), ctrl, 0xF, 0xF, true), \
                                                   __builtin_amdgcn_mov_dpp(__double2loint(a), ctrl, 0xF, 0xF, true)))
      v = FF_DPP_ADD(v, 0xB1); v = FF_DPP_ADD(v, 0x4E); v = FF_DPP_ADD(v, 0x124); v = FF_DPP_ADD(v, 0x128);
#undef FF_DPP_ADD
      v += ff_lane_read(v, lane ^ 16);
      return ff_lane_read(v, lane & ~31);
    };
    auto r2 = [&](const double* xx) -> double {
      double t = 0.0;
#pragma unroll
      for (int c = 0; c < D; c++) t = fma(xx[c], xx[c], t);
      return mine ? t : 0.0;
    };
    auto polyprod2 = [&](const double* xx) -> double {      // (P_up P_dn)^2
      const double mp = detprod(xx, false), pp = mp * ff_lane_read(mp, lane ^ 16);
      return pp * pp;
    };
    if (g0 != nullptr) {
#pragma unroll
      for (int c = 0; c < D; c++) x[c] = g0[b * M + i0 + c];
    } else {
      draw(0u);
#pragma unroll
      for (int c = 0; c < D; c++) x[c] = s_nrm[wk][i0 + c];
    }
    double Rl = r2(x), PP2 = polyprod2(x);
    int nacc = 0;
    for (int s = 0; s < steps; s++) {
      draw((uint32_t)(s + 1));
#pragma unroll
      for (int c = 0; c < D; c++) nx[c] = ff3_add_rn(x[c], ff3_mul_rn(tau, s_nrm[wk][i0 + c]));
      const double uu = s_uu[wk];
      const double Rn = r2(nx), dRt = wsum(Rl - Rn);
      const double lhs = uu * exp(fmin(fmax(-dRt, -700.0), 708.0)) * PP2;
      const double PPn2 = polyprod2(nx);
      const bool acc = (dRt >= -708.0) & (lhs < PPn2);      // IEEE: a NaN on either side rejects (src/base_dist.py:67-68)
      if (acc) {
#pragma unroll
        for (int c = 0; c < D; c++) x[c] = nx[c];
        Rl = Rn;
        PP2 = PPn2;
        nacc++;
      }
    }
    const double logp = logprob(x);
    if (!live) return;
    if (mine) {
#pragma unroll
      for (int c = 0; c < D; c++) x_out[b * M + i0 + c] = x[c];
    }
    if (sp == 0 && r == 0) {
      if (logp_out) logp_out[b] = logp;
      if (acc_count) acc_count[b] = nacc;
    }
    return;
  }
  if (NOISE || g0 != nullptr) {
#pragma unroll
    for (int c = 0; c < D; c++) x[c] = g0[b * M + i0 + c];
  } else {
    normals(0u, x);
  }
  double logp = logprob(x);
  int nacc = 0;
  for (int s = 0; s < steps; s++) {
    double z[D], uu;
    if (NOISE) {
#pragma unroll
      for (int c = 0; c < D; c++) z[c] = g[((int64_t)s * B + b) * M + i0 + c];
      uu = u[(int64_t)s * B + b];
    } else {
      normals((uint32_t)(s + 1), z);
      uu = ff_uniform(seed, wid, (uint32_t)(s + 1), D == 2 ? (uint32_t)n : 0xffffu);
    }
#pragma unroll
    for (int c = 0; c < D; c++) nx[c] = ff3_add_rn(x[c], ff3_mul_rn(tau, z[c]));
    const double nl = logprob(nx);
    const double p = exp(nl - logp);
    const bool acc = uu < p;          // IEEE comparison: NaN rejects, +inf accepts (src/base_dist.py:67-68)
    if (acc) {
#pragma unroll
      for (int c = 0; c < D; c++) x[c] = nx[c];
      logp = nl;
      nacc++;
    }
    if (accept && live && sp == 0 && r == 0) accept[(int64_t)s * B + b] = acc ? 1 : 0;
  }
  if (!live) return;
  if (mine) {
#pragma unroll
    for (int c = 0; c < D; c++) x_out[b * M + i0 + c] = x[c];
  }
  if (sp == 0 && r == 0) {
    if (logp_out) logp_out[b] = logp;
    if (acc_count) acc_count[b] = nacc;
  }
}

// launch for both dimensions (ff_walkers.hip hands its general d = 2 shapes over)
int ff_mcmc_rows_launch(void* stream, int d, bool noise, int64_t B, int nup, int ndn, const int* tu, const int* td, const int* ws, int steps,
                        double tau, const double* g0, const double* g, const double* u, uint64_t seed, int64_t woff, double* x_out,
                        double* logp_out, uint8_t* accept, int* acc_count) {
  const unsigned grid = (unsigned)((2 * B + 3) / 4);
  const int nsm = nup > ndn ? nup : ndn;
#define FF_MR(D_, N_) do { \
    if (nsm <= 8) FF_LAUNCH((ff_mcmc_rows_kernel<D_, N_, 8>), grid, FF_WAVE, stream, B, nup, ndn, tu, td, ws, steps, tau, g0, g, u, seed, woff, x_out, logp_out, accept, acc_count); \
    else if (nsm <= 10) FF_LAUNCH((ff_mcmc_rows_kernel<D_, N_, 10>), grid, FF_WAVE, stream, B, nup, ndn, tu, td, ws, steps, tau, g0, g, u, seed, woff, x_out, logp_out, accept, acc_count); \
    else FF_LAUNCH((ff_mcmc_rows_kernel<D_, N_, FF_MAX_NS>), grid, FF_WAVE, stream, B, nup, ndn, tu, td, ws, steps, tau, g0, g, u, seed, woff, x_out, logp_out, accept, acc_count); \
  } while (0)
  if (d == 2) { if (noise) FF_MR(2, true); else FF_MR(2, false); }
  else { if (noise) FF_MR(3, true); else FF_MR(3, false); }
#undef FF_MR
  return hipGetLastError() == hipSuccess ? FF_OK : FF_ELAUNCH;
}

// ---- fp32: Backflow.forward / .divergence with the sigmoid sums, radii and accumulations in single precision
// (walkers and results stay fp64 arrays at the boundary; the arithmetic in between is what an fp32 path would run)
FF_D void ff_mlp_point_f32(int H, const double* __restrict__ w1, const double* __restrict__ b1, const double* __restrict__ w2,
                           float r, float& val, float& dval) {
  float s = 0.f, g = 0.f;
  for (int h = 0; h < H; h++) {
    const float a = fmaf((float)w1[h], r, (float)b1[h]);
    const float sg = 1.0f / (1.0f + __expf(-a));
    s = fmaf((float)w2[h], sg, s);
    g = fmaf((float)w2[h] * (float)w1[h], sg * (1.0f - sg), g);
  }
  val = s; dval = g;
}

__global__ void __launch_bounds__(128)
ff_backflow_f32_kernel(int64_t B, int n, int d, ff_net net, const double* __restrict__ x, double* __restrict__ v,
                       double* __restrict__ div) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int M = n * d;
  float xl[3 * 24], vl[3 * 24];
  for (int i = 0; i < M; i++) { xl[i] = (float)x[b * M + i]; vl[i] = 0.f; }
  float dv = 0.f;
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++) {
      float rho[3], r2 = 0.f;
      for (int c = 0; c < d; c++) { rho[c] = xl[i * d + c] - xl[j * d + c]; r2 = fmaf(rho[c], rho[c], r2); }
      float r = sqrtf(r2), eta, deta;
      ff_mlp_point_f32(net.He, net.ew1, net.eb1, net.ew2, r, eta, deta);
      for (int c = 0; c < d; c++) { vl[i * d + c] = fmaf(eta, rho[c], vl[i * d + c]); vl[j * d + c] = fmaf(-eta, rho[c], vl[j * d + c]); }
      dv += 2.0f * fmaf(deta, r, d * eta);
    }
  if (net.Hm > 0)
    for (int i = 0; i < n; i++) {
      float r2 = 0.f;
      for (int c = 0; c < d; c++) r2 = fmaf(xl[i * d + c], xl[i * d + c], r2);
      float r = sqrtf(r2), mu, dmu;
      ff_mlp_point_f32(net.Hm, net.mw1, net.mb1, net.mw2, r, mu, dmu);
      for (int c = 0; c < d; c++) vl[i * d + c] = fmaf(mu, xl[i * d + c], vl[i * d + c]);
      dv += fmaf(dmu, r, d * mu);
    }
  if (v) for (int i = 0; i < M; i++) v[b * M + i] = (double)vl[i];
  if (div) div[b] = (double)dv;
}


// ---------------------------------------------------------------------------------------------------
// Local-energy finish in d = 3 (the d = 2 one: ff_eloc_slater_kernel / ff_eloc_contract_kernel in ff_cnf_fwd.hip): the
// sensitivities come from ff_eloc_sensitivities(..., d = 3) (row-layout kernel); here the Slater gradient / Hessian at z(t0):
//   grad_i = g0 . u_i - dDelta_i,   lap = sum_i u_i^T H0 u_i + g0 . kbar - sum_i L_i,   E_loc = -lap/4 - |grad|^2/8 + V(x)
// with H0 of 2 log|det|: same-particle block sum_j hess phi_j(r_a) Dinv_ja - g_a g_a^T, cross block -T_ac (x) T_ca
// (T_ac = sum_j grad phi_j(r_a) Dinv_jc) -- SURVEY.md A.2 / A.6 with one more coordinate.
// Q slots per walker: [0,M) g0 | [M, M+6n) S (xx,xy,xz,yy,yz,zz per particle) | T_up (3 nup^2, component-major) | T_dn | 2 log|det| per spin
// Slater table of the local-energy finish, SIXTEEN LANES PER DETERMINANT (two walkers x two spin species per wave): lane r
// of a group owns particle r of its species -- row r of D_ij = phi_j(r_i) and of the unit matrix beside it.  Gauss-Jordan
// with partial pivoting, without moving rows: per column the unused lane with the largest entry becomes the pivot (16-lane
// butterfly), normalises its row, publishes it through LDS, everyone else eliminates; the lane that was the pivot of column c
// ends up holding row c of D^-1.  Then lane a forms its particle's rows of T[comp][a][b] = sum_j d_comp phi_j(r_a) Dinv[j][b]
// and the same-particle Hessian sums S.  Everything is statically indexed (loops over FF_MAX_NS with a
// predicate): no private-memory arrays -- the one-lane-per-determinant kernel this replaces kept two 12 x 12 matrices in
// scratch and took 30 ms per 131 072 walkers of 20 particles.  Q layout: see ff_eloc_contract3d_kernel / ff_eloc_contract_kernel.
template <int D>
__global__ void __launch_bounds__(FF_WAVE)
ff_eloc_slater_rows_kernel(int64_t B, int nup, int ndn, const int* __restrict__ tab_up, const int* __restrict__ tab_dn,
                           const int* __restrict__ wstate, const double* __restrict__ z0, double* __restrict__ Q) {
  __shared__ ff_slater_rows_smem<4> sm;
  const int lane = threadIdx.x, grp = lane >> 4;
  const int64_t gid = (int64_t)blockIdx.x * 4 + grp;
  const int64_t b = gid >> 1;
  const int sp = (int)(gid & 1);
  const bool live = b < B;
  const int n = nup + ndn, M = D * n;
  const int st = (live && wstate) ? wstate[b] : 0;
  const int64_t nq = M + (D * (D + 1) / 2) * n + D * (nup * nup + ndn * ndn) + 2;
  ff_slater_rows_body<D, 4>(sm, lane, grp, live, sp, nup, ndn, tab_up, tab_dn, st, z0 + (live ? b : 0) * M, Q + (live ? b : 0) * nq);
}

// launch for both dimensions (the d = 2 finish of ff_cnf_fwd.hip uses it beyond the register-resident 4 x 4 determinants)
int ff_slater_rows_launch(void* stream, int d, int64_t B, int nup, int ndn, const int* tab_up, const int* tab_dn, const int* wstate,
                          const double* z0, double* Q) {
  const unsigned grid = (unsigned)((2 * B + 3) / 4);
  if (d == 2) FF_LAUNCH((ff_eloc_slater_rows_kernel<2>), grid, FF_WAVE, stream, B, nup, ndn, tab_up, tab_dn, wstate, z0, Q);
  else FF_LAUNCH((ff_eloc_slater_rows_kernel<3>), grid, FF_WAVE, stream, B, nup, ndn, tab_up, tab_dn, wstate, z0, Q);
  return hipGetLastError() == hipSuccess ? FF_OK : FF_ELAUNCH;
}

static size_t ff_contract3d_lds_bytes(int nup, int ndn) {
  const int n = nup + ndn, M = 3 * n, G = FF_WAVE / M, nq = M + 6 * n + 3 * (nup * nup + ndn * ndn) + 2;
  return sizeof(double) * ((size_t)G * M * M + (size_t)G * nq + 4 * FF_WAVE);
}
__global__ void __launch_bounds__(FF_WAVE)
ff_eloc_contract3d_kernel(int64_t B, int nup, int ndn, double Zc, int use_ho, const double* __restrict__ x,
                          const double* __restrict__ Q, const double* __restrict__ Jt, const double* __restrict__ kbar,
                          const double* __restrict__ dD, const double* __restrict__ delta, const double* __restrict__ Lpart,
                          double* __restrict__ logp, double* __restrict__ grad, double* __restrict__ lap,
                          double* __restrict__ V, double* __restrict__ eloc, double* __restrict__ glogp0) {
  FF_DYN_LDS(ff_fin3_lds);
  const int n = nup + ndn, M = 3 * n, G = FF_WAVE / M;
  const int lane = threadIdx.x, g = lane / M, i = lane - g * M;
  const bool ingrp = g < G;
  const int nq = M + 6 * n + 3 * (nup * nup + ndn * ndn) + 2;
  double* const s_u = ff_fin3_lds;                           // [g][i][k] = dz_k/dx_i
  double* const s_q = s_u + G * M * M;
  double* const s_x = s_q + G * nq;
  double (*const s_red)[FF_WAVE] = (double (*)[FF_WAVE])(s_x + FF_WAVE);
  const int64_t ngroups = (B + G - 1) / G;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t b0 = grp * G, b = b0 + g;
    const bool valid = ingrp && b < B;
    const int nw = (int)((B - b0) < G ? (B - b0) : G);
    __syncthreads();
    for (int e = lane; e < nw * M * M; e += FF_WAVE) s_u[e] = Jt[b0 * M * M + e];
    for (int e = lane; e < nw * nq; e += FF_WAVE) s_q[e] = Q[b0 * nq + e];
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
        const double* T = qw + M + 6 * n + (sp ? 3 * nup * nup : 0);
        double q = 0.0;
        for (int a = 0; a < ns; a++) {
          const double* ua = u + 3 * (off + a);
          const double* Sa = qw + M + 6 * (off + a);
          q += ua[0] * ua[0] * Sa[0] + 2.0 * ua[0] * ua[1] * Sa[1] + 2.0 * ua[0] * ua[2] * Sa[2] + ua[1] * ua[1] * Sa[3]
             + 2.0 * ua[1] * ua[2] * Sa[4] + ua[2] * ua[2] * Sa[5];
          for (int c = 0; c < ns; c++) {
            const double* uc = u + 3 * (off + c);
            double Wac = 0.0, Wca = 0.0;
            for (int cm = 0; cm < 3; cm++) {
              Wac = fma(ua[cm], T[cm * ns * ns + a * ns + c], Wac);
              Wca = fma(uc[cm], T[cm * ns * ns + c * ns + a], Wca);
            }
            q -= Wac * Wca;
          }
        }
        hq += 2.0 * q;
      }
      lap_i = hq - Lpart[b * M + i] + g0[i] * kbar[b * M + i];
      if (i % 3 == 0) {     // the lane of particle a's x-coordinate takes a's trap term and its pairs with the later particles
        const int a = i / 3;
        const double* xs = s_x + g * M;
        double pair = 0.0;
        for (int c = a + 1; c < n; c++) {
          const double dx = xs[3 * a] - xs[3 * c], dy = xs[3 * a + 1] - xs[3 * c + 1], dz = xs[3 * a + 2] - xs[3 * c + 2];
          pair += Zc / sqrt(dx * dx + dy * dy + dz * dz);
        }
        v_i = pair + (use_ho ? 0.5 * (xs[3 * a] * xs[3 * a] + xs[3 * a + 1] * xs[3 * a + 1] + xs[3 * a + 2] * xs[3 * a + 2]) : 0.0);
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

// The same contraction for walkers that fill a wave alone (M = 3 n > 32; BASELINE configs[4]: M = 60): ONE WALKER PER WORKGROUP
// of four waves, FOUR lanes per direction i -- they split the particles a of the Hessian sum and the components k of g0 . u_i
// (quad reductions by DPP).  The one-wave version above walks 2 x 10 x 10 (a, c) pairs per lane back to back, each a handful of
// dependent LDS reads, with three waves resident per CU (42 KB of LDS each): 10.5 ms per 131 072 walkers at 20 particles.
__global__ void __launch_bounds__(4 * FF_WAVE)
ff_wide_contract3d_kernel(int64_t B, int nup, int ndn, double Zc, int use_ho, const double* __restrict__ x,
                          const double* __restrict__ Q, const double* __restrict__ Jt, const double* __restrict__ kbar,
                          const double* __restrict__ dD, const double* __restrict__ delta, const double* __restrict__ Lpart,
                          double* __restrict__ logp, double* __restrict__ grad, double* __restrict__ lap,
                          double* __restrict__ V, double* __restrict__ eloc, double* __restrict__ glogp0) {
  FF_DYN_LDS(ff_fin4_lds);
  const int n = nup + ndn, M = 3 * n;
  const int nq = M + 6 * n + 3 * (nup * nup + ndn * ndn) + 2;
  const int tid = threadIdx.x, i = tid >> 2, s = tid & 3;
  const bool valid = i < M;
  double* const s_u = ff_fin4_lds;                           // [i][k] = dz_k/dx_i
  double* const s_q = s_u + M * M;
  double* const s_x = s_q + nq;
  double (*const s_red)[FF_WAVE] = (double (*)[FF_WAVE])(s_x + FF_WAVE);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    for (int e = tid; e < M * M; e += 4 * FF_WAVE) s_u[e] = Jt[b * M * M + e];
    for (int e = tid; e < nq; e += 4 * FF_WAVE) s_q[e] = Q[b * nq + e];
    if (tid < M) s_x[tid] = x[b * M + tid];
    __syncthreads();
    const double* u = s_u + (valid ? i : 0) * M;
    const double* g0 = s_q;
    double gi = 0.0, hq = 0.0;
    for (int k = s; k < M; k += 4) gi = fma(g0[k], u[k], gi);
    for (int sp = 0; sp < 2; sp++) {
      const int ns = sp ? ndn : nup, off = sp ? nup : 0;
      const double* T = s_q + M + 6 * n + (sp ? 3 * nup * nup : 0);
      double q = 0.0;
      // sum_{a,c} P_ac P_ca (P_ac = u_a . T[.][a][c]) is symmetric under a <-> c: the pairs a <= c only, the off-diagonal ones twice.
      // The kernel is bound by these LDS reads (nine per pair: 3.65 ms per 131 072 walkers at 20 particles with all ns^2 pairs).  The
      // particles a are dealt to the four lanes of a direction in boustrophedon order -- a costs ns - a pairs -- so that the longest
      // lane has 15 pairs of the 55 at ns = 10.
      for (int r4 = 0; 4 * r4 < ns; r4++) {
        const int a = 4 * r4 + ((r4 & 1) ? 3 - s : s);
        if (a >= ns) continue;
        const double* ua = u + 3 * (off + a);
        const double* Sa = s_q + M + 6 * (off + a);
        q += ua[0] * ua[0] * Sa[0] + 2.0 * ua[0] * ua[1] * Sa[1] + 2.0 * ua[0] * ua[2] * Sa[2] + ua[1] * ua[1] * Sa[3]
           + 2.0 * ua[1] * ua[2] * Sa[4] + ua[2] * ua[2] * Sa[5];
        for (int c = a; c < ns; c++) {
          const double* uc = u + 3 * (off + c);
          double Wac = 0.0, Wca = 0.0;
#pragma unroll
          for (int cm = 0; cm < 3; cm++) {
            Wac = fma(ua[cm], T[cm * ns * ns + a * ns + c], Wac);
            Wca = fma(uc[cm], T[cm * ns * ns + c * ns + a], Wca);
          }
          q -= (c == a ? 1.0 : 2.0) * Wac * Wca;
        }
      }
      hq += 2.0 * q;
    }
    gi += ff_swap1(gi); gi += ff_swap2(gi);
    hq += ff_swap1(hq); hq += ff_swap2(hq);
    double lap_i = 0.0, v_i = 0.0, g2 = 0.0;
    if (valid && s == 0) {
      gi -= dD[b * M + i];
      lap_i = hq - Lpart[b * M + i] + g0[i] * kbar[b * M + i];
      if (i % 3 == 0) {     // the lane of particle a's x-coordinate takes a's trap term and its pairs with the later particles
        const int a = i / 3;
        double pair = 0.0;
        for (int c = a + 1; c < n; c++) {
          const double dx = s_x[3 * a] - s_x[3 * c], dy = s_x[3 * a + 1] - s_x[3 * c + 1], dz = s_x[3 * a + 2] - s_x[3 * c + 2];
          pair += Zc / sqrt(dx * dx + dy * dy + dz * dz);
        }
        v_i = pair + (use_ho ? 0.5 * (s_x[3 * a] * s_x[3 * a] + s_x[3 * a + 1] * s_x[3 * a + 1] + s_x[3 * a + 2] * s_x[3 * a + 2]) : 0.0);
      }
      g2 = gi * gi;
      if (grad) grad[b * M + i] = gi;
      if (glogp0) glogp0[b * M + i] = g0[i];
      s_red[0][i] = g2; s_red[1][i] = lap_i; s_red[2][i] = v_i;
    }
    __syncthreads();
    if (tid == 0) {
      double g2s = 0.0, lapv = 0.0, Vv = 0.0;
      for (int k = 0; k < M; k++) { g2s += s_red[0][k]; lapv += s_red[1][k]; Vv += s_red[2][k]; }
      if (logp) logp[b] = (s_q[nq - 2] + s_q[nq - 1]) - delta[b];
      if (lap) lap[b] = lapv;
      if (V) V[b] = Vv;
      if (eloc) eloc[b] = -0.25 * lapv - 0.125 * g2s + Vv;
    }
  }
}

// The very noise ff_mcmc_sample3d consumes, materialised (tests: feed it to ff_mcmc_sample_noise3d) -- quads 0 .. ceil(3n / 4) - 1 of
// (walker, step) are the normals of the walker's coordinates in order, block 0xffff its uniform (ff_mcmc_rows_kernel<3, false>).
__global__ void __launch_bounds__(128)
ff_rng_fill3d_kernel(int64_t B, int n, int steps, uint64_t seed, int64_t woff, double* __restrict__ g0, double* __restrict__ g,
                     double* __restrict__ u) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int M = 3 * n, nq = (M + 3) >> 2;
  const uint64_t wid = (uint64_t)(woff + b);
  for (int s = 0; s <= steps; s++) {
    double* gs = s == 0 ? g0 + b * M : g + ((int64_t)(s - 1) * B + b) * M;
    for (int q = 0; q < nq; q++) {
      double z4[4];
      ff_normal_quad(seed, wid, (uint32_t)s, (uint32_t)q, z4);
      for (int k = 0; k < 4; k++) if (4 * q + k < M) gs[4 * q + k] = z4[k];
    }
    if (s > 0) u[(int64_t)(s - 1) * B + b] = ff_uniform(seed, wid, (uint32_t)s, 0xffffu);
  }
}

// =================================================================================================
extern void ff_set_error(const char* msg);
#define FF_CHECK(cond, code, msg) do { if (!(cond)) { ff_set_error(msg); return code; } } while (0)
#define FF_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { ff_set_error(hipGetErrorString(e_)); return FF_ELAUNCH; } } while (0)
static unsigned ff3_grid(int64_t B, int block) { return (unsigned)((B + block - 1) / block); }

extern "C" {

int ff_logprob3d(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                 const int32_t* walker_state, const double* x, double* logp, double* grad, double* lap) {
  FF_CHECK(B >= 0 && nup >= 0 && ndn >= 0 && nup + ndn > 0 && x && logp, FF_EINVAL, "ff_logprob3d: bad argument");
  FF_CHECK((nup == 0 || tab_up) && (ndn == 0 || tab_dn) && ((grad == nullptr) == (lap == nullptr)), FF_EINVAL, "ff_logprob3d: null pointer");
  FF_CHECK(nup <= FF_MAX_NS && ndn <= FF_MAX_NS, FF_EUNSUPPORTED, "ff_logprob3d: determinant larger than FF_MAX_NS");
  if (B == 0) return FF_OK;
  FF_LAUNCH(ff_logprob3d_kernel, ff3_grid(B, 64), 64, stream, B, nup, ndn, tab_up, tab_dn, walker_state, x, logp, grad, lap);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_mcmc_sample_noise3d(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                           const int32_t* walker_state, int steps, double tau, const double* g0, const double* g, const double* u,
                           double* x_out, double* logp_out, uint8_t* accept) {
  FF_CHECK(B >= 0 && nup >= 0 && ndn >= 0 && nup + ndn > 0 && steps >= 0 && x_out && g0 && (steps == 0 || (g && u)), FF_EINVAL,
           "ff_mcmc_sample_noise3d: bad argument");
  FF_CHECK((nup == 0 || tab_up) && (ndn == 0 || tab_dn), FF_EINVAL, "ff_mcmc_sample_noise3d: null orbital table");
  FF_CHECK(nup <= FF_MAX_NS && ndn <= FF_MAX_NS, FF_EUNSUPPORTED, "ff_mcmc_sample_noise3d: determinant larger than FF_MAX_NS");
  if (B == 0) return FF_OK;
  return ff_mcmc_rows_launch(stream, 3, true, B, nup, ndn, tab_up, tab_dn, walker_state, steps, tau, g0, g, u, (uint64_t)0, (int64_t)0, x_out,
                             logp_out, accept, (int*)nullptr);
}

int ff_mcmc_sample3d(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                     const int32_t* walker_state, int steps, double tau, uint64_t seed, int64_t walker_offset,
                     double* x_out, double* logp_out, int32_t* accept_count) {
  FF_CHECK(B >= 0 && nup >= 0 && ndn >= 0 && nup + ndn > 0 && steps >= 0 && x_out, FF_EINVAL, "ff_mcmc_sample3d: bad argument");
  FF_CHECK((nup == 0 || tab_up) && (ndn == 0 || tab_dn), FF_EINVAL, "ff_mcmc_sample3d: null orbital table");
  FF_CHECK(nup <= FF_MAX_NS && ndn <= FF_MAX_NS, FF_EUNSUPPORTED, "ff_mcmc_sample3d: determinant larger than FF_MAX_NS");
  if (B == 0) return FF_OK;
  return ff_mcmc_rows_launch(stream, 3, false, B, nup, ndn, tab_up, tab_dn, walker_state, steps, tau, (const double*)nullptr,
                             (const double*)nullptr, (const double*)nullptr, seed, walker_offset, x_out, logp_out, (uint8_t*)nullptr, accept_count);
}

int ff_rng_fill3d(void* stream, int64_t B, int n, int steps, uint64_t seed, int64_t walker_offset, double* g0, double* g, double* u) {
  FF_CHECK(B >= 0 && n > 0 && n <= 2 * FF_MAX_NS && steps >= 0 && g0 && (steps == 0 || (g && u)), FF_EINVAL, "ff_rng_fill3d: bad argument");
  if (B == 0) return FF_OK;
  FF_LAUNCH(ff_rng_fill3d_kernel, ff3_grid(B, 128), 128, stream, B, n, steps, seed, walker_offset, g0, g, u);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_backflow_v_div_f32(void* stream, int64_t B, int n, int d, const ff_net* net, const double* x, double* v, double* div) {
  FF_CHECK(B >= 0 && n > 0 && d > 0 && net && x && (v || div), FF_EINVAL, "ff_backflow_v_div_f32: bad argument");
  FF_CHECK(net->He > 0 && net->ew1 && net->eb1 && net->ew2 && (net->Hm == 0 || (net->mw1 && net->mb1 && net->mw2)), FF_EINVAL,
           "ff_backflow_v_div_f32: bad net");
  FF_CHECK(n <= 24 && d <= 3, FF_EUNSUPPORTED, "ff_backflow_v_div_f32: n > 24 or d > 3");
  if (B == 0) return FF_OK;
  FF_LAUNCH(ff_backflow_f32_kernel, ff3_grid(B, 128), 128, stream, B, n, d, *net, x, v, div);
  FF_LAUNCH_CHECK();
  return FF_OK;
}


int ff_eloc_finish3d(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                     const int32_t* walker_state, double Z, int use_ho, const double* x, const void* workspace,
                     double* logp, double* grad, double* lap, double* V, double* eloc, double* z_out, double* dlogp_out,
                     double* glogp0_out) {
  const int n = nup + ndn;
  FF_CHECK(B >= 0 && nup >= 0 && ndn >= 0 && n > 0 && x && workspace, FF_EINVAL, "ff_eloc_finish3d: bad argument");
  FF_CHECK((nup == 0 || tab_up) && (ndn == 0 || tab_dn), FF_EINVAL, "ff_eloc_finish3d: null orbital table");
  FF_CHECK(nup <= FF_MAX_NS && ndn <= FF_MAX_NS && 3 * n <= FF_WAVE, FF_EUNSUPPORTED, "ff_eloc_finish3d: walker too large");
  if (B == 0) return FF_OK;
  const size_t M = (size_t)n * 3;
  ff_eloc_ws w = ff_eloc_carve((void*)workspace, B, (size_t)n, 3);
  if (ff_slater_rows_launch(stream, 3, B, nup, ndn, tab_up, tab_dn, walker_state, (const double*)w.z0, w.Q) != FF_OK) return FF_ELAUNCH;
  if (3 * n > 32) {      // a walker per workgroup, four lanes per direction
    const size_t lds = sizeof(double) * ((size_t)9 * n * n + (size_t)(3 * n + 6 * n + 3 * (nup * nup + ndn * ndn) + 2) + 4 * FF_WAVE);
    FF_LAUNCH_LDS(ff_wide_contract3d_kernel, (unsigned)(B < 65536 ? B : 65536), 4 * FF_WAVE, lds, stream, B, nup, ndn, Z, use_ho, x,
                  (const double*)w.Q, (const double*)w.Jt, (const double*)w.kbar, (const double*)w.dD, (const double*)w.dl,
                  (const double*)w.Lp, logp, grad, lap, V, eloc, glogp0_out);
  } else {
    const int Gf = FF_WAVE / (3 * n);
    const int64_t ng = (B + Gf - 1) / Gf;
    FF_LAUNCH_LDS(ff_eloc_contract3d_kernel, (unsigned)(ng < 32768 ? ng : 32768), FF_WAVE, ff_contract3d_lds_bytes(nup, ndn), stream, B, nup, ndn, Z,
                  use_ho, x, (const double*)w.Q, (const double*)w.Jt, (const double*)w.kbar, (const double*)w.dD, (const double*)w.dl,
                  (const double*)w.Lp, logp, grad, lap, V, eloc, glogp0_out);
  }
  FF_LAUNCH_CHECK();
  if (z_out && hipMemcpyAsync(z_out, w.z0, sizeof(double) * (size_t)B * M, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return FF_ELAUNCH;
  if (dlogp_out && hipMemcpyAsync(dlogp_out, w.dl, sizeof(double) * (size_t)B, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return FF_ELAUNCH;
  return FF_OK;
}

}  // extern "C"
