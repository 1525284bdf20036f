// ff_walkers.hip -- one-lane-per-walker kernels: Metropolis sweep, Slater log-determinants and their
// derivatives, potentials, stand-alone backflow/MLP evaluation, energy moments.
//
// Data layout in HBM: walker coordinates (B, n, d) row-major fp64 exactly as the reference's tensors, so a
// wave of 64 consecutive walkers reads one contiguous 64*n*d*8-byte span (fully used cache lines).
// The Metropolis sweep keeps a walker's coordinates and log-probability in VGPRs for all `steps` proposals:
// per walker-step the only HBM traffic is the proposal noise (parity mode) or nothing at all (Philox mode).
#include <stdlib.h>
#include "ff_common.h"
#include "ff_slater.h"
#include "ff_rng.h"

#define FF_MAX_N 24  // particles per walker for the generic paths

// x + tau*g exactly as torch computes it (two roundings): hipcc's default -ffp-contract=fast would fuse
// the pair into one FMA and the walkers would no longer be bit-identical to the reference's.
// the empty asm makes the product opaque, so the backend cannot contract mul+add into v_fma_f64
FF_D double ff_mul_rn(double a, double b) { double r = a * b; FF_OPAQUE(r); return r; }
FF_D double ff_add_rn(double a, double b) { double r = a + b; FF_OPAQUE(r); return r; }

// ---------------------------------------------------------------------------------------------------
// FreeFermion.sample (src/base_dist.py:58-71).  NU/ND > 0: compile-time spin sizes, everything in VGPRs.
// NU = ND = -1: runtime sizes (private arrays).
// ou/od: orbital indices (generic path) -- or, for compile-time sizes, ou = [nx | ny] degrees of the up
// orbitals and od likewise for the down orbitals (decoded once, outside the step loop).
template <int NU, int ND>
FF_D double ff_logprob_value(int nup, int ndn, const int* ou, const int* od, const double* x, int md) {
  double s = 0.0;
  if constexpr (NU >= 0) {
    if constexpr (NU > 0) s += ff_slater_logabsdet_reg<NU>(ou, ou + NU, x, md);
    if constexpr (ND > 0) s += ff_slater_logabsdet_reg<ND>(od, od + ND, x + 2 * NU, md);
  } else {
    if (nup) s += ff_slater_general(nup, ou, x, nullptr, nullptr);
    if (ndn) s += ff_slater_general(ndn, od, x + 2 * nup, nullptr, nullptr);
  }
  return 2.0 * s;
}

template <int NU, int ND, bool NOISE>
__global__ void __launch_bounds__(128)
ff_mcmc_kernel(int64_t B, int nup_rt, int ndn_rt, const int* __restrict__ tab_up, const int* __restrict__ tab_dn,
               const int* __restrict__ wstate, int steps, double tau,
               const double* __restrict__ g0, const double* __restrict__ g, const double* __restrict__ u,
               uint64_t seed, int64_t woff,
               double* __restrict__ x_out, double* __restrict__ logp_out, uint8_t* __restrict__ accept,
               int* __restrict__ acc_count) {
  constexpr bool FIXED = (NU >= 0);
  const int nup = FIXED ? NU : nup_rt, ndn = FIXED ? ND : ndn_rt;
  const int n = nup + ndn, M = 2 * n;
  constexpr int MAXM = FIXED ? 2 * (NU + ND) : 2 * FF_MAX_N;
  constexpr int MAXU = FIXED ? (NU > 0 ? NU : 1) : FF_MAX_NS, MAXD = FIXED ? (ND > 0 ? ND : 1) : FF_MAX_NS;
  __shared__ int s_md;
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (threadIdx.x == 0) s_md = 0;
  __syncthreads();
  const bool live = b < B;
  if (!live) b = B - 1;          // idle tail lanes shadow the last walker (they take part in the barriers)
  const int st = wstate ? wstate[b] : 0;
  int ou[2 * MAXU], od[2 * MAXD];
#pragma unroll
  for (int j = 0; j < MAXU; j++) {
    const int k = (j < nup) ? tab_up[st * nup + j] : 0;
    if constexpr (FIXED) ff_orb_decode(k, ou[j], ou[MAXU + j]); else ou[j] = k;
  }
#pragma unroll
  for (int j = 0; j < MAXD; j++) {
    const int k = (j < ndn) ? tab_dn[st * ndn + j] : 0;
    if constexpr (FIXED) ff_orb_decode(k, od[j], od[MAXD + j]); else od[j] = k;
  }
  // largest Hermite degree in the workgroup, as a scalar (wave-uniform loop bounds in ff_herm_rec)
  int md = 0;
  if constexpr (FIXED) {
#pragma unroll
    for (int j = 0; j < 2 * MAXU; j++) md = (j % MAXU < nup && ou[j] > md) ? ou[j] : md;
#pragma unroll
    for (int j = 0; j < 2 * MAXD; j++) md = (j % MAXD < ndn && od[j] > md) ? od[j] : md;
    atomicMax(&s_md, md);
    __syncthreads();
    md = FF_UNIFORM(s_md);
  }

  double x[MAXM], nx[MAXM];
  const uint64_t wid = (uint64_t)(woff + b);
  if (NOISE || g0 != nullptr) {   // explicit initial walkers (parity mode; ff_mcmc_continue)
#pragma unroll
    for (int i = 0; i < MAXM; i++) if (i < M) x[i] = g0[b * M + i];
  } else {
#pragma unroll
    for (int q = 0; q < (MAXM + 3) / 4; q++)
      if (2 * q < n) {
        double z4[4];
        ff_normal_quad(seed, wid, 0u, (uint32_t)q, z4);
#pragma unroll
        for (int k = 0; k < 4; k++) if (4 * q + k < MAXM && 4 * q + k < M) x[4 * q + k] = z4[k];
      }
  }
  double logp = ff_logprob_value<NU, ND>(nup, ndn, ou, od, x, md);
  int nacc = 0;
  // parity mode: the noise of step s+1 is requested from HBM before step s is computed (one step of software
  // pipelining: a walker's chain is serial, so without it every step would expose a full HBM round trip)
  double gq[MAXM], uq = 0.0;
  if (NOISE && steps > 0) {
#pragma unroll
    for (int i = 0; i < MAXM; i++) if (i < M) gq[i] = g[b * M + i];
    uq = u[b];
  }
  for (int s = 0; s < steps; s++) {
    double ucur = 0.0;
    if (NOISE) {
#pragma unroll
      for (int i = 0; i < MAXM; i++) if (i < M) nx[i] = ff_add_rn(x[i], ff_mul_rn(tau, gq[i]));
      ucur = uq;
      if (s + 1 < steps) {
        const double* gs = g + ((int64_t)(s + 1) * B + b) * M;
#pragma unroll
        for (int i = 0; i < MAXM; i++) if (i < M) gq[i] = gs[i];
        uq = u[(int64_t)(s + 1) * B + b];
      }
    } else {
#pragma unroll
      for (int q = 0; q < (MAXM + 3) / 4; q++)
        if (2 * q < n) {
          double z4[4];
          ff_normal_quad(seed, wid, (uint32_t)(s + 1), (uint32_t)q, z4);
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (4 * q + k < MAXM && 4 * q + k < M) nx[4 * q + k] = ff_add_rn(x[4 * q + k], ff_mul_rn(tau, z4[k]));
        }
    }
    double nl = ff_logprob_value<NU, ND>(nup, ndn, ou, od, nx, md);
    // p = exp(new_logp - logp) with torch's edge semantics: NaN stays NaN (rejects), -inf gives exactly 0
    const double dlp = nl - logp;
    double p = exp(0.0);
    if constexpr (FIXED) p = !(dlp == dlp) ? dlp : (dlp < -708.0 ? 0.0 : ff_exp(fmin(dlp, 708.0)));
    else p = exp(dlp);
    double uu = NOISE ? ucur : ff_uniform(seed, wid, (uint32_t)(s + 1), (uint32_t)n);
    bool acc = uu < p;  // NaN p -> reject, +inf p -> accept (IEEE), as torch
    if (acc) {
#pragma unroll
      for (int i = 0; i < MAXM; i++) if (i < M) x[i] = nx[i];
      logp = nl;
      nacc++;
    }
    if (accept && live) accept[(int64_t)s * B + b] = acc ? 1 : 0;
  }
  if (!live) return;
#pragma unroll
  for (int i = 0; i < MAXM; i++) if (i < M) x_out[b * M + i] = x[i];
  if (logp_out) logp_out[b] = logp;
  if (acc_count) acc_count[b] = nacc;
}

// ---------------------------------------------------------------------------------------------------
// Metropolis chain with TWO lanes per walker, one per spin species (NU = ND = NS): at 65 536 walkers one lane per
// walker is one wave per SIMD and nothing hides the latency of its serial chain (Philox rounds, Box-Muller, pivots).
// Lane (b, spin) draws the proposals of its own particles, evaluates its own determinant; the two log|det| meet through
// one DPP swap (a + b is commutative: both lanes form the identical sum, and it is the reference's 2*(up + down)), both
// lanes take the same accept decision.  Same noise stream, same results, bit for bit, as ff_mcmc_kernel.
template <int NS, bool NOISE>
__global__ void __launch_bounds__(128)
ff_mcmc_spin_kernel(int64_t B, const int* __restrict__ tab_up, const int* __restrict__ tab_dn, const int* __restrict__ wstate,
                    int steps, double tau, const double* __restrict__ g0, const double* __restrict__ g,
                    const double* __restrict__ u, uint64_t seed, int64_t woff, double* __restrict__ x_out,
                    double* __restrict__ logp_out, uint8_t* __restrict__ accept, int* __restrict__ acc_count) {
  constexpr int MS = 2 * NS, M = 2 * MS, n = 2 * NS;
  // quads (Philox blocks of four normals, coordinates 4q..4q+3 of the walker) that cover one spin's coordinates
  constexpr int NQ0 = (MS - 1) / 4 + 1, NQ1 = (2 * MS - 1) / 4 - MS / 4 + 1, NQ = NQ0 > NQ1 ? NQ0 : NQ1;
  __shared__ int s_md;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t b = gid >> 1;
  const int sp = (int)(gid & 1), off = sp * MS;
  if (threadIdx.x == 0) s_md = 0;
  __syncthreads();
  const bool live = b < B;
  if (!live) b = B - 1;
  const int st = wstate ? wstate[b] : 0;
  const int* __restrict__ tab = sp ? tab_dn : tab_up;
  int oo[2 * NS];
#pragma unroll
  for (int j = 0; j < NS; j++) ff_orb_decode(tab[st * NS + j], oo[j], oo[NS + j]);
  int md = 0;
#pragma unroll
  for (int j = 0; j < 2 * NS; j++) md = oo[j] > md ? oo[j] : md;
  atomicMax(&s_md, md);
  __syncthreads();
  md = FF_UNIFORM(s_md);

  double x[MS], nx[MS];
  const uint64_t wid = (uint64_t)(woff + b);
  const int q0 = off >> 2;
  auto draw = [&](uint32_t step, double* dst, bool propose) {   // this spin's normals of one step
    double Z[NQ][4];
#pragma unroll
    for (int qq = 0; qq < NQ; qq++) ff_normal_quad(seed, wid, step, (uint32_t)(q0 + qq), Z[qq]);
#pragma unroll
    for (int i = 0; i < MS; i++) {
      // coordinate off + i of the walker sits in quad (off + i) / 4; both spins' positions are compile-time constants
      const double za = Z[i >> 2][i & 3], zb = Z[((MS + i) >> 2) - (MS >> 2)][(MS + i) & 3];
      const double zv = (MS % 4 == 0) ? za : (sp ? zb : za);
      dst[i] = propose ? ff_add_rn(x[i], ff_mul_rn(tau, zv)) : zv;
    }
  };
  if (NOISE || g0 != nullptr) {   // explicit initial walkers (parity mode; ff_mcmc_continue)
#pragma unroll
    for (int i = 0; i < MS; i++) x[i] = g0[b * M + off + i];
  } else {
    draw(0u, x, false);
  }
  const double L0 = ff_slater_logabsdet_reg<NS>(oo, oo + NS, x, md);
  double logp = 2.0 * (L0 + ff_swap1(L0));
  int nacc = 0;
  double gq[MS], uq = 0.0;
  if (NOISE && steps > 0) {
#pragma unroll
    for (int i = 0; i < MS; i++) gq[i] = g[b * M + off + i];
    uq = u[b];
  }
  for (int s = 0; s < steps; s++) {
    double ucur = 0.0;
    if (NOISE) {
#pragma unroll
      for (int i = 0; i < MS; i++) nx[i] = ff_add_rn(x[i], ff_mul_rn(tau, gq[i]));
      ucur = uq;
      if (s + 1 < steps) {
        const double* gs = g + ((int64_t)(s + 1) * B + b) * M + off;
#pragma unroll
        for (int i = 0; i < MS; i++) gq[i] = gs[i];
        uq = u[(int64_t)(s + 1) * B + b];
      }
    } else {
      draw((uint32_t)(s + 1), nx, true);
    }
    const double L = ff_slater_logabsdet_reg<NS>(oo, oo + NS, nx, md);
    const double nl = 2.0 * (L + ff_swap1(L));
    const double dlp = nl - logp;
    const double p = !(dlp == dlp) ? dlp : (dlp < -708.0 ? 0.0 : ff_exp(fmin(dlp, 708.0)));
    const double uu = NOISE ? ucur : ff_uniform(seed, wid, (uint32_t)(s + 1), (uint32_t)n);
    const bool acc = uu < p;
    if (acc) {
#pragma unroll
      for (int i = 0; i < MS; i++) x[i] = nx[i];
      logp = nl;
      nacc++;
    }
    if (accept && live && sp == 0) accept[(int64_t)s * B + b] = acc ? 1 : 0;
  }
  if (!live) return;
#pragma unroll
  for (int i = 0; i < MS; i++) x_out[b * M + off + i] = x[i];
  if (sp == 0) {
    if (logp_out) logp_out[b] = logp;
    if (acc_count) acc_count[b] = nacc;
  }
}

// ---------------------------------------------------------------------------------------------------
// The Philox-fed chain of the two-lanes-per-walker layout above (throughput mode; the noise-fed template stays the reference's
// arithmetic, operation for operation).  Same stream, same walkers as ff_rng_fill + ff_mcmc_spin_kernel<NS, true>
// (test_mcmc_full_size_properties); what differs is how a step is evaluated (~925 wave-instructions per step):
//  * the decision u < |psi(x')|^2 / |psi(x)|^2 (src/base_dist.py:66-68) is taken as
//        u e^{R' - R} (P_up P_dn)^2 < (P'_up P'_dn)^2,   R = sum_i r_i^2,  P = det[h_nx_j(x_i) h_ny_j(y_i)]
//    with P the determinant of the POLYNOMIAL parts of the orbitals (the Gaussians of a row factor out of the determinant):
//    no exp per particle, no log per determinant, and the serial tail LU -> log -> exp -> compare of a step becomes
//    determinant -> multiply -> compare -- the left side is formed while the determinant is;
//  * determinants up to 3 x 3 in closed form (no pivot search, no reciprocal);
//  * the walker's Philox blocks are dealt over its two lanes: lane 0 evaluates quads 0 .. mid, lane 1 the quads behind and the
//    block of the uniform; for odd NS the second half of the middle quad and the uniform change lanes by one DPP swap
//    (two blocks per lane and step at NS = 3 where every lane evaluated three);
//  * step s + 1's normals are drawn while step s is decided (they depend on nothing of it).
// log|psi|^2 of the final walker is evaluated once at the end by the same routine as everywhere else.
template <int NS>
__global__ void __launch_bounds__(128)
ff_mcmc_spin_philox_kernel(int64_t B, const int* __restrict__ tab_up, const int* __restrict__ tab_dn, const int* __restrict__ wstate,
                           int steps, double tau, const double* __restrict__ g0, uint64_t seed, int64_t woff,
                           double* __restrict__ x_out, double* __restrict__ logp_out, int* __restrict__ acc_count) {
  constexpr int MS = 2 * NS, M = 2 * MS, n = 2 * NS;
  constexpr bool ODD = (NS & 1) != 0;
  constexpr int MID = (NS - 1) / 2;                          // odd NS: the quad whose halves belong to different spins
  constexpr int NB = ODD ? (NS + 1) / 2 : NS / 2 + 1;        // Philox blocks per lane and step
  __shared__ int s_md;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t b = gid >> 1;
  const int sp = (int)(gid & 1), off = sp * MS;
  if (threadIdx.x == 0) s_md = 0;
  __syncthreads();
  const bool live = b < B;
  if (!live) b = B - 1;
  const int st = wstate ? wstate[b] : 0;
  const int* __restrict__ tab = sp ? tab_dn : tab_up;
  int oo[2 * NS];
#pragma unroll
  for (int j = 0; j < NS; j++) ff_orb_decode(tab[st * NS + j], oo[j], oo[NS + j]);
  int md = 0;
#pragma unroll
  for (int j = 0; j < 2 * NS; j++) md = oo[j] > md ? oo[j] : md;
  atomicMax(&s_md, md);
  __syncthreads();
  md = FF_UNIFORM(s_md);

  const uint64_t wid = (uint64_t)(woff + b);
  auto swap32 = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); };
  // this spin's normals of one step (z[i]: coordinate off + i of the walker) and the walker's uniform of that step
  auto draw = [&](uint32_t step, double* z, double& u) {
    ff_u4 R[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) {
      uint32_t slot;
      if constexpr (ODD) slot = sp ? (uint32_t)(k < NB - 1 ? MID + 1 + k : n) : (uint32_t)k;
      else slot = (k < NB - 1) ? (uint32_t)(sp * (NS / 2) + k) : (uint32_t)n;
      R[k] = ff_philox(seed, wid, step, slot);
    }
    auto word = [&](int pair, int c) -> uint32_t {     // word c of pair `pair` of this lane's own blocks (static indices)
      const ff_u4& r = R[pair >> 1];
      return (pair & 1) ? (c ? r.w : r.z) : (c ? r.y : r.x);
    };
    uint32_t uh, ul;
    if constexpr (ODD) {
      const ff_u4& last = R[NB - 1];       // lane 0: the middle quad (its second half is lane 1's first pair); lane 1: the uniform's block
      const uint32_t r0 = swap32(sp ? last.x : last.z), r1 = swap32(sp ? last.y : last.w);
#pragma unroll
      for (int j = 0; j < NS; j++) {
        const uint32_t a0 = word(j, 0), a1 = word(j, 1);
        const uint32_t b0 = j == 0 ? r0 : word(j > 0 ? j - 1 : 0, 0), b1 = j == 0 ? r1 : word(j > 0 ? j - 1 : 0, 1);
        ff_normal_pair32(sp ? b0 : a0, sp ? b1 : a1, z[2 * j], z[2 * j + 1]);
      }
      uh = sp ? last.x : r0; ul = sp ? last.y : r1;
    } else {
#pragma unroll
      for (int j = 0; j < NS; j++) ff_normal_pair32(word(j, 0), word(j, 1), z[2 * j], z[2 * j + 1]);
      uh = R[NB - 1].x; ul = R[NB - 1].y;
    }
    u = ff_uniform_words(uh, ul);
  };

  double x[MS], zq[MS], uq = 0.0;
  if (g0 != nullptr) {              // ff_mcmc_continue: explicit initial walkers
#pragma unroll
    for (int i = 0; i < MS; i++) x[i] = g0[b * M + off + i];
  } else {
    draw(0u, x, uq);
  }
  if (steps > 0) draw(1u, zq, uq);
  double Rc = 0.0;
#pragma unroll
  for (int i = 0; i < MS; i++) Rc = fma(x[i], x[i], Rc);
  double PP2;
  {
    const double P = ff_slater_polydet_reg<NS>(oo, oo + NS, x, md), PP = P * ff_swap1(P);
    PP2 = PP * PP;
  }
  int nacc = 0;
  for (int s = 0; s < steps; s++) {
    double nx[MS];
#pragma unroll
    for (int i = 0; i < MS; i++) nx[i] = ff_add_rn(x[i], ff_mul_rn(tau, zq[i]));     // two roundings, as the noise-fed kernel
    const double u = uq;
    if (s + 1 < steps) draw((uint32_t)(s + 2), zq, uq);
    double Rn = 0.0;
#pragma unroll
    for (int i = 0; i < MS; i++) Rn = fma(nx[i], nx[i], Rn);
    const double dR = Rc - Rn, dRt = dR + ff_swap1(dR);
    const double lhs = u * ff_exp(fmin(fmax(-dRt, -700.0), 708.0)) * PP2;
    const double P = ff_slater_polydet_reg<NS>(oo, oo + NS, nx, md), PPn = P * ff_swap1(P), PPn2 = PPn * PPn;
    // IEEE comparisons: a NaN on either side rejects; e^{R - R'} below the double range is the reference's p = 0 (reject)
    const bool acc = (dRt >= -708.0) & (lhs < PPn2);
    if (acc) {
#pragma unroll
      for (int i = 0; i < MS; i++) x[i] = nx[i];
      Rc = Rn;
      PP2 = PPn2;
      nacc++;
    }
  }
  const double L0 = ff_slater_logabsdet_reg<NS>(oo, oo + NS, x, md);
  const double logp = 2.0 * (L0 + ff_swap1(L0));
  if (!live) return;
#pragma unroll
  for (int i = 0; i < MS; i++) x_out[b * M + off + i] = x[i];
  if (sp == 0) {
    if (logp_out) logp_out[b] = logp;
    if (acc_count) acc_count[b] = nacc;
  }
}

// ---------------------------------------------------------------------------------------------------
// Metropolis chain of ONE spin species (ndown = 0: the finite-temperature runs, src/BetaFermionHO2D.py) with TWO lanes per
// walker.  One lane per walker is one wave per SIMD at 65 536 walkers and nothing hides its serial chain; there is no second
// species to split off here, so the lanes split the PARTICLES: lane h of a walker owns the Philox quads q = h, h + 2, ...
// (quad q = the four normals of particles 2q, 2q + 1), proposes those particles' moves and evaluates their rows of the Slater
// matrix; the rows change hands through one DPP swap per entry, then both lanes run the same LU on the same matrix -- the
// same determinant bits, hence the same accept decision, as ff_mcmc_kernel.  Same noise stream, same results, bit for bit.
template <int NS, bool NOISE>
__global__ void __launch_bounds__(128)
ff_mcmc_pair_kernel(int64_t B, const int* __restrict__ tab_up, const int* __restrict__ wstate, int steps, double tau,
                    const double* __restrict__ g0, const double* __restrict__ g, const double* __restrict__ u, uint64_t seed,
                    int64_t woff, double* __restrict__ x_out, double* __restrict__ logp_out, uint8_t* __restrict__ accept,
                    int* __restrict__ acc_count) {
  constexpr int M = 2 * NS, NQT = (NS + 1) / 2, NQL = (NQT + 1) / 2, NPL = 2 * NQL, MPL = 2 * NPL;   // quads total / per lane, slots
  __shared__ int s_md;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t b = gid >> 1;
  const int h = (int)(gid & 1);
  if (threadIdx.x == 0) s_md = 0;
  __syncthreads();
  const bool live = b < B;
  if (!live) b = B - 1;
  const int st = wstate ? wstate[b] : 0;
  int oo[2 * NS];
#pragma unroll
  for (int j = 0; j < NS; j++) ff_orb_decode(tab_up[st * NS + j], oo[j], oo[NS + j]);
  int md = 0;
#pragma unroll
  for (int j = 0; j < 2 * NS; j++) md = oo[j] > md ? oo[j] : md;
  atomicMax(&s_md, md);
  __syncthreads();
  md = FF_UNIFORM(s_md);

  // slot s of this lane: particle pid(s) = 4 (s / 2) + 2 h + (s & 1)  (quad q = 2 (s / 2) + h); it exists if pid < NS
  double xm[MPL], nxm[MPL];
#pragma unroll
  for (int i = 0; i < MPL; i++) xm[i] = 0.0;
  const uint64_t wid = (uint64_t)(woff + b);
  auto pid = [&](int sl) -> int { return 4 * (sl >> 1) + 2 * h + (sl & 1); };
  auto draw = [&](uint32_t step, double* dst, bool propose) {
#pragma unroll
    for (int k = 0; k < NQL; k++) {
      double z4[4];
      ff_normal_quad(seed, wid, step, (uint32_t)(2 * k + h), z4);
#pragma unroll
      for (int c = 0; c < 4; c++) dst[4 * k + c] = propose ? ff_add_rn(xm[4 * k + c], ff_mul_rn(tau, z4[c])) : z4[c];
    }
  };
  auto load = [&](const double* src, double* dst) {     // this lane's coordinates of a (B, NS, 2) array row
#pragma unroll
    for (int sl = 0; sl < NPL; sl++) {
      const int p = pid(sl) < NS ? pid(sl) : 0;
      dst[2 * sl] = src[2 * p]; dst[2 * sl + 1] = src[2 * p + 1];
    }
  };
  // log|psi|^2 of the walker whose coordinates (this lane's share) are in `xs`
  auto logprob = [&](const double* xs) -> double {
    double Dm[NPL][NS];
#pragma unroll
    for (int sl = 0; sl < NPL; sl++) ff_slater_row_reg<NS>(oo, oo + NS, xs[2 * sl], xs[2 * sl + 1], md, Dm[sl]);
    double Dfull[NS][NS];
#pragma unroll
    for (int i = 0; i < NS; i++) {
      const int q = i >> 1, owner = q & 1, sl = 2 * (q >> 1) + (i & 1);   // static: which lane and slot hold row i
#pragma unroll
      for (int j = 0; j < NS; j++) {
        const double mine = Dm[sl][j], other = ff_swap1(mine);
        Dfull[i][j] = (h == owner) ? mine : other;
      }
    }
    return 2.0 * (0.0 + ff_lu_logabsdet_reg<NS>(Dfull));
  };
  // determinant (up to its sign) of the polynomial parts of the orbitals, same exchange of rows
  auto polydet = [&](const double* xs) -> double {
    double Dm[NPL][NS];
#pragma unroll
    for (int sl = 0; sl < NPL; sl++) ff_poly_row_reg<NS>(oo, oo + NS, xs[2 * sl], xs[2 * sl + 1], md, Dm[sl]);
    double Dfull[NS][NS];
#pragma unroll
    for (int i = 0; i < NS; i++) {
      const int q = i >> 1, owner = q & 1, sl = 2 * (q >> 1) + (i & 1);
#pragma unroll
      for (int j = 0; j < NS; j++) {
        const double mine = Dm[sl][j], other = ff_swap1(mine);
        Dfull[i][j] = (h == owner) ? mine : other;
      }
    }
    return ff_det_reg<NS>(Dfull);
  };

  if (NOISE || g0 != nullptr) load(g0 + b * M, xm);
  else draw(0u, xm, false);
  int nacc = 0;
  double logp;
  if constexpr (NOISE) {
    logp = logprob(xm);
    double gq[MPL], uq = 0.0;
    if (steps > 0) { load(g + b * M, gq); uq = u[b]; }
    for (int s = 0; s < steps; s++) {
      double ucur = 0.0;
#pragma unroll
      for (int i = 0; i < MPL; i++) nxm[i] = ff_add_rn(xm[i], ff_mul_rn(tau, gq[i]));
      ucur = uq;
      if (s + 1 < steps) { load(g + ((int64_t)(s + 1) * B + b) * M, gq); uq = u[(int64_t)(s + 1) * B + b]; }
      const double nl = logprob(nxm);
      const double dlp = nl - logp;
      const double p = !(dlp == dlp) ? dlp : (dlp < -708.0 ? 0.0 : ff_exp(fmin(dlp, 708.0)));
      const double uu = ucur;
      const bool acc = uu < p;
      if (acc) {
#pragma unroll
        for (int i = 0; i < MPL; i++) xm[i] = nxm[i];
        logp = nl;
        nacc++;
      }
      if (accept && live && h == 0) accept[(int64_t)s * B + b] = acc ? 1 : 0;
    }
  } else {
    // Philox-fed chain: the decision as u e^{R' - R} P^2 < P'^2 on the determinant of the polynomial parts (see
    // ff_mcmc_spin_philox_kernel); the walkers are those of the branch above on the materialised stream
    auto r2sum = [&](const double* xs) -> double {       // this lane's share of sum_i r_i^2 (absent particles: none)
      double r = 0.0;
#pragma unroll
      for (int sl = 0; sl < NPL; sl++) { const double t = fma(xs[2 * sl], xs[2 * sl], xs[2 * sl + 1] * xs[2 * sl + 1]); r += (pid(sl) < NS) ? t : 0.0; }
      return r;
    };
    double Rc = r2sum(xm);
    double P2 = polydet(xm);
    P2 *= P2;
    double uq = 0.0;
    if (steps > 0) { draw(1u, nxm, false); uq = ff_uniform(seed, wid, 1u, (uint32_t)NS); }
    for (int s = 0; s < steps; s++) {
      double cur[MPL];
#pragma unroll
      for (int i = 0; i < MPL; i++) cur[i] = ff_add_rn(xm[i], ff_mul_rn(tau, nxm[i]));
      const double uu = uq;
      // step s + 1's normals and uniform: they depend on nothing of this step
      if (s + 1 < steps) { draw((uint32_t)(s + 2), nxm, false); uq = ff_uniform(seed, wid, (uint32_t)(s + 2), (uint32_t)NS); }
      const double Rn = r2sum(cur), dR = Rc - Rn, dRt = dR + ff_swap1(dR);
      const double lhs = uu * ff_exp(fmin(fmax(-dRt, -700.0), 708.0)) * P2;
      double Pn2 = polydet(cur);
      Pn2 *= Pn2;
      const bool acc = (dRt >= -708.0) & (lhs < Pn2);
      if (acc) {
#pragma unroll
        for (int i = 0; i < MPL; i++) xm[i] = cur[i];
        Rc = Rn;
        P2 = Pn2;
        nacc++;
      }
    }
    logp = logprob(xm);
  }
  if (!live) return;
#pragma unroll
  for (int sl = 0; sl < NPL; sl++) {
    if (pid(sl) < NS) { x_out[b * M + 2 * pid(sl)] = xm[2 * sl]; x_out[b * M + 2 * pid(sl) + 1] = xm[2 * sl + 1]; }
  }
  if (h == 0) {
    if (logp_out) logp_out[b] = logp;
    if (acc_count) acc_count[b] = nacc;
  }
}

// materialise the Philox noise stream of ff_mcmc_kernel<.., false>
__global__ void __launch_bounds__(128)
ff_rng_fill_kernel(int64_t B, int n, int steps, uint64_t seed, int64_t woff, double* __restrict__ g0,
                   double* __restrict__ g, double* __restrict__ u) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int M = 2 * n;
  const uint64_t wid = (uint64_t)(woff + b);
  for (int q = 0; 2 * q < n; q++) {
    double z4[4];
    ff_normal_quad(seed, wid, 0u, (uint32_t)q, z4);
    for (int k = 0; k < 4; k++) if (4 * q + k < M) g0[b * M + 4 * q + k] = z4[k];
  }
  for (int s = 0; s < steps; s++) {
    double* gs = g + ((int64_t)s * B + b) * M;
    for (int q = 0; 2 * q < n; q++) {
      double z4[4];
      ff_normal_quad(seed, wid, (uint32_t)(s + 1), (uint32_t)q, z4);
      for (int k = 0; k < 4; k++) if (4 * q + k < M) gs[4 * q + k] = z4[k];
    }
    u[(int64_t)s * B + b] = ff_uniform(seed, wid, (uint32_t)(s + 1), (uint32_t)n);
  }
}

// ---------------------------------------------------------------------------------------------------
// LogAbsSlaterDet forward / backward (src/slater.py:13-62), one lane per walker
__global__ void __launch_bounds__(128)
ff_slater_fwd_kernel(int64_t B, int n, const int* __restrict__ tab, const int* __restrict__ wstate,
                     const double* __restrict__ x, double* __restrict__ lad) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double xl[2 * FF_MAX_NS];
  for (int i = 0; i < 2 * n; i++) xl[i] = x[b * 2 * n + i];
  lad[b] = ff_slater_general(n, tab + (wstate ? wstate[b] : 0) * n, xl, nullptr, nullptr);
}

__global__ void __launch_bounds__(128)
ff_slater_bwd_kernel(int64_t B, int n, const int* __restrict__ tab, const int* __restrict__ wstate,
                     const double* __restrict__ x, const double* __restrict__ gout, double* __restrict__ gx) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double xl[2 * FF_MAX_NS], T[2 * FF_MAX_NS * FF_MAX_NS];
  for (int i = 0; i < 2 * n; i++) xl[i] = x[b * 2 * n + i];
  ff_slater_general(n, tab + (wstate ? wstate[b] : 0) * n, xl, T, nullptr);
  double go = gout[b];
  for (int a = 0; a < n; a++) {
    gx[b * 2 * n + 2 * a] = go * T[a * n + a];
    gx[b * 2 * n + 2 * a + 1] = go * T[n * n + a * n + a];
  }
}

// FreeFermion.log_prob with gradient and Laplacian (what y_grad_laplacian extracts, src/utils.py:40-65)
__global__ void __launch_bounds__(128)
ff_logprob_kernel(int64_t B, int nup, int ndn, const int* __restrict__ tab_up, const int* __restrict__ tab_dn,
                  const int* __restrict__ wstate, const double* __restrict__ x, double* __restrict__ logp,
                  double* __restrict__ grad, double* __restrict__ lap) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int n = nup + ndn, st = wstate ? wstate[b] : 0;
  const bool deriv = grad || lap;
  double xl[2 * FF_MAX_NS], T[2 * FF_MAX_NS * FF_MAX_NS], S[3 * FF_MAX_NS];
  double lp = 0.0, lpl = 0.0;
  for (int sp = 0; sp < 2; sp++) {
    const int ns = sp ? ndn : nup, off = sp ? nup : 0;
    if (!ns) continue;
    const int* orb = (sp ? tab_dn : tab_up) + st * ns;
    for (int i = 0; i < 2 * ns; i++) xl[i] = x[b * 2 * n + 2 * off + i];
    lp += ff_slater_general(ns, orb, xl, deriv ? T : nullptr, deriv ? S : nullptr);
    if (deriv)
      for (int a = 0; a < ns; a++) {
        double gxa = T[a * ns + a], gya = T[ns * ns + a * ns + a];
        if (grad) { grad[b * 2 * n + 2 * (off + a)] = 2.0 * gxa; grad[b * 2 * n + 2 * (off + a) + 1] = 2.0 * gya; }
        lpl += S[3 * a] + S[3 * a + 2] - gxa * gxa - gya * gya;
      }
  }
  logp[b] = 2.0 * lp;
  if (lap) lap[b] = 2.0 * lpl;
}

// Register-resident instantiations of the three kernels above for one compile-time determinant size NS (both spin
// species the same size, or one species): same arithmetic as the generic routine (ff_slater_fixed), no private arrays.
template <int NS>
__global__ void __launch_bounds__(128)
ff_slater_fwd_fixed_kernel(int64_t B, const int* __restrict__ tab, const int* __restrict__ wstate,
                           const double* __restrict__ x, double* __restrict__ lad) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double xl[2 * NS];
#pragma unroll
  for (int i = 0; i < 2 * NS; i++) xl[i] = x[b * 2 * NS + i];
  lad[b] = ff_slater_fixed<NS, false>(tab + (wstate ? wstate[b] : 0) * NS, xl, nullptr, nullptr);
}

template <int NS>
__global__ void __launch_bounds__(128)
ff_slater_bwd_fixed_kernel(int64_t B, const int* __restrict__ tab, const int* __restrict__ wstate,
                           const double* __restrict__ x, const double* __restrict__ gout, double* __restrict__ gx) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double xl[2 * NS], T[2 * NS * NS];
#pragma unroll
  for (int i = 0; i < 2 * NS; i++) xl[i] = x[b * 2 * NS + i];
  ff_slater_fixed<NS, true>(tab + (wstate ? wstate[b] : 0) * NS, xl, T, nullptr);
  const double go = gout[b];
#pragma unroll
  for (int a = 0; a < NS; a++) {
    gx[b * 2 * NS + 2 * a] = go * T[a * NS + a];
    gx[b * 2 * NS + 2 * a + 1] = go * T[NS * NS + a * NS + a];
  }
}

template <int NS, bool DERIV>
__global__ void __launch_bounds__(128)
ff_logprob_fixed_kernel(int64_t B, int nup, int ndn, const int* __restrict__ tab_up, const int* __restrict__ tab_dn,
                        const int* __restrict__ wstate, const double* __restrict__ x, double* __restrict__ logp,
                        double* __restrict__ grad, double* __restrict__ lap) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int n = nup + ndn, st = wstate ? wstate[b] : 0;
  double lp = 0.0, lpl = 0.0;
#pragma unroll 1
  for (int sp = 0; sp < 2; sp++) {
    const int ns = sp ? ndn : nup, off = sp ? nup : 0;
    if (!ns) continue;                       // ns == NS otherwise
    double xl[2 * NS], T[DERIV ? 2 * NS * NS : 1], S[DERIV ? 3 * NS : 1];
#pragma unroll
    for (int i = 0; i < 2 * NS; i++) xl[i] = x[b * 2 * n + 2 * off + i];
    lp += ff_slater_fixed<NS, DERIV>((sp ? tab_dn : tab_up) + st * NS, xl, T, S);
    if constexpr (DERIV) {
#pragma unroll
      for (int a = 0; a < NS; a++) {
        const double gxa = T[a * NS + a], gya = T[NS * NS + a * NS + a];
        if (grad) { grad[b * 2 * n + 2 * (off + a)] = 2.0 * gxa; grad[b * 2 * n + 2 * (off + a) + 1] = 2.0 * gya; }
        lpl += S[3 * a] + S[3 * a + 2] - gxa * gxa - gya * gya;
      }
    }
  }
  logp[b] = 2.0 * lp;
  if (lap) lap[b] = 2.0 * lpl;
}

// ---------------------------------------------------------------------------------------------------
// HO.V + CoulombPairPotential.V (src/potentials.py:13,23-47)
__global__ void __launch_bounds__(128)
ff_potential_kernel(int64_t B, int n, int d, double Z, int use_ho, const double* __restrict__ x, double* __restrict__ V) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double xl[3 * FF_MAX_N];
  const int M = n * d;
  for (int i = 0; i < M; i++) xl[i] = x[b * M + i];
  double pair = 0.0, ho = 0.0;
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++) {
      double r2 = 0.0;
      for (int c = 0; c < d; c++) { double t = xl[i * d + c] - xl[j * d + c]; r2 = fma(t, t, r2); }
      pair += Z / sqrt(r2);
    }
  for (int i = 0; i < M; i++) ho = fma(xl[i], xl[i], ho);
  V[b] = pair + (use_ho ? 0.5 * ho : 0.0);
}

// The same for compile-time (N, D): HBM-streaming version.  One wave takes 64 consecutive walkers: their 64*N*D
// coordinates are one contiguous span, read with fully coalesced 8-byte loads into LDS (row stride N*D+1: conflict-free
// when every lane then reads its own walker), coordinates in registers, 1/r from v_rsq_f64 + two Newton steps instead
// of sqrt and a division (15 pairs at n = 6: ~200 instructions per walker, far below the time the 104 bytes take).
template <int N, int D>
__global__ void __launch_bounds__(FF_WAVE)
ff_potential_stream_kernel(int64_t B, double Z, int use_ho, const double* __restrict__ x, double* __restrict__ V) {
  constexpr int M = N * D;
  __shared__ double s_x[FF_WAVE * (M + 1)];
  const int lane = threadIdx.x;
  const int64_t ntiles = (B + FF_WAVE - 1) / FF_WAVE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t b0 = tile * FF_WAVE;
    const int nw = (int)((B - b0) < FF_WAVE ? (B - b0) : FF_WAVE);
    __syncthreads();
    if constexpr (M % 2 == 0) {                     // 16-byte loads (a pair never straddles two walkers)
      const double2* __restrict__ x2 = reinterpret_cast<const double2*>(x + b0 * M);
#pragma unroll
      for (int k = 0; k < M / 2; k++) {
        const int e2 = k * FF_WAVE + lane;         // pair of the tile's span
        if (2 * e2 < nw * M) {
          const double2 v = x2[e2];
          double* dst = &s_x[((2 * e2) / M) * (M + 1) + (2 * e2) % M];
          dst[0] = v.x; dst[1] = v.y;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < M; k++) {
        const int e = k * FF_WAVE + lane;          // element of the tile's span
        if (e < nw * M) s_x[(e / M) * (M + 1) + e % M] = x[b0 * M + e];
      }
    }
    __syncthreads();
    if (lane < nw) {
      double xl[M];
#pragma unroll
      for (int i = 0; i < M; i++) xl[i] = s_x[lane * (M + 1) + i];
      double pair = 0.0, ho = 0.0;
#pragma unroll
      for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = i + 1; j < N; j++) {
          double r2 = 0.0;
#pragma unroll
          for (int c = 0; c < D; c++) { const double t = xl[i * D + c] - xl[j * D + c]; r2 = fma(t, t, r2); }
          double r, ri;
          ff_sqrt_rcp(r2, r, ri);
          pair += ri;
        }
#pragma unroll
      for (int i = 0; i < M; i++) ho = fma(xl[i], xl[i], ho);
      V[b0 + lane] = fma(Z, pair, use_ho ? 0.5 * ho : 0.0);
    }
  }
}

// MLP.forward / MLP.grad (src/MLP.py:30-45) on a flat list of scalars
FF_D void ff_mlp_point(int H, const double* __restrict__ w1, const double* __restrict__ b1, const double* __restrict__ w2,
                       double r, double& val, double& dval) {
  double s = 0.0, g = 0.0;
  for (int h = 0; h < H; h++) {
    double a = ff_sigmoid(fma(w1[h], r, b1[h]));
    s = fma(w2[h], a, s);
    g = fma(w2[h] * w1[h], a * (1.0 - a), g);
  }
  val = s; dval = g;
}

__global__ void __launch_bounds__(128)
ff_mlp_kernel(int64_t N, int H, const double* __restrict__ w1, const double* __restrict__ b1, const double* __restrict__ w2,
              const double* __restrict__ r, double* __restrict__ val, double* __restrict__ dval) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double v, g;
  ff_mlp_point(H, w1, b1, w2, r[i], v, g);
  val[i] = v;
  if (dval) dval[i] = g;
}

// value and first two derivatives of the scalar MLP at r
FF_D void ff_mlp_point2(int H, const double* __restrict__ w1, const double* __restrict__ b1, const double* __restrict__ w2,
                        double r, double& val, double& d1, double& d2) {
  double s = 0.0, g = 0.0, q = 0.0;
  for (int h = 0; h < H; h++) {
    const double a = ff_sigmoid(fma(w1[h], r, b1[h])), sp = a * (1.0 - a), ww = w2[h] * w1[h];
    s = fma(w2[h], a, s);
    g = fma(ww, sp, g);
    q = fma(ww * w1[h], sp * (1.0 - 2.0 * a), q);
  }
  val = s; d1 = g; d2 = q;
}

// MLP.forward / MLP.grad for any input dimension (src/MLP.py:30-45; the backflow potentials use D_in = 1: ff_mlp_kernel):
// val[i] = fc2(sigmoid(fc1(x_i))), grad[i][:] = (fc2.weight * sigmoid') @ fc1.weight.  One lane per point.
#define FF_MLP_DMAX 64
__global__ void __launch_bounds__(128)
ff_mlp_nd_kernel(int64_t N, int Din, int H, const double* __restrict__ w1, const double* __restrict__ b1, const double* __restrict__ w2,
                 const double* __restrict__ x, double* __restrict__ val, double* __restrict__ grad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double xl[FF_MLP_DMAX], gl[FF_MLP_DMAX];
  for (int k = 0; k < Din; k++) { xl[k] = x[i * Din + k]; gl[k] = 0.0; }
  double s = 0.0;
  for (int h = 0; h < H; h++) {
    double a = b1[h];
    for (int k = 0; k < Din; k++) a = fma(w1[(int64_t)h * Din + k], xl[k], a);
    const double sg = ff_sigmoid(a), c = w2[h] * sg * (1.0 - sg);
    s = fma(w2[h], sg, s);
    if (grad) for (int k = 0; k < Din; k++) gl[k] = fma(c, w1[(int64_t)h * Din + k], gl[k]);
  }
  if (val) val[i] = s;
  if (grad) for (int k = 0; k < Din; k++) grad[i * Din + k] = gl[k];
}

// Vector-Jacobian products of the backflow field for autograd through Backflow.forward / .divergence
// (tests/test_equivariant_funs.py:25-35 of the reference differentiates v with respect to x):
//   Aw   = (dv/dx)^T w = (dv/dx) w      (dv/dx is symmetric: blocks B = eta I + (eta'/r) rho rho^T of every pair, mu likewise)
//   gdiv = grad_x div v                 (per pair 2 (eta'' r + (1 + D) eta') rho / r; per particle (mu'' r + (1 + D) mu') x / r)
// One lane per walker, direct sigmoids, any n <= FF_MAX_N, d <= 3.  w / Aw or gdiv may be NULL.
__global__ void __launch_bounds__(128)
ff_backflow_vjp_kernel(int64_t B, int n, int d, ff_net net, const double* __restrict__ x, const double* __restrict__ w,
                       double* __restrict__ Aw, double* __restrict__ gdiv) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int M = n * d;
  double xl[3 * FF_MAX_N], wl[3 * FF_MAX_N], al[3 * FF_MAX_N], gl[3 * FF_MAX_N];
  for (int i = 0; i < M; i++) { xl[i] = x[b * M + i]; wl[i] = w ? w[b * M + i] : 0.0; al[i] = 0.0; gl[i] = 0.0; }
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++) {
      double rho[3], dw[3], r2 = 0.0, rdw = 0.0;
      for (int c = 0; c < d; c++) { rho[c] = xl[i * d + c] - xl[j * d + c]; dw[c] = wl[i * d + c] - wl[j * d + c]; r2 = fma(rho[c], rho[c], r2); rdw = fma(rho[c], dw[c], rdw); }
      const double r = sqrt(r2);
      double f0, f1, f2;
      ff_mlp_point2(net.He, net.ew1, net.eb1, net.ew2, r, f0, f1, f2);
      const double k1 = f1 * rdw / r, gq = 2.0 * fma(f2, r, (1.0 + d) * f1) / r;
      for (int c = 0; c < d; c++) {
        const double t = fma(k1, rho[c], f0 * dw[c]);
        al[i * d + c] += t; al[j * d + c] -= t;
        gl[i * d + c] = fma(gq, rho[c], gl[i * d + c]); gl[j * d + c] = fma(-gq, rho[c], gl[j * d + c]);
      }
    }
  if (net.Hm > 0)
    for (int i = 0; i < n; i++) {
      double r2 = 0.0, rw = 0.0;
      for (int c = 0; c < d; c++) { r2 = fma(xl[i * d + c], xl[i * d + c], r2); rw = fma(xl[i * d + c], wl[i * d + c], rw); }
      const double r = sqrt(r2);
      double f0, f1, f2;
      ff_mlp_point2(net.Hm, net.mw1, net.mb1, net.mw2, r, f0, f1, f2);
      const double k1 = f1 * rw / r, gq = fma(f2, r, (1.0 + d) * f1) / r;
      for (int c = 0; c < d; c++) {
        al[i * d + c] += fma(k1, xl[i * d + c], f0 * wl[i * d + c]);
        gl[i * d + c] = fma(gq, xl[i * d + c], gl[i * d + c]);
      }
    }
  if (Aw) for (int i = 0; i < M; i++) Aw[b * M + i] = al[i];
  if (gdiv) for (int i = 0; i < M; i++) gdiv[b * M + i] = gl[i];
}

// Backflow.forward / .divergence (src/equivariant_funs.py:83-102), any n <= FF_MAX_N, d <= 3, any H.
// (The j == i term of the reference's "+eye" formulation cancels analytically and is skipped.)
__global__ void __launch_bounds__(128)
ff_backflow_kernel(int64_t B, int n, int d, ff_net net, const double* __restrict__ x, double* __restrict__ v,
                   double* __restrict__ div) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int M = n * d;
  double xl[3 * FF_MAX_N], vl[3 * FF_MAX_N];
  for (int i = 0; i < M; i++) { xl[i] = x[b * M + i]; vl[i] = 0.0; }
  double dv = 0.0;
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++) {
      double rho[3], r2 = 0.0;
      for (int c = 0; c < d; c++) { rho[c] = xl[i * d + c] - xl[j * d + c]; r2 = fma(rho[c], rho[c], r2); }
      double r = sqrt(r2), eta, deta;
      ff_mlp_point(net.He, net.ew1, net.eb1, net.ew2, r, eta, deta);
      for (int c = 0; c < d; c++) { vl[i * d + c] = fma(eta, rho[c], vl[i * d + c]); vl[j * d + c] = fma(-eta, rho[c], vl[j * d + c]); }
      dv += 2.0 * fma(deta, r, d * eta);
    }
  if (net.Hm > 0)
    for (int i = 0; i < n; i++) {
      double r2 = 0.0;
      for (int c = 0; c < d; c++) r2 = fma(xl[i * d + c], xl[i * d + c], r2);
      double r = sqrt(r2), mu, dmu;
      ff_mlp_point(net.Hm, net.mw1, net.mb1, net.mw2, r, mu, dmu);
      for (int c = 0; c < d; c++) vl[i * d + c] = fma(mu, xl[i * d + c], vl[i * d + c]);
      dv += fma(dmu, r, d * mu);
    }
  if (v) for (int i = 0; i < M; i++) v[b * M + i] = vl[i];
  if (div) div[b] = dv;
}

// The reduction shift is a numerical convenience (sums about the previous sweep's mean have no cancellation): a non-finite
// one -- a failed integration yields NaN local energies by design, hence a NaN mean -- counts as 0, so that one bad sweep
// cannot poison every later one (every kernel that reads a shift goes through here).
FF_D double ff_finite_shift(double c) { return (c - c == 0.0) ? c : 0.0; }

// ---------------------------------------------------------------------------------------------------
// out[0] = sum (e - shift), out[1] = sum (e - shift)^2; one workgroup, fixed summation tree (deterministic for a given
// block size); four independent accumulator pairs per thread keep several loads in flight
__global__ void __launch_bounds__(1024)
ff_moments_kernel(int64_t B, const double* __restrict__ e, double shift_host, const double* __restrict__ shift_dev,
                  double shift_dev_scale, double* __restrict__ out) {
  __shared__ double s1[1024], s2[1024];
  const double shift = ff_finite_shift(shift_dev ? shift_dev[0] * shift_dev_scale : shift_host);
  const int nt = blockDim.x, t = threadIdx.x;
  double a[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
  int64_t i = t;
  for (; i + 3 * (int64_t)nt < B; i += 4 * (int64_t)nt) {
#pragma unroll
    for (int k = 0; k < 4; k++) { const double v = e[i + k * (int64_t)nt] - shift; a[k] += v; q[k] = fma(v, v, q[k]); }
  }
  for (int k = 0; i < B; i += nt, k++) { const double v = e[i] - shift; a[k] += v; q[k] = fma(v, v, q[k]); }
  s1[t] = (a[0] + a[1]) + (a[2] + a[3]);
  s2[t] = (q[0] + q[1]) + (q[2] + q[3]);
  __syncthreads();
  int w = 1;
  while (w * 2 < nt) w *= 2;   // largest power of two below the block size
  for (; w > 0; w >>= 1) {
    if (t < w && t + w < nt) { s1[t] += s1[t + w]; s2[t] += s2[t + w]; }
    __syncthreads();
  }
  if (t == 0) { out[0] = s1[0]; out[1] = s2[0]; }
}

// Adam for a handful of small tensors in ONE launch (ff_adam_step; src/FermionHO2D.py:61 builds torch.optim.Adam over the six tensors
// of the two MLPs: 300 numbers).  PyTorch's fused implementation takes two launches of its multi-tensor machinery, 24 us each on this
// GPU -- 3 % of a 1.5 ms iteration for 300 numbers.  The update is torch.optim.Adam's single-tensor formula, operation for operation
// (torch/optim/adam.py, _single_tensor_adam: lerp for the first moment, sqrt(v) / sqrt(1 - beta2^t) + eps, step size lr / (1 - beta1^t);
// weight decay as the L2 term it is there; amsgrad / maximize are not offered).  Workgroup = tensor.
#define FF_ADAM_MAXT 16
struct ff_adam_args {
  int64_t size[FF_ADAM_MAXT];
  double* p[FF_ADAM_MAXT];
  const double* g[FF_ADAM_MAXT];
  double* m[FF_ADAM_MAXT];
  double* v[FF_ADAM_MAXT];
  double lr, beta1, beta2, eps, wd, bc1, sqrt_bc2;
};
__global__ void __launch_bounds__(256) ff_adam_kernel(ff_adam_args A) {
  const int t = blockIdx.x;
  double* __restrict__ p = A.p[t];
  const double* __restrict__ g = A.g[t];
  double* __restrict__ m = A.m[t];
  double* __restrict__ v = A.v[t];
  const double step_size = A.lr / A.bc1;
  for (int64_t i = threadIdx.x; i < A.size[t]; i += blockDim.x) {
    double gi = g[i];
    const double pi = p[i];
    if (A.wd != 0.0) gi = fma(A.wd, pi, gi);
    const double mi = fma(gi - m[i], 1.0 - A.beta1, m[i]);             // exp_avg.lerp_(grad, 1 - beta1)
    const double vi = fma(A.beta2, v[i], (1.0 - A.beta2) * gi * gi);   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    m[i] = mi; v[i] = vi;
    const double denom = sqrt(vi) / A.sqrt_bc2 + A.eps;
    p[i] = pi - step_size * (mi / denom);                               // param.addcdiv_(exp_avg, denom, value = -step_size)
  }
}

// One idle wave that returns after `ticks` of the 100 MHz constant clock: holds a side stream back for a few microseconds
// so that the kernel the main stream launches at the same moment gets its waves placed first (ff_stream_delay)
__global__ void ff_delay_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// Ground-state estimator sums in one pass (src/VMC.py:56-59): with c = shift[0] (any number every rank agrees on, e.g. the
// previous sweep's mean: no cancellation)  out = [sum (e - c), sum (e - c)^2, sum logp, sum logp (e - c)].  One workgroup,
// fixed tree: deterministic.  The sums of all ranks add up; ff_energy_finish turns the totals into E, the centred sum of
// squares and the surrogate mean(logp (e - E)).
__global__ void __launch_bounds__(1024)
ff_energy_sums_kernel(int64_t B, const double* __restrict__ e, const double* __restrict__ logp, const double* __restrict__ shift_dev,
                      double* __restrict__ out) {
  __shared__ double sm[4][1024];
  const double shift = ff_finite_shift(shift_dev[0]);
  const int nt = blockDim.x, t = threadIdx.x;
  double a[2] = {0.0, 0.0}, q[2] = {0.0, 0.0}, l[2] = {0.0, 0.0}, m[2] = {0.0, 0.0};
  int64_t i = t;
  for (; i + (int64_t)nt < B; i += 2 * (int64_t)nt) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const double v = e[i + k * (int64_t)nt] - shift, lp = logp[i + k * (int64_t)nt];
      a[k] += v; q[k] = fma(v, v, q[k]); l[k] += lp; m[k] = fma(lp, v, m[k]);
    }
  }
  if (i < B) { const double v = e[i] - shift, lp = logp[i]; a[0] += v; q[0] = fma(v, v, q[0]); l[0] += lp; m[0] = fma(lp, v, m[0]); }
  sm[0][t] = a[0] + a[1]; sm[1][t] = q[0] + q[1]; sm[2][t] = l[0] + l[1]; sm[3][t] = m[0] + m[1];
  __syncthreads();
  int w = 1;
  while (w * 2 < nt) w *= 2;
  for (; w > 0; w >>= 1) {
    if (t < w && t + w < nt) {
#pragma unroll
      for (int k = 0; k < 4; k++) sm[k][t] += sm[k][t + w];
    }
    __syncthreads();
  }
  if (t < 4) out[t] = sm[t][0];
}

// The same four sums from many workgroups (one workgroup reads 1 MB at one CU's bandwidth: 19 us at 65 536 walkers): workgroup k
// reduces walkers [k FF_EST_SEG, (k + 1) FF_EST_SEG) with a fixed tree and stores its partials; the workgroup that finishes LAST
// (a counter in the workspace) adds the partials in workgroup order -- one fixed order whoever is last -- and, with n_global > 0
// (single rank: nothing to all-reduce), goes on to E, the centred sum of squares and the surrogate (ff_energy_finish).  It
// leaves the counter at zero: the workspace is zeroed by the caller ONCE, before its first use.
#define FF_EST_SEG 1024
#define FF_EST_THREADS 256
FF_D void ff_energy_finish_values(const double* sums, double shift, double n, double* est) {
  const double d = sums[0] / n;                       // E - c
  est[0] = shift + d;
  est[1] = sums[1] - sums[0] * d;                     // sum (e - E)^2 = sum (e - c)^2 - n (E - c)^2
  est[2] = (sums[3] - d * sums[2]) / n;               // mean(logp (e - E))
}
__global__ void __launch_bounds__(FF_EST_THREADS)
ff_energy_estimate_kernel(int64_t B, const double* __restrict__ e, const double* __restrict__ logp, const double* __restrict__ shift_dev,
                          double n_global, double* __restrict__ sums4, double* __restrict__ est3, double* __restrict__ part,
                          unsigned* __restrict__ counter) {
  __shared__ double sm[4][FF_EST_THREADS];
  __shared__ unsigned s_last;
  const double shift = ff_finite_shift(shift_dev[0]);
  const int t = threadIdx.x, nt = blockDim.x;
  const int64_t j0 = (int64_t)blockIdx.x * FF_EST_SEG;
  int w0 = 1;
  while (w0 * 2 < nt) w0 *= 2;      // largest power of two below the block size
  double a = 0.0, q = 0.0, l = 0.0, m = 0.0;
  for (int k = t; k < FF_EST_SEG && j0 + k < B; k += nt) {
    const double v = e[j0 + k] - shift, lp = logp[j0 + k];
    a += v; q = fma(v, v, q); l += lp; m = fma(lp, v, m);
  }
  sm[0][t] = a; sm[1][t] = q; sm[2][t] = l; sm[3][t] = m;
  __syncthreads();
  for (int w = w0; w > 0; w >>= 1) {
    if (t < w && t + w < nt) {
#pragma unroll
      for (int k = 0; k < 4; k++) sm[k][t] += sm[k][t + w];
    }
    __syncthreads();
  }
  if (t < 4) part[(int64_t)blockIdx.x * 4 + t] = sm[t][0];
  __threadfence();
  __syncthreads();
  if (t == 0) s_last = atomicAdd(counter, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // the last workgroup: partials in workgroup order (thread t takes k = t, t + nt, ... and the tree joins them -- the same
  // association for every run)
  double s4[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k = t; k < (int)gridDim.x; k += nt) {
#pragma unroll
    for (int c = 0; c < 4; c++) s4[c] += part[(int64_t)k * 4 + c];
  }
#pragma unroll
  for (int c = 0; c < 4; c++) sm[c][t] = s4[c];
  __syncthreads();
  for (int w = w0; w > 0; w >>= 1) {
    if (t < w && t + w < nt) {
#pragma unroll
      for (int k = 0; k < 4; k++) sm[k][t] += sm[k][t + w];
    }
    __syncthreads();
  }
  if (t == 0) {
    double tot[4] = {sm[0][0], sm[1][0], sm[2][0], sm[3][0]};
#pragma unroll
    for (int c = 0; c < 4; c++) sums4[c] = tot[c];
    if (n_global > 0.0 && est3) ff_energy_finish_values(tot, shift, n_global, est3);
    *counter = 0u;
  }
}

// est = [E, sum (e - E)^2, sum logp (e - E) / n] from the (all-reduced) sums of ff_energy_sums_kernel over n walkers
__global__ void ff_energy_finish_kernel(const double* __restrict__ sums, const double* __restrict__ shift_dev, double n,
                                        double* __restrict__ est) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  ff_energy_finish_values(sums, ff_finite_shift(shift_dev[0]), n, est);
}

// Per-state sums of the finite-temperature estimator (src/VMC.py:164-169: the per-state baseline of gradF_theta, and the
// state counts behind S and gradF_phi).  walker_state is sorted (src/VMC.py:94-96), so state s is one contiguous segment:
// one workgroup per state finds its bounds by bisection and sums the segment with a fixed tree -- deterministic, and no
// atomics (at beta = 10 every walker sits in state 0: 65536 atomic adds on one address took 6 ms).
__global__ void __launch_bounds__(256)
ff_state_sums_kernel(int64_t B, const int* __restrict__ ws, const double* __restrict__ e, double* __restrict__ sums,
                     double* __restrict__ counts) {
  __shared__ double sm[256];
  const int s = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  auto lower = [&](int key) -> int64_t {   // first index with ws[i] >= key
    int64_t lo = 0, hi = B;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ws[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
  };
  const int64_t b0 = lower(s), b1 = lower(s + 1);
  double a = 0.0;
  for (int64_t i = b0 + t; i < b1; i += nt) a += e[i];
  sm[t] = a;
  __syncthreads();
  int w = 1;
  while (w * 2 < nt) w *= 2;
  for (; w > 0; w >>= 1) {
    if (t < w && t + w < nt) sm[t] += sm[t + w];
    __syncthreads();
  }
  if (t == 0) { sums[s] = sm[0]; counts[s] = (double)(b1 - b0); }
}

// Finite-temperature estimator in two launches around the all-reduce (src/VMC.py:146-171).
// (1) ff_state_part_kernel: the sorted state list cut into FF_SS_K slices per state -- partial sums
//     part[s][k] = (sum e, count, sum logp, sum logp e) over slice k of state s, one workgroup each, fixed tree (at beta = 10
//     every walker sits in state 0: one workgroup per state took 31 us for 65 536 walkers).  Partial sums of all ranks add up.
#define FF_SS_K 16
__global__ void __launch_bounds__(256)
ff_state_part_kernel(int64_t B, const int* __restrict__ ws, const double* __restrict__ e, const double* __restrict__ logp,
                     double* __restrict__ part) {
  __shared__ double sm[3][256];
  const int s = blockIdx.x / FF_SS_K, k = blockIdx.x % FF_SS_K, t = threadIdx.x, nt = blockDim.x;
  auto lower = [&](int key) -> int64_t {   // first index with ws[i] >= key
    int64_t lo = 0, hi = B;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ws[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
  };
  const int64_t b0 = lower(s), b1 = lower(s + 1), len = b1 - b0;
  const int64_t c0 = b0 + len * k / FF_SS_K, c1 = b0 + len * (k + 1) / FF_SS_K;
  double a = 0.0, l = 0.0, m = 0.0;
  for (int64_t i = c0 + t; i < c1; i += nt) { const double ei = e[i], li = logp[i]; a += ei; l += li; m = fma(li, ei, m); }
  sm[0][t] = a; sm[1][t] = l; sm[2][t] = m;
  __syncthreads();
  int w = 1;
  while (w * 2 < nt) w *= 2;
  for (; w > 0; w >>= 1) {
    if (t < w && t + w < nt) { sm[0][t] += sm[0][t + w]; sm[1][t] += sm[1][t + w]; sm[2][t] += sm[2][t + w]; }
    __syncthreads();
  }
  if (t == 0) {
    double* o = part + ((size_t)s * FF_SS_K + k) * 4;
    o[0] = sm[0][0]; o[1] = (double)(c1 - c0); o[2] = sm[1][0]; o[3] = sm[2][0];
  }
}

// (2) ff_beta_finish_kernel, one workgroup: from buf = [sum (e - c), sum (e - c)^2 | part (all ranks added)], the logits and
//     beta it forms everything BetaVMC.forward reports and both surrogates' ingredients:
//     est = [E, sum (e - E)^2, F, sum (f - F)^2, S, S_analytical, gradF_phi value, gradF_theta value]
//     (f_b = e_b + log p(s_b) / beta; S = -mean log p(s_b); gradF_phi = mean(log p(s_b) (f_b - F)), src/VMC.py:162;
//      gradF_theta = mean(logp_b (e_b - mean_e[s_b])), src/VMC.py:164-169),
//     gphi[s] = d gradF_phi / d logits[s],  mean_e[s] = the per-state baseline,  logp_all = log_softmax(logits).
__global__ void __launch_bounds__(256)
ff_beta_finish_kernel(const double* __restrict__ buf, const double* __restrict__ shift_dev, const double* __restrict__ logits, int Ns,
                      double beta, double n, double* __restrict__ est, double* __restrict__ gphi, double* __restrict__ mean_e,
                      double* __restrict__ logp_all) {
  __shared__ double sm[256];
  const int t = threadIdx.x, nt = blockDim.x;
  auto block_sum = [&](double v) -> double {     // fixed tree; every thread gets the total
    sm[t] = v;
    __syncthreads();
    int w = 1;
    while (w * 2 < nt) w *= 2;
    for (; w > 0; w >>= 1) {
      if (t < w && t + w < nt) sm[t] += sm[t + w];
      __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return r;
  };
  auto block_max = [&](double v) -> double {
    sm[t] = v;
    __syncthreads();
    int w = 1;
    while (w * 2 < nt) w *= 2;
    for (; w > 0; w >>= 1) {
      if (t < w && t + w < nt) sm[t] = fmax(sm[t], sm[t + w]);
      __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return r;
  };
  const double* part = buf + 2;
  double mx = -1e300;
  for (int s = t; s < Ns; s += nt) mx = fmax(mx, logits[s]);
  mx = block_max(mx);
  double z = 0.0;
  for (int s = t; s < Ns; s += nt) z += exp(logits[s] - mx);
  const double lz = mx + log(block_sum(z));
  // per state: sums over the slices in a fixed order
  double sF = 0.0, sCE = 0.0, sCC = 0.0, sS = 0.0, sSa = 0.0;
  for (int s = t; s < Ns; s += nt) {
    double se = 0.0, cnt = 0.0;
    for (int k = 0; k < FF_SS_K; k++) { const double* o = part + ((size_t)s * FF_SS_K + k) * 4; se += o[0]; cnt += o[1]; }
    const double lp = logits[s] - lz, c = lp / beta;
    logp_all[s] = lp;
    mean_e[s] = se / fmax(cnt, 1.0);
    sF += se + cnt * c; sCE += c * se; sCC += cnt * c * c; sS += cnt * lp; sSa += lp * exp(lp);
  }
  sF = block_sum(sF); sCE = block_sum(sCE); sCC = block_sum(sCC); sS = block_sum(sS); sSa = block_sum(sSa);
  const double c0 = ff_finite_shift(shift_dev[0]), d = buf[0] / n;            // E - c
  const double E = c0 + d, Ess = buf[1] - buf[0] * d, F = sF / n;
  // sum f^2 = sum e^2 + 2 sum_s c_s sum_e(s) + sum_s cnt_s c_s^2,  sum e^2 = sum (e - c)^2 + 2 c sum (e - c) + n c^2
  const double se2 = buf[1] + 2.0 * c0 * buf[0] + n * c0 * c0;
  const double Fss = se2 + 2.0 * sCE + sCC - n * F * F;
  double sG = 0.0, sC = 0.0, sT = 0.0;
  for (int s = t; s < Ns; s += nt) {
    double se = 0.0, cnt = 0.0, sl = 0.0, sle = 0.0;
    for (int k = 0; k < FF_SS_K; k++) { const double* o = part + ((size_t)s * FF_SS_K + k) * 4; se += o[0]; cnt += o[1]; sl += o[2]; sle += o[3]; }
    const double lp = logits[s] - lz;
    const double cF = (se + cnt * lp / beta - cnt * F) / n;
    gphi[s] = cF;                      // completed below
    sG += lp * cF; sC += cF;
    sT += sle - (se / fmax(cnt, 1.0)) * sl;
  }
  sG = block_sum(sG); sC = block_sum(sC); sT = block_sum(sT);
  for (int s = t; s < Ns; s += nt) gphi[s] -= exp(logits[s] - lz) * sC;
  if (t == 0) {
    est[0] = E; est[1] = Ess; est[2] = F; est[3] = Fss; est[4] = -sS / n; est[5] = -sSa; est[6] = sG; est[7] = sT / n;
  }
}

// =================================================================================================
// C ABI
// =================================================================================================
extern void ff_set_error(const char* msg);
#define FF_CHECK(cond, code, msg) do { if (!(cond)) { ff_set_error(msg); return code; } } while (0)
#define FF_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { ff_set_error(hipGetErrorString(e_)); return FF_ELAUNCH; } } while (0)
static inline unsigned ff_grid(int64_t B, int block) { return (unsigned)((B + block - 1) / block); }

template <int NU, int ND>
static void launch_mcmc(bool noise, void* stream, int64_t B, int nup, int ndn, const int* tu, const int* td, const int* ws,
                        int steps, double tau, const double* g0, const double* g, const double* u, uint64_t seed, int64_t woff,
                        double* x_out, double* logp_out, uint8_t* accept, int* acc_count) {
  if constexpr (NU == ND && NU >= 1 && NU <= 6) {
    // two lanes per walker (one per spin): twice the waves, half the chain per lane (one lane per walker: 0.66 -> 0.45 ms).
    // Explicit noise: the reference's arithmetic operation for operation; Philox: the determinant-ratio kernel (3h)
    if (noise)
      FF_LAUNCH((ff_mcmc_spin_kernel<NU, true>), ff_grid(2 * B, 128), 128, stream, B, tu, td, ws, steps, tau, g0, g, u, seed, woff,
                x_out, logp_out, accept, acc_count);
    else
      FF_LAUNCH((ff_mcmc_spin_philox_kernel<NU>), ff_grid(2 * B, 128), 128, stream, B, tu, td, ws, steps, tau, g0, seed, woff,
                x_out, logp_out, acc_count);
    return;
  }
  if constexpr (ND == 0 && NU >= 2 && NU <= 6) {
    // one spin species: two lanes per walker split the particles (ff_mcmc_pair_kernel)
    if (noise)
      FF_LAUNCH((ff_mcmc_pair_kernel<NU, true>), ff_grid(2 * B, 128), 128, stream, B, tu, ws, steps, tau, g0, g, u, seed, woff, x_out,
                logp_out, accept, acc_count);
    else
      FF_LAUNCH((ff_mcmc_pair_kernel<NU, false>), ff_grid(2 * B, 128), 128, stream, B, tu, ws, steps, tau, g0, g, u, seed, woff, x_out,
                logp_out, accept, acc_count);
    return;
  }
  if (noise)
    FF_LAUNCH((ff_mcmc_kernel<NU, ND, true>), ff_grid(B, 128), 128, stream, B, nup, ndn, tu, td, ws, steps, tau, g0, g, u, seed, woff,
              x_out, logp_out, accept, acc_count);
  else
    FF_LAUNCH((ff_mcmc_kernel<NU, ND, false>), ff_grid(B, 128), 128, stream, B, nup, ndn, tu, td, ws, steps, tau, g0, g, u, seed, woff,
              x_out, logp_out, accept, acc_count);
}

extern int ff_mcmc_rows_launch(void* stream, int d, bool noise, int64_t B, int nup, int ndn, const int* tu, const int* td, const int* ws,
                               int steps, double tau, const double* g0, const double* g, const double* u, uint64_t seed, int64_t woff,
                               double* x_out, double* logp_out, uint8_t* accept, int* acc_count);

static int mcmc_dispatch(bool noise, void* stream, int64_t B, int nup, int ndn, const int* tu, const int* td, const int* ws,
                         int steps, double tau, const double* g0, const double* g, const double* u, uint64_t seed, int64_t woff,
                         double* x_out, double* logp_out, uint8_t* accept, int* acc_count) {
  FF_CHECK(B >= 0 && nup >= 0 && ndn >= 0 && nup + ndn > 0 && steps >= 0, FF_EINVAL, "ff_mcmc: bad sizes");
  FF_CHECK(nup <= FF_MAX_NS && ndn <= FF_MAX_NS, FF_EUNSUPPORTED, "ff_mcmc: determinant larger than FF_MAX_NS");
  FF_CHECK(x_out && (nup == 0 || tu) && (ndn == 0 || td), FF_EINVAL, "ff_mcmc: null pointer");
  FF_CHECK(!noise || (g0 && (steps == 0 || (g && u))), FF_EINVAL, "ff_mcmc: null noise pointer");
  if (B == 0) return FF_OK;
#define FF_MC(NU_, ND_) if (nup == NU_ && ndn == ND_) { launch_mcmc<NU_, ND_>(noise, stream, B, nup, ndn, tu, td, ws, steps, tau, g0, g, u, seed, woff, x_out, logp_out, accept, acc_count); FF_LAUNCH_CHECK(); return FF_OK; }
  FF_MC(3, 3) FF_MC(3, 0) FF_MC(6, 0) FF_MC(6, 6) FF_MC(1, 0) FF_MC(2, 0) FF_MC(4, 0)
  FF_MC(1, 1) FF_MC(2, 2) FF_MC(4, 4) FF_MC(5, 5) FF_MC(5, 0) FF_MC(10, 0)
#undef FF_MC
  // every other shape: sixteen lanes per determinant (ff_ho3d.hip; bit-identical to the one-lane general kernel it replaced)
  return ff_mcmc_rows_launch(stream, 2, noise, B, nup, ndn, tu, td, ws, steps, tau, g0, g, u, seed, woff, x_out, logp_out, accept, acc_count);
}

// ---------------------------------------------------------------------------------------------------
// Walker schedule: indices by descending cost class (two-pass counting sort, deterministic).
// Every walker of the fused ODE kernels adapts its own step size; walkers that pass close to a point where a radius
// vanishes need several times the usual number of steps.  Taking the expensive walkers first (and putting walkers of
// equal cost into the same waves) keeps them out of the tail of a launch.  cost is clamped to [0, FF_ORD_BINS); ties
// keep a fixed order (segment, thread, index).  Workspace: FF_ORD_BINS counters per FF_ORD_SEG walkers.
#define FF_ORD_BINS 32
#define FF_ORD_THREADS 256
#define FF_ORD_SEG 512       // (2048 until round 5: 32 workgroups for 65 536 walkers left seven eighths of the GPU idle for two launches)
FF_D int ff_ord_row(int c) { return FF_ORD_BINS - 1 - (c < 0 ? 0 : (c > FF_ORD_BINS - 1 ? FF_ORD_BINS - 1 : c)); }   // row 0 = most expensive

// First-step scale of a cost class (ff_walker_schedule).  The local-energy pass opens every walker with scale x (the largest step the
// flow pass accepted along the same trajectory), rounded down to equal steps.  A first step that fails its error test costs a whole
// step -- six evaluations, for the walker's wave -- but so does a scale that is too small, for EVERY walker of the class (one more of the
// equal steps): a rejection rate of 10 % is worth about 0.6 evaluations per walker (2 on the four-walkers-per-wave kernel, whose walkers
// wait for each other), the extra step 6.  On the benchmark's synthetic weights 0.9 is accepted by 99 % of the walkers; after a few
// hundred training iterations 80 % of them reject it (tools/probes/policy_sweep.py: 29 evaluations per walker where 23 do).  So the
// scale follows the pass: of the n walkers of the class in the previous pass nr rejected their first step -- more than shrink_at: scale
// x 0.93; within [0.25, 1]; classes with fewer than 64 walkers keep theirs.  (shrink_at is the caller's, by how many walkers the
// local-energy kernel advances in lockstep; 10 % at four, not the 20 % a walker on its own would tolerate: the four walkers of a matrix-core wave advance in lockstep, a rejection costs its WAVE two more attempts, and at a rate p
// that is 1 - (1 - p)^4 of the waves -- 34 % at p = 0.10, 59 % at 0.20.  Measured on settled tables, tools/probes/table_settle.py:
// trained flow 24.4 -> 23.9 evaluations per walker and pass 1.218 -> 1.170 ms, driver-1000 32.5 -> 31.2 and 1.586 -> 1.527 ms,
// nothing on the driver-300 and synthetic weights; 0.07: 23.8 / 30.6, passes 1.179 / 1.508.)  (Round 5's first thresholds, 4 % and
// 1 %, traded 6 evaluations of every walker for 7 of one in twenty: 13.9 -> 19.7 evaluations at 6 + 6 particles.)
// Growth (x 1.02) needs fewer than 5 % rejections AND evidence that a plan one step shorter would pass: of the n3 walkers that were
// planned for k >= 3 equal steps, ne accepted -- somewhere along the trajectory, where the step-size control of the solver tried it
// -- a step as large as the interval / (k - 1) of the shorter plan; growth wants n3 >= 16 and ne >= 0.7 n3.  Without that condition
// the table probes blindly: 200 iterations into the benchmark's training run every class sat at three steps of 1/3 with no
// rejections, grew by 1.02 for four iterations until two steps of 1/2 were planned, had 40-70 % of them rejected, shrank, and so
// on -- one iteration in five at 24 evaluations per walker instead of 21 (tools/probes/h_table_drift.py); with it the same walkers
// report ne / n3 = 0.00-0.05 at three steps and the table stays.  (Walkers planned for two steps cannot show such a step -- the
// second one is capped by the rest of the interval -- and have nothing to gain short of a single step; they do not vote.)
// Error control is untouched: every step passes the same test whatever it opened with.
#ifndef FF_GROW_EVIDENCE
#define FF_GROW_EVIDENCE 0.7      // (A/B knob)
#endif
FF_D double ff_scale_update(double cur, unsigned n, unsigned nr, unsigned n3, unsigned ne, double shrink_at) {
  if (!(cur > 0.0)) cur = 0.6;
  if (n >= 64u) {
    const double f = (double)nr / (double)n;
    if (f > shrink_at) cur *= 0.93;
    else if (f < 0.5 * shrink_at && n3 >= 16u && (double)ne >= FF_GROW_EVIDENCE * (double)n3) cur *= 1.02;
  }
  return fmin(1.0, fmax(0.25, cur));
}
// the two votes of one walker of the previous pass for ff_scale_update: bit 0 -- it was planned for three or more equal steps (without an
// interval: every walker), bit 1 -- and accepted a step of the plan one shorter (without an interval: 1.25 x the step it opened with)
FF_D unsigned ff_scale_votes(double hs0, double he0, double interval) {
  if (!(interval > 0.0)) return 1u | (he0 >= 1.25 * hs0 ? 2u : 0u);
  const double k = rint(interval / hs0);
  if (k < 3.0) return 0u;
  return 1u | (he0 >= 0.999 * interval / (k - 1.0) ? 2u : 0u);
}

// Sort key of a walker in ff_walker_schedule: its cost class -- raised by four for every equal step beyond two that its local-energy pass
// is planned to take (interval / (hval x the class's factor), rounded up).  The four walkers of a wave of the matrix-core kernel advance
// in lockstep, so a wave takes as many evaluations as its slowest walker: walkers that will take three steps belong with each other (and
// with the close-approach classes, which take that many anyway), not scattered over the two-step waves of their class.  The key only
// orders the work; tolerances, routing and the table's statistics go by the class itself.
FF_D int ff_sched_key(int c, double hv, const double* __restrict__ tab, double interval) {
  int cc = c < 0 ? 0 : (c > FF_ORD_BINS - 1 ? FF_ORD_BINS - 1 : c);
  if (tab == nullptr || !(interval > 0.0) || !(hv > 0.0)) return cc;
  double f = tab[cc];
  if (!(f > 0.0)) f = 0.6;
  const double hq = hv * f;
  int k = hq >= interval ? 1 : (int)ceil(interval / hq - 1e-9);
  k = k > 8 ? 8 : k;
  const int key = cc + 4 * (k > 2 ? k - 2 : 0);
  return key > FF_ORD_BINS - 1 ? FF_ORD_BINS - 1 : key;
}

// pass 1: per-segment histogram (+ optionally the segment's sum of hval, fixed tree: the sweeps want the mean accepted step of
// the flow pass, and a torch mean() was two more launches)
// (ff_walker_schedule: + per-segment counts, by cost class, of the walkers of the PREVIOUS local-energy pass and of those whose first
// step was rejected there -- the largest step that pass accepted for the walker, prev_he, is smaller than the step it opened with,
// prev_hs; integer counts, so their sum over segments does not depend on any order)
__global__ void __launch_bounds__(FF_ORD_THREADS) ff_order_count_kernel(int64_t B, const int32_t* __restrict__ cost,
                                                                        unsigned* __restrict__ hist, const double* __restrict__ hval,
                                                                        double* __restrict__ hsum, int64_t Bprev,
                                                                        const int32_t* __restrict__ prev_cost, const double* __restrict__ prev_hs,
                                                                        const double* __restrict__ prev_he, unsigned* __restrict__ pstat,
                                                                        const double* __restrict__ tab_in, double interval) {
  FF_SETPRIO();
  __shared__ unsigned h[FF_ORD_BINS], pn[FF_ORD_BINS], pr[FF_ORD_BINS], p3[FF_ORD_BINS], pe[FF_ORD_BINS];
  __shared__ double sh[FF_ORD_THREADS];
  const int t = threadIdx.x;
  if (t < FF_ORD_BINS) { h[t] = 0; pn[t] = 0; pr[t] = 0; p3[t] = 0; pe[t] = 0; }
  __syncthreads();
  const int64_t j0 = (int64_t)blockIdx.x * FF_ORD_SEG;
  double acc = 0.0;
  for (int k = t; k < FF_ORD_SEG && j0 + k < B; k += FF_ORD_THREADS) {
    const double hv = hval ? hval[j0 + k] : 0.0;
    atomicAdd(&h[ff_ord_row(ff_sched_key(cost[j0 + k], hv, tab_in, interval))], 1u);
    acc += hv;
  }
  if (pstat) {
    for (int k = t; k < FF_ORD_SEG && j0 + k < Bprev; k += FF_ORD_THREADS) {
      const double hs0 = prev_hs[j0 + k], he0 = prev_he[j0 + k];
      if (hs0 > 0.0 && he0 > 0.0) {       // (a failed or cold-started walker says nothing about the scale)
        const int row = ff_ord_row(prev_cost[j0 + k]);
        atomicAdd(&pn[row], 1u);
        // "first step rejected" is INFERRED: he0 is the largest step accepted anywhere in the pass, so a walker whose first step was
        // rejected and whose later steps grew back to hs0 counts as accepted -- the shrink rule under-counts rejections.
        // Performance only (the per-step error test is untouched), and the thresholds of ff_scale_update were set against THIS count:
        // left as it is with the controller frozen; an explicit flag from the kernels is the clean form.
        if (he0 < 0.999 * hs0) atomicAdd(&pr[row], 1u);
        const unsigned v = ff_scale_votes(hs0, he0, interval);
        if (v & 1u) atomicAdd(&p3[row], 1u);
        if (v & 2u) atomicAdd(&pe[row], 1u);
      }
    }
  }
  sh[t] = acc;
  __syncthreads();
  if (t < FF_ORD_BINS) hist[(int64_t)blockIdx.x * FF_ORD_BINS + t] = h[t];
  if (pstat && t < FF_ORD_BINS) {
    unsigned* ps = pstat + ((int64_t)blockIdx.x * FF_ORD_BINS + t) * 4;
    ps[0] = pn[t]; ps[1] = pr[t]; ps[2] = p3[t]; ps[3] = pe[t];
  }
  if (hval) {
    for (int q = FF_ORD_THREADS / 2; q > 0; q >>= 1) {
      if (t < q) sh[t] += sh[t + q];
      __syncthreads();
    }
    if (t == 0) hsum[blockIdx.x] = sh[0];
  }
}

// pass 2: position = (walkers in more expensive rows) + (same row, earlier segments) + (same row and segment, earlier
// (thread, index)); the last term from per-thread counters, so no atomics decide an order
__global__ void __launch_bounds__(FF_ORD_THREADS) ff_order_place_kernel(int64_t B, const int32_t* __restrict__ cost,
                                                                        const unsigned* __restrict__ hist, int nseg,
                                                                        int32_t* __restrict__ order, const double* __restrict__ hsum,
                                                                        double* __restrict__ hmean, const unsigned* __restrict__ pstat,
                                                                        int nseg_prev, const double* __restrict__ tab_in,
                                                                        double* __restrict__ tab_out, const double* __restrict__ hval,
                                                                        double* __restrict__ hs_out, double interval,
                                                                        const double* __restrict__ counts, double shrink_at) {
  FF_SETPRIO();
  __shared__ unsigned cnt[FF_ORD_BINS][FF_ORD_THREADS + 1];
  __shared__ unsigned tot[FF_ORD_THREADS / FF_ORD_BINS][FF_ORD_BINS], before[FF_ORD_THREADS / FF_ORD_BINS][FF_ORD_BINS];
  __shared__ unsigned pn[FF_ORD_THREADS / FF_ORD_BINS][FF_ORD_BINS], pr[FF_ORD_THREADS / FF_ORD_BINS][FF_ORD_BINS];
  __shared__ unsigned p3[FF_ORD_THREADS / FF_ORD_BINS][FF_ORD_BINS], pe[FF_ORD_THREADS / FF_ORD_BINS][FF_ORD_BINS];
  __shared__ unsigned base[FF_ORD_BINS];
  __shared__ double s_tab[FF_ORD_BINS], s_hs[FF_ORD_THREADS];
  const int t = threadIdx.x, seg = blockIdx.x;
  // row totals over all segments and over the earlier segments (thread t: row t % BINS, every (THREADS/BINS)-th segment) -- and, the same
  // way, the previous pass' first-step statistics by class (ff_walker_schedule)
  {
    constexpr int SL = FF_ORD_THREADS / FF_ORD_BINS;
    const int row = t % FF_ORD_BINS, sl = t / FF_ORD_BINS;
    unsigned a = 0, b = 0, n = 0, nr = 0, n3 = 0, ne = 0;
    for (int k = sl; k < nseg; k += SL) {
      const unsigned v = hist[(int64_t)k * FF_ORD_BINS + row];
      a += v;
      b += k < seg ? v : 0u;
    }
    if (pstat && !counts)
      for (int k = sl; k < nseg_prev; k += SL) {
        const unsigned* ps = pstat + ((int64_t)k * FF_ORD_BINS + row) * 4;
        n += ps[0]; nr += ps[1]; n3 += ps[2]; ne += ps[3];
      }
    tot[sl][row] = a;
    before[sl][row] = b;
    pn[sl][row] = n;
    pr[sl][row] = nr;
    p3[sl][row] = n3;
    pe[sl][row] = ne;
  }
  if (hmean && blockIdx.x == 0) {      // the segments' sums: strided partial sums, then a fixed tree -- one summation order whatever the timing
    double a = 0.0;
    for (int k = t; k < nseg; k += FF_ORD_THREADS) a += hsum[k];
    s_hs[t] = a;
  }
  __syncthreads();
  // ff_walker_schedule: the first-step scale of every cost class, learned from the previous pass (ff_scale_update) -- every workgroup
  // forms the same table from the same integer counts; workgroup 0 stores it for the next call
  if (tab_in && t < FF_ORD_BINS) {
    const int row = t;                 // row 0 = class 31
    unsigned n = 0, nr = 0, n3 = 0, ne = 0;
    if (counts) {      // the previous pass' statistics summed over every rank's shard (ff_scale_counts + the caller's all-reduce)
      const int c = FF_ORD_BINS - 1 - row;
      n = (unsigned)counts[c]; nr = (unsigned)counts[FF_ORD_BINS + c]; n3 = (unsigned)counts[2 * FF_ORD_BINS + c]; ne = (unsigned)counts[3 * FF_ORD_BINS + c];
    } else {
      for (int sl = 0; sl < FF_ORD_THREADS / FF_ORD_BINS; sl++) { n += pn[sl][row]; nr += pr[sl][row]; n3 += p3[sl][row]; ne += pe[sl][row]; }
    }
    const double v = ff_scale_update(tab_in[FF_ORD_BINS - 1 - row], n, nr, n3, ne, shrink_at);
    s_tab[row] = v;
    if (blockIdx.x == 0 && tab_out) tab_out[FF_ORD_BINS - 1 - row] = v;
  }
  if (hmean && blockIdx.x == 0) {
    for (int q = FF_ORD_THREADS / 2; q > 0; q >>= 1) {
      if (t < q) s_hs[t] += s_hs[t + q];
      __syncthreads();
    }
    if (t == 0) hmean[0] = s_hs[0] / (double)B;
  }
  for (int k = 0; k < FF_ORD_BINS; k++) cnt[k][t] = 0;
  const int64_t j0 = (int64_t)seg * FF_ORD_SEG;
  constexpr int PERT = FF_ORD_SEG / FF_ORD_THREADS;
  int krow[PERT];      // row of the sort key of this thread's walkers (ff_sched_key: a division and a ceil -- once)
#pragma unroll
  for (int q = 0; q < PERT; q++) {
    const int k = t + q * FF_ORD_THREADS;
    krow[q] = -1;
    if (j0 + k < B) {
      krow[q] = ff_ord_row(ff_sched_key(cost[j0 + k], hs_out ? hval[j0 + k] : 0.0, hs_out ? tab_in : nullptr, interval));
      cnt[krow[q]][t]++;
    }
  }
  __syncthreads();
  if (t == 0) {
    unsigned run = 0;
    for (int r = 0; r < FF_ORD_BINS; r++) {
      unsigned a = 0, b = 0;
      for (int sl = 0; sl < FF_ORD_THREADS / FF_ORD_BINS; sl++) { a += tot[sl][r]; b += before[sl][r]; }
      base[r] = run + b;
      run += a;
    }
  }
  // exclusive scan of each row's per-thread counters: one wave-sized group of threads per row, serial over 4 chunks
  {
    const int row = t / (FF_ORD_THREADS / FF_ORD_BINS), part = t % (FF_ORD_THREADS / FF_ORD_BINS);
    constexpr int PER = FF_ORD_BINS;   // THREADS / (THREADS / BINS) counters per thread
    unsigned sum = 0;
    for (int k = 0; k < PER; k++) sum += cnt[row][part * PER + k];
    __syncthreads();
    tot[part][row] = sum;              // tot is free again (base is final only after the barrier below)
    __syncthreads();
    unsigned off = base[row];
    for (int q = 0; q < part; q++) off += tot[q][row];
    for (int k = 0; k < PER; k++) { const unsigned c = cnt[row][part * PER + k]; cnt[row][part * PER + k] = off; off += c; }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PERT; q++) {
    const int k = t + q * FF_ORD_THREADS;
    if (krow[q] < 0) continue;
    const int row = ff_ord_row(cost[j0 + k]);
    order[cnt[krow[q]][t]++] = (int32_t)(j0 + k);
    if (hs_out) {
      // the largest step this class is trusted with -- and of the steps of that size the interval takes, the EQUAL ones: two steps of
      // 0.5 are accepted where 0.57 + 0.43 risks a rejection for the same number of evaluations
      double hq = hval[j0 + k] * s_tab[row];
      if (interval > 0.0 && hq > 0.0 && hq < interval) hq = interval / ceil(interval / hq - 1e-9);
      hs_out[j0 + k] = hq;
    }
  }
}

extern "C" {

int ff_mcmc_sample_noise(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                         const int32_t* walker_state, int steps, double tau, const double* g0, const double* g,
                         const double* u, double* x_out, double* logp_out, uint8_t* accept) {
  return mcmc_dispatch(true, stream, B, nup, ndn, tab_up, tab_dn, walker_state, steps, tau, g0, g, u, 0, 0, x_out, logp_out, accept, nullptr);
}

int ff_mcmc_sample(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                   const int32_t* walker_state, int steps, double tau, uint64_t seed, int64_t walker_offset,
                   double* x_out, double* logp_out, int32_t* accept_count) {
  return mcmc_dispatch(false, stream, B, nup, ndn, tab_up, tab_dn, walker_state, steps, tau, nullptr, nullptr, nullptr, seed,
                       walker_offset, x_out, logp_out, nullptr, accept_count);
}

int ff_mcmc_continue(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
                     const int32_t* walker_state, int steps, double tau, uint64_t seed, int64_t walker_offset,
                     const double* x_init, double* x_out, double* logp_out, int32_t* accept_count) {
  FF_CHECK(x_init != nullptr || B == 0, FF_EINVAL, "ff_mcmc_continue: null x_init");
  return mcmc_dispatch(false, stream, B, nup, ndn, tab_up, tab_dn, walker_state, steps, tau, x_init, nullptr, nullptr, seed,
                       walker_offset, x_out, logp_out, nullptr, accept_count);
}

int ff_rng_fill(void* stream, int64_t B, int n, int steps, uint64_t seed, int64_t walker_offset, double* g0, double* g, double* u) {
  FF_CHECK(B >= 0 && n > 0 && steps >= 0 && g0 && (steps == 0 || (g && u)), FF_EINVAL, "ff_rng_fill: bad argument");
  if (B == 0) return FF_OK;
  FF_LAUNCH(ff_rng_fill_kernel, ff_grid(B, 128), 128, stream, B, n, steps, seed, walker_offset, g0, g, u);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_slater_logabsdet_fwd(void* stream, int64_t B, int n, const int32_t* orb_table, const int32_t* walker_state,
                            const double* x, double* logabsdet) {
  FF_CHECK(B >= 0 && n > 0 && orb_table && x && logabsdet, FF_EINVAL, "ff_slater_logabsdet_fwd: bad argument");
  FF_CHECK(n <= FF_MAX_NS, FF_EUNSUPPORTED, "ff_slater_logabsdet_fwd: n > FF_MAX_NS");
  if (B == 0) return FF_OK;
#define FF_SFW(NS_) case NS_: FF_LAUNCH((ff_slater_fwd_fixed_kernel<NS_>), ff_grid(B, 128), 128, stream, B, orb_table, walker_state, x, logabsdet); FF_LAUNCH_CHECK(); return FF_OK;
  switch (n) { FF_SFW(1) FF_SFW(2) FF_SFW(3) FF_SFW(4) FF_SFW(5) FF_SFW(6) default: break; }
#undef FF_SFW
  FF_LAUNCH(ff_slater_fwd_kernel, ff_grid(B, 128), 128, stream, B, n, orb_table, walker_state, x, logabsdet);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_slater_logabsdet_bwd(void* stream, int64_t B, int n, const int32_t* orb_table, const int32_t* walker_state,
                            const double* x, const double* grad_out, double* grad_x) {
  FF_CHECK(B >= 0 && n > 0 && orb_table && x && grad_out && grad_x, FF_EINVAL, "ff_slater_logabsdet_bwd: bad argument");
  FF_CHECK(n <= FF_MAX_NS, FF_EUNSUPPORTED, "ff_slater_logabsdet_bwd: n > FF_MAX_NS");
  if (B == 0) return FF_OK;
#define FF_SBW(NS_) case NS_: FF_LAUNCH((ff_slater_bwd_fixed_kernel<NS_>), ff_grid(B, 128), 128, stream, B, orb_table, walker_state, x, grad_out, grad_x); FF_LAUNCH_CHECK(); return FF_OK;
  switch (n) { FF_SBW(1) FF_SBW(2) FF_SBW(3) FF_SBW(4) default: break; }
#undef FF_SBW
  FF_LAUNCH(ff_slater_bwd_kernel, ff_grid(B, 128), 128, stream, B, n, orb_table, walker_state, x, grad_out, grad_x);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_logprob(void* stream, int64_t B, int nup, int ndn, const int32_t* tab_up, const int32_t* tab_dn,
               const int32_t* walker_state, const double* x, double* logp, double* grad, double* lap) {
  FF_CHECK(B >= 0 && nup >= 0 && ndn >= 0 && nup + ndn > 0 && x && logp, FF_EINVAL, "ff_logprob: bad argument");
  FF_CHECK((nup == 0 || tab_up) && (ndn == 0 || tab_dn), FF_EINVAL, "ff_logprob: null orbital table");
  FF_CHECK(nup <= FF_MAX_NS && ndn <= FF_MAX_NS, FF_EUNSUPPORTED, "ff_logprob: determinant larger than FF_MAX_NS");
  if (B == 0) return FF_OK;
  {
    const int nsf = (nup == ndn || ndn == 0) ? nup : (nup == 0 ? ndn : 0);   // one determinant size for both spin species
    const bool deriv = grad || lap;
#define FF_LPF(NS_) case NS_: \
      if (deriv) FF_LAUNCH((ff_logprob_fixed_kernel<NS_, true>), ff_grid(B, 128), 128, stream, B, nup, ndn, tab_up, tab_dn, walker_state, x, logp, grad, lap); \
      else FF_LAUNCH((ff_logprob_fixed_kernel<NS_, false>), ff_grid(B, 128), 128, stream, B, nup, ndn, tab_up, tab_dn, walker_state, x, logp, grad, lap); \
      FF_LAUNCH_CHECK(); return FF_OK;
    switch (nsf) { FF_LPF(1) FF_LPF(2) FF_LPF(3) FF_LPF(4) default: break; }
#undef FF_LPF
  }
  FF_LAUNCH(ff_logprob_kernel, ff_grid(B, 128), 128, stream, B, nup, ndn, tab_up, tab_dn, walker_state, x, logp, grad, lap);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_mlp_eval(void* stream, int64_t N, int H, const double* w1, const double* b1, const double* w2, const double* r,
                double* val, double* dval) {
  FF_CHECK(N >= 0 && H > 0 && w1 && b1 && w2 && r && val, FF_EINVAL, "ff_mlp_eval: bad argument");
  if (N == 0) return FF_OK;
  FF_LAUNCH(ff_mlp_kernel, ff_grid(N, 128), 128, stream, N, H, w1, b1, w2, r, val, dval);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_mlp_eval_nd(void* stream, int64_t N, int D_in, int H, const double* w1, const double* b1, const double* w2, const double* x,
                   double* val, double* grad) {
  FF_CHECK(N >= 0 && D_in > 0 && H > 0 && w1 && b1 && w2 && x && (val || grad), FF_EINVAL, "ff_mlp_eval_nd: bad argument");
  FF_CHECK(D_in <= FF_MLP_DMAX, FF_EUNSUPPORTED, "ff_mlp_eval_nd: D_in > 64");
  if (N == 0) return FF_OK;
  FF_LAUNCH(ff_mlp_nd_kernel, ff_grid(N, 128), 128, stream, N, D_in, H, w1, b1, w2, x, val, grad);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_backflow_vjp(void* stream, int64_t B, int n, int d, const ff_net* net, const double* x, const double* w, double* Aw, double* gdiv) {
  FF_CHECK(B >= 0 && n > 0 && d > 0 && net && x && ((w && Aw) || gdiv) && (Aw == nullptr || w != nullptr), FF_EINVAL, "ff_backflow_vjp: bad argument");
  FF_CHECK(net->He > 0 && net->ew1 && net->eb1 && net->ew2 && (net->Hm == 0 || (net->mw1 && net->mb1 && net->mw2)), FF_EINVAL,
           "ff_backflow_vjp: bad net");
  FF_CHECK(n <= FF_MAX_N && d <= 3, FF_EUNSUPPORTED, "ff_backflow_vjp: n > 24 or d > 3");
  if (B == 0) return FF_OK;
  FF_LAUNCH(ff_backflow_vjp_kernel, ff_grid(B, 128), 128, stream, B, n, d, *net, x, w, Aw, gdiv);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_backflow_v_div(void* stream, int64_t B, int n, int d, const ff_net* net, const double* x, double* v, double* div) {
  FF_CHECK(B >= 0 && n > 0 && d > 0 && net && x && (v || div), FF_EINVAL, "ff_backflow_v_div: bad argument");
  FF_CHECK(net->He > 0 && net->ew1 && net->eb1 && net->ew2 && (net->Hm == 0 || (net->mw1 && net->mb1 && net->mw2)), FF_EINVAL,
           "ff_backflow_v_div: bad net");
  FF_CHECK(n <= FF_MAX_N && d <= 3, FF_EUNSUPPORTED, "ff_backflow_v_div: n > 24 or d > 3");
  if (B == 0) return FF_OK;
  FF_LAUNCH(ff_backflow_kernel, ff_grid(B, 128), 128, stream, B, n, d, *net, x, v, div);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_potential(void* stream, int64_t B, int n, int d, double Z, int use_ho, const double* x, double* V) {
  FF_CHECK(B >= 0 && n > 0 && d > 0 && x && V, FF_EINVAL, "ff_potential: bad argument");
  FF_CHECK(n <= FF_MAX_N && d <= 3, FF_EUNSUPPORTED, "ff_potential: n > 24 or d > 3");
  if (B == 0) return FF_OK;
  {
    const int64_t ntiles = (B + FF_WAVE - 1) / FF_WAVE;
    const unsigned pgrid = (unsigned)(ntiles < 65536 ? ntiles : 65536);
#define FF_PS(N_, D_) if (n == N_ && d == D_) { FF_LAUNCH((ff_potential_stream_kernel<N_, D_>), pgrid, FF_WAVE, stream, B, Z, use_ho, x, V); FF_LAUNCH_CHECK(); return FF_OK; }
    FF_PS(6, 2) FF_PS(12, 2) FF_PS(3, 2) FF_PS(2, 2) FF_PS(4, 2) FF_PS(5, 2) FF_PS(8, 2) FF_PS(10, 2) FF_PS(6, 3) FF_PS(4, 3)
#undef FF_PS
  }
  FF_LAUNCH(ff_potential_kernel, ff_grid(B, 128), 128, stream, B, n, d, Z, use_ho, x, V);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

// counts[c] += walkers of class c of this pass that opened with a step, counts[32 + c] += those of them whose first step was rejected,
// counts[64 + c] / counts[96 + c] += the two votes of ff_scale_votes
// (integers in doubles: exact whatever the order of the atomics, and what an all-reduce over ranks adds up)
__global__ void __launch_bounds__(FF_ORD_THREADS) ff_scale_counts_kernel(int64_t B, const int32_t* __restrict__ cost, const double* __restrict__ hs,
                                                                         const double* __restrict__ he, double interval, double* __restrict__ counts) {
  __shared__ unsigned pc[4][FF_ORD_BINS];
  const int t = threadIdx.x;
  if (t < 4 * FF_ORD_BINS) pc[t / FF_ORD_BINS][t % FF_ORD_BINS] = 0;
  __syncthreads();
  const int64_t j0 = (int64_t)blockIdx.x * FF_ORD_SEG;
  for (int k = t; k < FF_ORD_SEG && j0 + k < B; k += FF_ORD_THREADS) {
    const double hs0 = hs[j0 + k], he0 = he[j0 + k];
    if (hs0 > 0.0 && he0 > 0.0) {
      const int c = FF_ORD_BINS - 1 - ff_ord_row(cost[j0 + k]);
      atomicAdd(&pc[0][c], 1u);
      if (he0 < 0.999 * hs0) atomicAdd(&pc[1][c], 1u);
      const unsigned v = ff_scale_votes(hs0, he0, interval);
      if (v & 1u) atomicAdd(&pc[2][c], 1u);
      if (v & 2u) atomicAdd(&pc[3][c], 1u);
    }
  }
  __syncthreads();
  if (t < 4 * FF_ORD_BINS && pc[t / FF_ORD_BINS][t % FF_ORD_BINS]) atomicAdd(&counts[t], (double)pc[t / FF_ORD_BINS][t % FF_ORD_BINS]);
}

int ff_scale_counts(void* stream, int64_t B, const int32_t* cost, const double* hs, const double* he, double interval, double* counts128) {
  FF_CHECK(B >= 0 && counts128 && (B == 0 || (cost && hs && he)), FF_EINVAL, "ff_scale_counts: bad argument");
  if (B == 0) return FF_OK;
  FF_LAUNCH(ff_scale_counts_kernel, (unsigned)((B + FF_ORD_SEG - 1) / FF_ORD_SEG), FF_ORD_THREADS, stream, B, cost, hs, he, interval, counts128);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

// [nseg][BINS] histogram | nseg segment sums of hval | [nseg][BINS][4] statistics of the previous pass (ff_walker_schedule)
size_t ff_walker_order_workspace_bytes(int64_t B) {
  const size_t nseg = (size_t)((B + FF_ORD_SEG - 1) / FF_ORD_SEG > 0 ? (B + FF_ORD_SEG - 1) / FF_ORD_SEG : 1);
  return sizeof(unsigned) * FF_ORD_BINS * nseg + sizeof(double) * nseg + sizeof(unsigned) * 4 * FF_ORD_BINS * nseg;
}

int ff_walker_schedule(void* stream, int64_t B, const int32_t* cost, int32_t* order, void* workspace, const double* hval, double* hmean,
                       const double* scale_in, double* scale_out, const int32_t* prev_cost, const double* prev_hs, const double* prev_he,
                       const double* prev_counts, double interval, double* hs_out, double shrink_at) {
  FF_CHECK(B >= 0 && (B == 0 || (cost && order && workspace)), FF_EINVAL, "ff_walker_order: bad argument");
  FF_CHECK((hval == nullptr) == (hmean == nullptr), FF_EINVAL, "ff_walker_order_mean: hval and hmean go together");
  FF_CHECK(B < ((int64_t)1 << 31), FF_EUNSUPPORTED, "ff_walker_order: B >= 2^31");
  FF_CHECK(scale_in == nullptr || (hval && hs_out && scale_out && scale_out != scale_in), FF_EINVAL,
           "ff_walker_schedule: the scale table needs hval, hs_out and a second table to write");
  FF_CHECK((prev_cost == nullptr) == (prev_hs == nullptr) && (prev_cost == nullptr) == (prev_he == nullptr) && (prev_cost == nullptr || scale_in),
           FF_EINVAL, "ff_walker_schedule: prev_cost, prev_hs, prev_he go together (and with the scale table)");
  FF_CHECK(prev_counts == nullptr || (scale_in && prev_cost == nullptr), FF_EINVAL, "ff_walker_schedule: prev_counts OR the previous pass' arrays");
  if (B == 0) return FF_OK;
  const int nseg = (int)((B + FF_ORD_SEG - 1) / FF_ORD_SEG);
  double* hsum = (double*)((unsigned*)workspace + (size_t)FF_ORD_BINS * nseg);
  unsigned* pstat = prev_cost ? (unsigned*)(hsum + nseg) : nullptr;
  // (the previous pass is this rank's shard of the same batch: the statistics cover min(B, B_prev) = B walkers of it)
  FF_LAUNCH(ff_order_count_kernel, (unsigned)nseg, FF_ORD_THREADS, stream, B, cost, (unsigned*)workspace, hval, hsum, B, prev_cost, prev_hs,
            prev_he, pstat, scale_in ? scale_in : (const double*)nullptr, scale_in ? interval : 0.0);
  FF_LAUNCH_CHECK();
  FF_LAUNCH(ff_order_place_kernel, (unsigned)nseg, FF_ORD_THREADS, stream, B, cost, (const unsigned*)workspace, nseg, order,
            (const double*)hsum, hmean, (const unsigned*)pstat, nseg, scale_in, scale_out, hval, scale_in ? hs_out : (double*)nullptr,
            interval, prev_counts, shrink_at > 0.0 ? (shrink_at < 0.02 ? 0.02 : (shrink_at > 0.5 ? 0.5 : shrink_at)) : 0.10);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_walker_order_mean(void* stream, int64_t B, const int32_t* cost, int32_t* order, void* workspace, const double* hval, double* hmean) {
  return ff_walker_schedule(stream, B, cost, order, workspace, hval, hmean, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, nullptr, 0.0);
}

int ff_walker_order(void* stream, int64_t B, const int32_t* cost, int32_t* order, void* workspace) {
  return ff_walker_order_mean(stream, B, cost, order, workspace, nullptr, nullptr);
}

int ff_reduce_moments(void* stream, int64_t B, const double* e, double shift, const double* shift_dev, double shift_dev_scale,
                      double* out2) {
  FF_CHECK(B > 0 && e && out2, FF_EINVAL, "ff_reduce_moments: bad argument");
  FF_LAUNCH(ff_moments_kernel, 1, FF_RBLOCK(1024), stream, B, e, shift, shift_dev, shift_dev_scale, out2);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_adam_step(void* stream, int ntensors, const int64_t* sizes, double* const* params, const double* const* grads, double* const* exp_avg,
                 double* const* exp_avg_sq, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step) {
  FF_CHECK(ntensors >= 0 && (ntensors == 0 || (sizes && params && grads && exp_avg && exp_avg_sq)), FF_EINVAL, "ff_adam_step: bad argument");
  FF_CHECK(step >= 1 && lr >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && weight_decay >= 0.0, FF_EINVAL,
           "ff_adam_step: step >= 1, lr >= 0, betas in [0, 1), eps >= 0, weight_decay >= 0");
  for (int t0 = 0; t0 < ntensors; t0 += FF_ADAM_MAXT) {
    ff_adam_args a = {};
    const int nt = ntensors - t0 < FF_ADAM_MAXT ? ntensors - t0 : FF_ADAM_MAXT;
    for (int t = 0; t < nt; t++) {
      FF_CHECK(sizes[t0 + t] >= 0 && (sizes[t0 + t] == 0 || (params[t0 + t] && grads[t0 + t] && exp_avg[t0 + t] && exp_avg_sq[t0 + t])), FF_EINVAL,
               "ff_adam_step: null tensor");
      a.size[t] = sizes[t0 + t]; a.p[t] = params[t0 + t]; a.g[t] = grads[t0 + t]; a.m[t] = exp_avg[t0 + t]; a.v[t] = exp_avg_sq[t0 + t];
    }
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = 1.0 - pow(beta1, (double)step); a.sqrt_bc2 = sqrt(1.0 - pow(beta2, (double)step));
    FF_LAUNCH(ff_adam_kernel, (unsigned)nt, 256, stream, a);
    FF_LAUNCH_CHECK();
  }
  return FF_OK;
}

int ff_stream_delay(void* stream, double microseconds) {
  FF_CHECK(microseconds >= 0.0 && microseconds <= 1e5, FF_EINVAL, "ff_stream_delay: 0 .. 1e5 microseconds");
  FF_LAUNCH(ff_delay_kernel, 1, FF_RBLOCK(64), stream, (unsigned long long)(microseconds * 100.0));   // 100 MHz constant clock
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_reduce_energy(void* stream, int64_t B, const double* e, const double* logp, const double* shift_dev, double* sums4) {
  FF_CHECK(B > 0 && e && logp && shift_dev && sums4, FF_EINVAL, "ff_reduce_energy: bad argument");
  FF_LAUNCH(ff_energy_sums_kernel, 1, FF_RBLOCK(1024), stream, B, e, logp, shift_dev, sums4);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

size_t ff_energy_estimate_workspace_bytes(int64_t B) {
  const size_t nb = (size_t)((B + FF_EST_SEG - 1) / FF_EST_SEG > 0 ? (B + FF_EST_SEG - 1) / FF_EST_SEG : 1);
  return sizeof(double) * (4 * nb + 1);
}

int ff_energy_estimate(void* stream, int64_t B, const double* e, const double* logp, const double* shift_dev, int64_t n_global,
                       double* sums4, double* est3, void* workspace) {
  FF_CHECK(B > 0 && e && logp && shift_dev && sums4 && workspace && n_global >= 0 && (n_global == 0 || est3), FF_EINVAL,
           "ff_energy_estimate: bad argument");
  const unsigned nb = (unsigned)((B + FF_EST_SEG - 1) / FF_EST_SEG);
  double* part = (double*)workspace + 1;       // [0]: the counter (zero between calls) | partial sums
  FF_LAUNCH(ff_energy_estimate_kernel, nb, FF_RBLOCK(FF_EST_THREADS), stream, B, e, logp, shift_dev, (double)n_global, sums4, est3, part,
            (unsigned*)workspace);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_energy_finish(void* stream, const double* sums4, const double* shift_dev, int64_t n_global, double* est3) {
  FF_CHECK(sums4 && shift_dev && est3 && n_global > 0, FF_EINVAL, "ff_energy_finish: bad argument");
  FF_LAUNCH(ff_energy_finish_kernel, 1, FF_RBLOCK(64), stream, sums4, shift_dev, (double)n_global, est3);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

size_t ff_beta_buffer_doubles(int nstates) { return nstates > 0 ? 2 + (size_t)nstates * FF_SS_K * 4 : 0; }

int ff_beta_state_partials(void* stream, int64_t B, int nstates, const int32_t* walker_state, const double* e, const double* logp,
                           double* buf) {
  FF_CHECK(B >= 0 && nstates > 0 && buf && (B == 0 || (walker_state && e && logp)), FF_EINVAL, "ff_beta_state_partials: bad argument");
  FF_LAUNCH(ff_state_part_kernel, (unsigned)(nstates * FF_SS_K), FF_RBLOCK(256), stream, B, walker_state, e, logp, buf + 2);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_beta_finish(void* stream, const double* buf, const double* shift_dev, const double* logits, int nstates, double beta,
                   int64_t n_global, double* est8, double* gphi, double* mean_e, double* logp_all) {
  FF_CHECK(buf && shift_dev && logits && nstates > 0 && beta > 0.0 && n_global > 0 && est8 && gphi && mean_e && logp_all, FF_EINVAL,
           "ff_beta_finish: bad argument");
  FF_LAUNCH(ff_beta_finish_kernel, 1, FF_RBLOCK(256), stream, buf, shift_dev, logits, nstates, beta, (double)n_global, est8, gphi, mean_e,
            logp_all);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

int ff_state_sums(void* stream, int64_t B, int nstates, const int32_t* walker_state, const double* e, double* sums, double* counts) {
  FF_CHECK(B >= 0 && nstates > 0 && sums && counts && (B == 0 || (walker_state && e)), FF_EINVAL, "ff_state_sums: bad argument");
  FF_LAUNCH(ff_state_sums_kernel, (unsigned)nstates, FF_RBLOCK(256), stream, B, walker_state, e, sums, counts);
  FF_LAUNCH_CHECK();
  return FF_OK;
}

}  // extern "C"
