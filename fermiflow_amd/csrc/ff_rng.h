// ff_rng.h -- counter-based Philox4x32-10 + Box-Muller for the throughput-mode MCMC.
// (The reference draws from torch's global generator, src/base_dist.py:62,65,68; bit-parity with it is
// only possible by feeding its noise explicitly -- ff_mcmc_sample_noise.  This generator gives every
// (walker, step, particle pair) its own counter, so results do not depend on how walkers are sharded.)
#pragma once
#include "ff_common.h"

struct ff_u4 { uint32_t x, y, z, w; };

FF_D ff_u4 ff_philox(uint64_t key, uint64_t c01, uint32_t c2, uint32_t c3) {
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  uint32_t c0 = (uint32_t)c01, c1 = (uint32_t)(c01 >> 32);
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  ff_u4 o = {c0, c1, c2, c3};
  return o;
}

FF_D uint64_t ff_bits53(uint32_t hi, uint32_t lo) { return (((uint64_t)hi << 32) | lo) >> 11; }

// two independent N(0,1) from one Philox block
FF_D void ff_normal_pair(uint64_t key, uint64_t walker, uint32_t step, uint32_t slot, double& z0, double& z1) {
  ff_u4 r = ff_philox(key, walker, step, slot);
  double u1 = (double)(ff_bits53(r.x, r.y) + 1) * 1.1102230246251565e-16;  // (0,1]
  double u2 = (double)ff_bits53(r.z, r.w) * 1.1102230246251565e-16;        // [0,1)
  double rad = sqrt(-2.0 * ff_log(u1));
  double s, c;
  ff_sincospi(2.0 * u2, &s, &c);
  z0 = rad * c;
  z1 = rad * s;
}

// four independent N(0,1) from one Philox block: Box-Muller on two pairs of 32-bit uniforms (u1 in (0,1] and u2 in
// [0,1) on a 2^-32 grid: the normal's tail ends at 6.7 sigma and the angle grid maps onto itself under z -> -z, so the
// Metropolis proposal stays exactly symmetric).  Slot `quad` of a (walker, step) serves particles 2*quad and 2*quad+1.
FF_D void ff_normal_quad(uint64_t key, uint64_t walker, uint32_t step, uint32_t quad, double* z) {
  ff_u4 r = ff_philox(key, walker, step, quad);
  const double s32 = 2.3283064365386963e-10;   // 2^-32
  const double ua = ((double)r.x + 1.0) * s32, ub = (double)r.y * s32;
  const double uc = ((double)r.z + 1.0) * s32, ud = (double)r.w * s32;
  const double ra = sqrt(-2.0 * ff_log(ua)), rc = sqrt(-2.0 * ff_log(uc));
  double sn, cs;
  ff_sincospi(2.0 * ub, &sn, &cs);
  z[0] = ra * cs; z[1] = ra * sn;
  ff_sincospi(2.0 * ud, &sn, &cs);
  z[2] = rc * cs; z[3] = rc * sn;
}

// uniform in [0,1) (torch.rand_like semantics)
FF_D double ff_uniform(uint64_t key, uint64_t walker, uint32_t step, uint32_t slot) {
  ff_u4 r = ff_philox(key, walker, step, slot);
  return (double)ff_bits53(r.x, r.y) * 1.1102230246251565e-16;
}
