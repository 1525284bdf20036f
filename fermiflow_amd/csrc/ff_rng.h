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

// Two N(0,1) from two 32-bit words (Box-Muller) on the hardware fp32 transcendentals -- v_log_f32 (log2), v_sqrt_f32,
// v_sin_f32 / v_cos_f32 (argument in revolutions) -- promoted to fp64.  A Metropolis chain only needs its proposal to be
// SYMMETRIC to be exact, and this one is by construction: the radius uniform (a + 1/2) 2^-32 lies in (0, 1] (no infinite
// proposal; the largest a round to u1 = 1.0f exactly: rad = sqrt(-0.0) = -0.0 and both normals are signed zeros -- harmless),
// the angle covers the first quadrant only (30 bits of b) and each normal takes its sign from a bit of b of its
// own, so g and -g (indeed any sign pattern) are equally likely bit for bit whatever the rounding of the transcendentals.
// 24-bit normals with the tail to 6.7 sigma; the fp64 chain (log, sqrt, sincospi: ~90 fp64 instructions per pair) was a
// third of the Metropolis kernel.  ff_rng_fill materialises exactly these values.
FF_D void ff_normal_pair32(uint32_t a, uint32_t b, double& z0, double& z1) {
  const float u1 = fmaf((float)a, 2.3283064365386963e-10f, 1.1641532182693481e-10f);              // (a + 1/2) 2^-32
  const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));      // sqrt(-2 ln u1)
  const float phi = (float)(b >> 2) * 2.3283064365386963e-10f;                                     // [0, 1/4] revolutions
  const float c = rad * __builtin_amdgcn_cosf(phi), s = rad * __builtin_amdgcn_sinf(phi);
  z0 = (double)__uint_as_float(__float_as_uint(c) ^ (b << 31));
  z1 = (double)__uint_as_float(__float_as_uint(s) ^ ((b << 30) & 0x80000000u));
}

// four independent N(0,1) from one Philox block.  Slot `quad` of a (walker, step) serves particles 2*quad and 2*quad+1
// (d = 2; coordinates 4*quad .. 4*quad+3 of the walker in general).
FF_D void ff_normal_quad(uint64_t key, uint64_t walker, uint32_t step, uint32_t quad, double* z) {
  ff_u4 r = ff_philox(key, walker, step, quad);
  ff_normal_pair32(r.x, r.y, z[0], z[1]);
  ff_normal_pair32(r.z, r.w, z[2], z[3]);
}

// uniform in [0,1) (torch.rand_like semantics)
FF_D double ff_uniform_words(uint32_t hi, uint32_t lo) { return (double)ff_bits53(hi, lo) * 1.1102230246251565e-16; }
FF_D double ff_uniform(uint64_t key, uint64_t walker, uint32_t step, uint32_t slot) {
  ff_u4 r = ff_philox(key, walker, step, slot);
  return ff_uniform_words(r.x, r.y);
}
