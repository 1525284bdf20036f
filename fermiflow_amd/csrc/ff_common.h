// ff_common.h -- shared definitions for the FermiFlow MI355X (gfx950) kernels.
//
// The kernels are written for 64-lane wavefronts and one wave per workgroup (blockDim.x == 64), so a
// workgroup barrier is a single-wave s_barrier and walkers with different adaptive step counts never
// hold another wave back.
//
// FF_HOSTSIM: a TEST-ONLY build mode (tests/hostsim/) in which the very same kernel sources are compiled
// with g++ and each workgroup is run by host threads with a real barrier.  It exists because the
// build container has no GPU; it is never built by __graft_entry__.build(), never loaded by the
// fermiflow_amd package, and is not a fallback path.  This include switch is the ONLY place the product
// sources know about it: tests/hostsim/hip_shim.h emulates the HIP subset and the gfx950 builtins the
// kernels use (and pre-defines the few FF_* macros below that carry an #ifndef).
#pragma once
#include <stdint.h>
#include <math.h>
#include "../../include/fermiflow.h"

#define FF_WAVE 64
#define FF_MAX_ORB 36      // HO2D().orbitals has 36 entries (src/orbitals.py:81)
#define FF_MAX_NS 12       // largest single-spin determinant handled natively
#define FF_HMAX 256        // hidden width supported by the fused ODE kernels (reference default: 50; --Deta/--Dmu, src/FermionHO2D.py:24-27)
#define FF_HPAD (FF_HMAX + 8)  // LDS weight table length: zero-padded so unrolled unit loops may overrun H

#ifdef FF_HOSTSIM
#include "../../tests/hostsim/hip_shim.h"
#else
#include <hip/hip_runtime.h>
#define FF_LAUNCH(kernel, grid, block, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (hipStream_t)(stream), __VA_ARGS__)
#define FF_LAUNCH_LDS(kernel, grid, block, lds_bytes, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (lds_bytes), (hipStream_t)(stream), __VA_ARGS__)
#define FF_DYN_LDS(name) extern __shared__ double name[]
#endif

// Barrier among the lanes of ONE wave, for workgroups whose waves run independently of each other (the two-wave tabulated
// adjoint): LDS operations of a wave execute in program order, so all it takes is that the compiler keeps them in order and
// that they have completed -- a workgroup-scope fence (s_waitcnt), no s_barrier.  In a single-wave workgroup this is
// __syncthreads().
#ifndef FF_WAVE_SYNC
#define FF_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); \
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#endif
// Barrier of a SINGLE-WAVE workgroup.  The LDS serves the operations of one wave in program order -- a store followed by a load of the
// same word from another lane of that wave needs no wait in between (the load's own s_waitcnt in front of its first use is all the
// hardware asks for) --, so what remains of __syncthreads() is the promise that the compiler keeps the operations in order: a
// wavefront-scope fence (no instruction) and a wave barrier (no instruction).  __syncthreads() itself costs s_waitcnt lgkmcnt(0) --
// the wave parks until its last STORE has completed -- at every one of the ten phase boundaries of an evaluation.
// -DFF_WG1_FULL_SYNC restores __syncthreads() (A/B).
// FF_WAVE_ORDER(): the same promise among the lanes of one wave of a multi-wave workgroup whose waves run independently (where
// FF_WAVE_SYNC() also waits for completion: needed only where ANOTHER wave is told that this wave's operations have landed).
#ifndef FF_WAVE_ORDER
#ifdef FF_WG1_FULL_SYNC
#define FF_WAVE_ORDER() FF_WAVE_SYNC()
#else
#define FF_WAVE_ORDER() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                             __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); } while (0)
#endif
#endif
#ifndef FF_WG1_SYNC
#ifdef FF_WG1_FULL_SYNC
#define FF_WG1_SYNC() __syncthreads()
#else
#define FF_WG1_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                           __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); } while (0)
#endif
#endif
#ifndef FF_HAVE_WAVE_BALLOT
#define ff_wave_ballot(pred) __ballot(pred)      // (on the GPU a ballot IS per wave; the host simulator's needs to know which wave)
#endif

// Stage of a Dormand-Prince step as a compile-time constant (the kernels' evaluate() bodies are instantiated once per stage 1 .. 6,
// straight-line code between them, and once for the rare evaluations in front of a step, where the stage is a run-time value)
#ifndef FF_ASSUME
#define FF_ASSUME(x) __builtin_assume(x)
#endif
#define FF_STAGE_DYN 99
template <int V> struct ff_stage_c { static constexpr int value = V; };

// the lane's index within its wave, recomputed where it is needed (two instructions) instead of kept in a register across a loop --
// in a single-wave workgroup this IS threadIdx.x, which lives in an input register the allocator can only spill
#ifndef FF_LANE_SELF
#define FF_LANE_SELF() ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
#endif
// a value the program knows to be wave-uniform -> scalar register (lets loops on it be scalar loops)
#define FF_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)

// block size of the reduction kernels (blockDim-generic trees)
#ifndef FF_RBLOCK
#define FF_RBLOCK(n) (n)
#endif

// An integer the compiler must treat as freshly computed here: blocks the hoisting of everything derived from it (LDS
// addresses, decoded indices) out of the enclosing loop.  hipcc otherwise precomputes dozens of loop-invariant addresses
// in the prologue of the fused ODE kernels and spills them -- recomputing them costs one or two integer ops each.
#ifndef FF_OPAQUE
#define FF_OPAQUE(x) asm volatile("" : "+v"(x))
#endif

#define FF_D __device__ __forceinline__
#define FF_HD __host__ __device__ __forceinline__

// status codes of the C ABI
#define FF_OK 0
#define FF_EINVAL 1        // invalid argument (null pointer, non-positive size, ...)
#define FF_EUNSUPPORTED 2  // configuration has no native kernel instantiation
#define FF_ELAUNCH 3       // HIP launch/runtime failure

// --- reciprocal: v_rcp_f64 + two Newton steps (full double precision for normal inputs) ----------
FF_D double ff_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}

// r = sqrt(r2) and 1/r from one v_rsq_f64 + two Newton steps (r within ~1 ulp; r2 = 0 gives r = 0, 1/r = inf)
FF_D void ff_sqrt_rcp(double r2, double& r, double& ri) {
  double y = __builtin_amdgcn_rsq(r2);
  const double hx = 0.5 * r2;
  y = fma(fma(-hx * y, y, 0.5), y, y);
  y = fma(fma(-hx * y, y, 0.5), y, y);
  ri = y;
  r = r2 > 0.0 ? r2 * y : 0.0;
}

// value held by the neighbouring lane (lane ^ 1): one DPP quad permutation per 32-bit half, no LDS
FF_D double ff_swap1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// value held by lane ^ 2 of the same quad (DPP quad_perm [2,3,0,1])
FF_D double ff_swap2(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// value of v on lane src (any lane of the wave; two ds_bpermute_b32: the LDS crossbar, no LDS memory)
FF_D double ff_lane_read(double v, int src) {
#ifdef FF_DIAG_NO_BPERMUTE      // (timing diagnostic: no LDS crossbar round trip; the numbers are then wrong)
  return v;
#endif
  const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products per wave, D = A B + C, one f64 per lane and operand.
// Lane l = 16 k + 4 blk + i supplies A_blk[i][k]; lane 16 k + 4 blk + j supplies B_blk[k][j]; lane 16 i + 4 blk + j holds
// C/D_blk[i][j] (measured on gfx950: tools/probes/mfma_f64.hip) -- the block is (l / 4) % 4, NOT l / 16.
FF_D double ff_mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
// element (c, r) of the lane's 4 x 4 block delivered to lane (r, c) (l = 16 r + 4 blk + c; tl = 16 c + 4 blk + r): the value fed as the
// A operand is read as A_blk[c][r], so the product with the identity (idn = 1 on the lanes r == c) comes back transposed
FF_D double ff_block_transpose(double v, double idn, int tl) {
#ifdef FF_TRANSPOSE_BPERMUTE      // (A/B: two ds_bpermute_b32 through the LDS crossbar)
  return ff_lane_read(v, tl);
#else
  return ff_mfma4(v, idn, 0.0);
#endif
}

// v_mfma_f64_16x16x4_f64: D (16x16) = A (16x4) B (4x16) + C per wave.  Lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16];
// register v of lane l holds C/D[4 v + l / 16][l % 16] (measured on gfx950: tools/probes/mfma_f64.hip, wide_probe.hip).
// 64 cycles per instruction = 16 FMA per cycle and SIMD: the fp64 peak of the vector pipe, in one issue slot.
typedef double ff_d4 __attribute__((vector_size(32)));
FF_D ff_d4 ff_mfma16(double a, double b, ff_d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
typedef float ff_f4 __attribute__((vector_size(16)));      // accumulator of v_mfma_f32_16x16x4_f32: register v of lane l holds C/D[4 (l / 16) + v][l % 16]

// --- exp for |x| <= 708.  Rounding and scaling use the integer pipe instead of the quarter-rate
//     v_rndne_f64 / v_cvt_i32_f64 / v_ldexp_f64: adding 1.5*2^52 leaves round(x*log2e) in the low mantissa word, and
//     2^k is applied by adding k to the exponent field (the polynomial value is in [0.7,1.42], |k| <= 1010: always normal).
FF_D double ff_scale2(double p, int k) {
  return __hiloint2double(__double2hiint(p) + (k << 20), __double2loint(p));
}
FF_D double ff_exp(double x) {
  const double L2E = 1.4426950408889634074, LN2H = 6.93147180369123816490e-01, LN2L = 1.90821492927058770002e-10;
  const double MAGIC = 6755399441055744.0;
  const double t = fma(x, L2E, MAGIC);
  const int k = __double2loint(t);
  const double kf = t - MAGIC;
  double r = fma(-kf, LN2H, x);
  r = fma(-kf, LN2L, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ff_scale2(p, k);
}

// sigmoid(a) = 1/(1+exp(-a))  (torch.nn.Sigmoid, src/MLP.py:16)
FF_D double ff_sigmoid(double a) {
  a = fmin(fmax(a, -700.0), 700.0);
  return ff_rcp(1.0 + ff_exp(-a));
}

// --- table-driven variant for kernels with >= 2 waves per SIMD (the LDS look-up sits in the dependent chain):
//     x = (64 m + j) ln2/64 + r, |r| <= ln2/128, exp(x) = 2^m * 2^(j/64) * (1 + r + ... + r^5/120);
//     tab[j] = 2^(j/64) lives in LDS (ff_fill_exp2_table).  Relative error ~1.5e-16.
FF_D void ff_fill_exp2_table(double* tab, int lane) {
  if (lane < 64) tab[lane] = exp2((double)lane * (1.0 / 64.0));
}
FF_D double ff_exp_tab(double x, const double* __restrict__ tab) {
  const double INV = 92.332482616893656 /* 64/ln2 */, C_HI = 0.010830424667801708 /* ln2/64, low 24 bits clear */,
               C_LO = 2.8447437476627285e-11, MAGIC = 6755399441055744.0;
  const double t = fma(x, INV, MAGIC);
  const int k = __double2loint(t);
  const double kf = t - MAGIC;
  double r = fma(-kf, C_HI, x);
  r = fma(-kf, C_LO, r);
  double p = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ff_scale2(tab[k & 63] * p, k >> 6);
}
template <bool TAB>
FF_D double ff_sigmoid_sel(double a, const double* __restrict__ tab) {
  a = fmin(fmax(a, -700.0), 700.0);
  return ff_rcp(1.0 + (TAB ? ff_exp_tab(-a, tab) : ff_exp(-a)));
}

// --- NV sigmoids side by side.  Written step by step over the NV lanes-of-work with scheduling fences in between:
//     under the register pressure of the local-energy kernel hipcc otherwise runs the NV dependency chains one
//     after the other, and a single resident wave per SIMD then stalls on every fp64 latency.
#define FF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// wave priority for the instruction arbiter (0..3): the small kernels of the critical path raise theirs, because they run beside the
// prefetched Metropolis kernel.  Round 4 measured this as zero-sum (the adjoint stage 0.576 -> 0.532 ms, the wait for the walkers 0.040
// -> 0.070: somebody waited for the sampler either way) and left it off.  Since round 5 the sampler works two iterations ahead and
// nobody waits for it: what it takes from the kernels beside it is simply lost (ff_dep_contract_kernel: 48 us beside it, 14 alone),
// so they go first.  Default 3; -DFF_PRIO=0 switches it off.
#ifndef FF_PRIO
#define FF_PRIO 3
#endif
#ifdef FF_HOSTSIM
#define FF_SETPRIO() do {} while (0)
#else
#define FF_SETPRIO() __builtin_amdgcn_s_setprio(FF_PRIO)
#endif
template <int NV, bool TAB>
FF_D void ff_sigmoid_n(const double* a_in, double* sg, const double* __restrict__ tab) {
  const double MAGIC = 6755399441055744.0;
  double x[NV], t[NV], r[NV], p[NV];
  int k[NV];
#pragma unroll
  for (int q = 0; q < NV; q++) x[q] = -fmin(fmax(a_in[q], -700.0), 700.0);
  if (TAB) {
    const double INV = 92.332482616893656, C_HI = 0.010830424667801708, C_LO = 2.8447437476627285e-11;
#pragma unroll
    for (int q = 0; q < NV; q++) t[q] = fma(x[q], INV, MAGIC);
    FF_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < NV; q++) { k[q] = __double2loint(t[q]); t[q] -= MAGIC; }
#pragma unroll
    for (int q = 0; q < NV; q++) r[q] = fma(-t[q], C_HI, x[q]);
    FF_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < NV; q++) { r[q] = fma(-t[q], C_LO, r[q]); t[q] = tab[k[q] & 63]; }
    FF_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < NV; q++) p[q] = fma(r[q], 1.0 / 120.0, 1.0 / 24.0);
    FF_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < NV; q++) p[q] = fma(p[q], r[q], 1.0 / 6.0);
    FF_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < NV; q++) p[q] = fma(p[q], r[q], 0.5);
    FF_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < NV; q++) p[q] = fma(p[q], r[q], 1.0);
    FF_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < NV; q++) p[q] = fma(p[q], r[q], 1.0);
    FF_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < NV; q++) p[q] = ff_scale2(t[q] * p[q], k[q] >> 6);
  } else {
    const double L2E = 1.4426950408889634074, LN2H = 6.93147180369123816490e-01, LN2L = 1.90821492927058770002e-10;
    const double C[14] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0,
                          1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0};
#pragma unroll
    for (int q = 0; q < NV; q++) t[q] = fma(x[q], L2E, MAGIC);
    FF_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < NV; q++) { k[q] = __double2loint(t[q]); t[q] -= MAGIC; }
#pragma unroll
    for (int q = 0; q < NV; q++) r[q] = fma(-t[q], LN2H, x[q]);
    FF_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < NV; q++) { r[q] = fma(-t[q], LN2L, r[q]); p[q] = fma(C[0], r[q], C[1]); }
#pragma unroll
    for (int c = 2; c < 14; c++) {
      FF_SCHED_FENCE();
#pragma unroll
      for (int q = 0; q < NV; q++) p[q] = fma(p[q], r[q], C[c]);
    }
    FF_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < NV; q++) p[q] = ff_scale2(p[q], k[q]);
  }
  FF_SCHED_FENCE();
#pragma unroll
  for (int q = 0; q < NV; q++) p[q] += 1.0;
  FF_SCHED_FENCE();
#pragma unroll
  for (int q = 0; q < NV; q++) t[q] = __builtin_amdgcn_rcp(p[q]);
  FF_SCHED_FENCE();
#pragma unroll
  for (int q = 0; q < NV; q++) r[q] = fma(-p[q], t[q], 1.0);
  FF_SCHED_FENCE();
#pragma unroll
  for (int q = 0; q < NV; q++) t[q] = fma(r[q], t[q], t[q]);
  FF_SCHED_FENCE();
#pragma unroll
  for (int q = 0; q < NV; q++) r[q] = fma(-p[q], t[q], 1.0);
  FF_SCHED_FENCE();
#pragma unroll
  for (int q = 0; q < NV; q++) sg[q] = fma(r[q], t[q], t[q]);
  FF_SCHED_FENCE();
}

// Optional per-walker inputs (ff_ode.walker_order / walker_h_init / walker_class) are read WITHOUT a branch: a lane that
// has no entry reads element 0 of `safe`, an array of the launch that always exists, and discards it.  This is not a
// micro-optimisation: ROCm 7.2's register allocator was caught placing VGPR->AGPR copies in front of the exec restore at the
// join of exactly these `if (p) v = p[b];` blocks (tools/check_agpr_spills.py, docs/LOG.md round 2) -- with a null pointer nobody
// takes the branch, the copy executes with exec = 0 and the value it should have parked is garbage from then on.
template <class T>
FF_D T ff_opt_load(const T* p, bool cond, long long idx, const void* safe, T dflt) {
  const bool use = cond && p != nullptr;
  const T* q = use ? p + idx : reinterpret_cast<const T*>(safe);
  const T v = *q;
  return use ? v : dflt;
}

// --- log(x) for the Metropolis kernel (Box-Muller radius, log|det|): exponent split + atanh series
//     log m = 2 s (1 + z/3 + ... + z^10/21), s = (m-1)/(m+1), z = s^2, m in [sqrt(1/2), sqrt(2)); ~1 ulp.
//     Zero, denormal, negative and non-finite arguments take the library routine.
FF_D double ff_log(double x) {
  if (!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308)) return log(x);
  int hi = __double2hiint(x);
  int e = (hi >> 20) - 1023;
  double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, __double2loint(x));   // [1, 2)
  if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
  const double s = (m - 1.0) * ff_rcp(m + 1.0), z = s * s;
  double p = 1.0 / 21.0;
  p = fma(p, z, 1.0 / 19.0);
  p = fma(p, z, 1.0 / 17.0);
  p = fma(p, z, 1.0 / 15.0);
  p = fma(p, z, 1.0 / 13.0);
  p = fma(p, z, 1.0 / 11.0);
  p = fma(p, z, 1.0 / 9.0);
  p = fma(p, z, 1.0 / 7.0);
  p = fma(p, z, 1.0 / 5.0);
  p = fma(p, z, 1.0 / 3.0);
  p = fma(p, z, 1.0);
  const double ef = (double)e;
  return fma(ef, 6.93147180369123816490e-01, fma(2.0 * s, p, ef * 1.90821492927058770002e-10));
}

// --- sin(pi t), cos(pi t) for t in [0, 2): quadrant split + Taylor on |pi r| <= pi/4
FF_D void ff_sincospi(double t, double* sn, double* cs) {
  const double q = rint(t * 2.0);            // nearest multiple of 1/2
  const double r = fma(q, -0.5, t) * 3.14159265358979323846;
  const double r2 = r * r;
  double ps = -1.0 / 1307674368000.0;        // -1/15!
  ps = fma(ps, r2, 1.0 / 6227020800.0);
  ps = fma(ps, r2, -1.0 / 39916800.0);
  ps = fma(ps, r2, 1.0 / 362880.0);
  ps = fma(ps, r2, -1.0 / 5040.0);
  ps = fma(ps, r2, 1.0 / 120.0);
  ps = fma(ps, r2, -1.0 / 6.0);
  ps = fma(ps * r2, r, r);
  double pc = 1.0 / 20922789888000.0;        // 1/16!
  pc = fma(pc, r2, -1.0 / 87178291200.0);
  pc = fma(pc, r2, 1.0 / 479001600.0);
  pc = fma(pc, r2, -1.0 / 3628800.0);
  pc = fma(pc, r2, 1.0 / 40320.0);
  pc = fma(pc, r2, -1.0 / 720.0);
  pc = fma(pc, r2, 1.0 / 24.0);
  pc = fma(pc, r2, -0.5);
  pc = fma(pc, r2, 1.0);
  const int k = ((int)q) & 3;                // angle = k*pi/2 + pi r
  const double s0 = (k & 1) ? pc : ps, c0 = (k & 1) ? ps : pc;
  *sn = (k & 2) ? -s0 : s0;
  *cs = ((k + 1) & 2) ? -c0 : c0;
}

// number of (i<j) pairs before row i for n particles; pair index of (i,j), i<j
FF_HD int ff_pair_index(int n, int i, int j) { return i * (2 * n - i - 1) / 2 + (j - i - 1); }
