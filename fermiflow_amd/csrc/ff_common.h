// ff_common.h -- shared definitions for the FermiFlow MI355X (gfx950) kernels.
//
// The kernels are written for 64-lane wavefronts and one wave per workgroup (blockDim.x == 64), so a
// workgroup barrier is a single-wave s_barrier and walkers with different adaptive step counts never
// hold another wave back.
//
// FF_HOSTSIM: a TEST-ONLY build mode (tests/hostsim/) in which the very same kernel sources are compiled
// with g++ and each workgroup is run by 64 host threads with a real barrier.  It exists because the
// build container has no GPU; it is never built by __graft_entry__.build(), never loaded by the
// fermiflow_amd package, and is not a fallback path.
#pragma once
#include <stdint.h>
#include <math.h>
#include "../../include/fermiflow.h"

#define FF_WAVE 64
#define FF_MAX_ORB 36      // HO2D().orbitals has 36 entries (src/orbitals.py:81)
#define FF_MAX_NS 12       // largest single-spin determinant handled natively
#define FF_HMAX 64         // hidden width supported by the fused ODE kernels (reference default: 50)

#ifdef FF_HOSTSIM
#include "../../tests/hostsim/hip_shim.h"
#else
#include <hip/hip_runtime.h>
#define FF_LAUNCH(kernel, grid, block, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (hipStream_t)(stream), __VA_ARGS__)
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
#ifdef FF_HOSTSIM
  return 1.0 / x;
#else
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
#endif
}

// --- exp for |x| <= 708: Cody-Waite reduction + degree-13 Taylor/Horner, ~1 ulp ----------------------
FF_D double ff_exp(double x) {
  const double L2E = 1.4426950408889634074, LN2H = 6.93147180369123816490e-01, LN2L = 1.90821492927058770002e-10;
  double k = rint(x * L2E);
  double r = fma(-k, LN2H, x);
  r = fma(-k, LN2L, r);
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
  return ldexp(p, (int)k);
}

// sigmoid(a) = 1/(1+exp(-a))  (torch.nn.Sigmoid, src/MLP.py:16)
FF_D double ff_sigmoid(double a) {
  a = fmin(fmax(a, -700.0), 700.0);
  return ff_rcp(1.0 + ff_exp(-a));
}

// number of (i<j) pairs before row i for n particles; pair index of (i,j), i<j
FF_HD int ff_pair_index(int n, int i, int j) { return i * (2 * n - i - 1) / 2 + (j - i - 1); }
