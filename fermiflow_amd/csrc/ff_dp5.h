// ff_dp5.h -- the Dormand-Prince 5(4) stage machine of the one-walker-per-workgroup ("wide") kernels, written once.
//
// Same rules as every fused integrator of this library (ff_ode.h: Hairer initial step or a warm start, RMS error norm over
// the walker's own state, 0.9 err^-1/5 clamped to [0.2, 10], predictive bound, no growth right after a rejection) and the
// same 5-vector storage as ff_ode_fwd_kernel: y, c0..c2 hold k0..k2 up to stage 3; once k3 is known the inputs of stages
// 4, 5, the candidate y_new and the error accumulator c3 are formed and overwrite them.  An accepted step takes k6 as
// the next k0 (FSAL); a rejected one re-evaluates f(y) in a "stage 0".
// In these kernels ONE walker occupies the whole workgroup, so every decision is workgroup-uniform: no wave votes.
//
// Stage index s: -2 f(y) at the start, -1 the probe of the initial-step heuristic, 0 f(y) after a rejection, 1..6 the
// stages of a step (6 = f(y_new), the error estimate and the accept / reject decision).
#pragma once
#include "ff_ode.h"

// coefficients of the stage input  in = gy y + g0 c0 + g1 c1 + g2 c2  (one expression for all stages)
FF_D void ff_dp5_coeffs(int s, double h, double h0dir, double& gy, double& g0, double& g1, double& g2) {
  gy = 1.0; g0 = 0.0; g1 = 0.0; g2 = 0.0;
  switch (s) {
    case -1: g0 = h0dir; break;
    case 1: g0 = h * FF_A10; break;
    case 2: g0 = h * FF_A20; g1 = h * FF_A21; break;
    case 3: g0 = h * FF_A30; g1 = h * FF_A31; g2 = h * FF_A32; break;
    case 4: gy = 0.0; g0 = 1.0; break;
    case 5: gy = 0.0; g1 = 1.0; break;
    case 6: gy = 0.0; g2 = 1.0; break;
    default: break;   // -2, 0: the state itself
  }
}

struct ff_dp5_ctl {
  double rtol, atol, nt_inv;   // tolerances; 1 / number of state components in the RMS norm
  double hwarm;                // warm start: first step size to try (<= 0: cold start)
  int max_steps;
  double h0v, d1v, hmax_acc;   // scratch of the initial-step heuristic; largest accepted step
};

// Consumes out = f(stage input of stage s).  wgt(v): weight of component v in the error norm; gsum(part): the sum of a
// per-lane partial over the workgroup (identical on all lanes).  Returns the next stage, or 99 when the walker is done.
// c3 is any indexable per-lane vector (registers, or lane-private LDS columns).
template <int NV, class C3, class W, class G>
FF_D int ff_dp5_consume(int s, ff_stepper& S, ff_dp5_ctl& C, double* y, double* c0, double* c1, double* c2, C3& c3,
                        const double* out, W wgt, G gsum) {
  const double h = S.h, rtol = C.rtol, atol = C.atol;
  if (s == -2) {
#pragma unroll
    for (int v = 0; v < NV; v++) c0[v] = out[v];
    double p0 = 0.0, p1 = 0.0;
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const double isc = wgt(v) * ff_rcp(fma(fabs(y[v]), rtol, atol));
      p0 = fma(y[v] * isc, y[v] * isc, p0);
      p1 = fma(c0[v] * isc, c0[v] * isc, p1);
    }
    const double d0 = sqrt(gsum(p0) * C.nt_inv);
    C.d1v = sqrt(gsum(p1) * C.nt_inv);
    C.h0v = S.h0(d0, C.d1v);
    if (C.hwarm > 0.0) {   // the walker brings its own first step: no probe evaluation
      S.habs = fmin(C.hwarm, S.interval);
      S.plan();
      return S.done ? 99 : 1;
    }
    return S.done ? 99 : -1;
  }
  if (s == -1) {
    double p2 = 0.0;
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const double t = (out[v] - c0[v]) * wgt(v) * ff_rcp(fma(fabs(y[v]), rtol, atol));
      p2 = fma(t, t, p2);
    }
    const double d2 = sqrt(gsum(p2) * C.nt_inv) / C.h0v;
    S.init_habs(C.h0v, C.d1v, d2);
    S.plan();
    return 1;
  }
  if (s == 0) {
#pragma unroll
    for (int v = 0; v < NV; v++) c0[v] = out[v];
    return 1;
  }
  if (s == 1) {
#pragma unroll
    for (int v = 0; v < NV; v++) c1[v] = out[v];
    return 2;
  }
  if (s == 2) {
#pragma unroll
    for (int v = 0; v < NV; v++) c2[v] = out[v];
    return 3;
  }
  if (s == 3) {
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const double k0v = c0[v], k1v = c1[v], k2v = c2[v], k3v = out[v], yv = y[v];
      c0[v] = fma(h, FF_A40 * k0v + FF_A41 * k1v + FF_A42 * k2v + FF_A43 * k3v, yv);
      c1[v] = fma(h, FF_A50 * k0v + FF_A51 * k1v + FF_A52 * k2v + FF_A53 * k3v, yv);
      c2[v] = fma(h, FF_B0 * k0v + FF_B2 * k2v + FF_B3 * k3v, yv);
      c3[v] = h * (FF_E0 * k0v + FF_E2 * k2v + FF_E3 * k3v);
    }
    return 4;
  }
  if (s == 4) {
#pragma unroll
    for (int v = 0; v < NV; v++) {
      c1[v] = fma(h * FF_A54, out[v], c1[v]);
      c2[v] = fma(h * FF_B4, out[v], c2[v]);
      c3[v] = fma(h * FF_E4, out[v], c3[v]);
    }
    return 5;
  }
  if (s == 5) {
#pragma unroll
    for (int v = 0; v < NV; v++) {
      c2[v] = fma(h * FF_B5, out[v], c2[v]);
      c3[v] = fma(h * FF_E5, out[v], c3[v]);
    }
    return 6;
  }
  // s == 6: out = f(y_new), c2 = the candidate y_new
  double pe = 0.0;
#pragma unroll
  for (int v = 0; v < NV; v++) {
    const double e = fma(h * FF_E6, out[v], c3[v]);
    const double t = e * wgt(v) * ff_rcp(fma(fmax(fabs(y[v]), fabs(c2[v])), rtol, atol));
    pe = fma(t, t, pe);
  }
  const double err = sqrt(gsum(pe) * C.nt_inv);
  const bool acc = S.decide(err, C.max_steps);
  if (acc) {
    C.hmax_acc = fmax(C.hmax_acc, fabs(h));
#pragma unroll
    for (int v = 0; v < NV; v++) { y[v] = c2[v]; c0[v] = out[v]; }
  }
  S.plan();
  if (S.done) return 99;
  return acc ? 1 : 0;
}
