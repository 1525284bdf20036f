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

// ---------------------------------------------------------------------------------------------------------------------
// Two-segment variant for the matrix-core local-energy kernel: segment A (NA components of type TA -- double, or float for
// the single-precision sensitivity path -- all with the error weight wA; its error accumulator c3A is any indexable vector)
// and segment B (NB double components with weights wgtB(v)).  Same stage machine, same decisions: the arithmetic of segment A
// runs in TA, its error norms are accumulated in TA and added to segment B's in double.
FF_D double ff_t_abs(double x) { return __builtin_fabs(x); }
FF_D float ff_t_abs(float x) { return __builtin_fabsf(x); }
FF_D double ff_t_max(double a, double b) { return __builtin_fmax(a, b); }
FF_D float ff_t_max(float a, float b) { return __builtin_fmaxf(a, b); }
FF_D double ff_t_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
FF_D float ff_t_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
FF_D double ff_t_rcp(double x) { return ff_rcp(x); }
FF_D float ff_t_rcp(float x) { return __builtin_amdgcn_rcpf(x); }      // v_rcp_f32, 1 ulp: error norms only

// PIN: the caller lays the stages out one behind the other (a compile-time s): the updated vectors are pinned where they are computed
// -- the compiler otherwise sinks them into the stages that use them and keeps k0 .. k3 alive through the right-hand sides in between.
template <int NA, class TA, class C3A, int NB, class C3B, class YA, class WB, class G, bool PIN = false>
FF_D int ff_dp5_consume2(int s, ff_stepper& S, ff_dp5_ctl& C, YA& yA, TA* c0A, TA* c1A, TA* c2A, C3A& c3A, const TA* outA, double wA_,
                         double* yB, double* c0B, double* c1B, double* c2B, C3B& c3B, const double* outB, WB wgtB, G gsum) {
  const double h = S.h, rtol = C.rtol, atol = C.atol;
  const TA hA = (TA)h, rtA = (TA)rtol, atA = (TA)atol, wA = (TA)wA_;
  constexpr bool ACC0 = C3A::in_lds;      // from stage 4 on the error accumulator lives in c0 (see there)
  if (s == -2) {
    TA qa0 = 0, qa1 = 0;
#pragma unroll
    for (int v = 0; v < NA; v++) {
      c0A[v] = outA[v];
      const TA isc = wA * ff_t_rcp(ff_t_fma(ff_t_abs(yA[v]), rtA, atA));
      qa0 = ff_t_fma(yA[v] * isc, yA[v] * isc, qa0);
      qa1 = ff_t_fma(c0A[v] * isc, c0A[v] * isc, qa1);
    }
    double p0 = (double)qa0, p1 = (double)qa1;
#pragma unroll
    for (int v = 0; v < NB; v++) {
      c0B[v] = outB[v];
      const double isc = wgtB(v) * ff_rcp(fma(fabs(yB[v]), rtol, atol));
      p0 = fma(yB[v] * isc, yB[v] * isc, p0);
      p1 = fma(c0B[v] * isc, c0B[v] * isc, p1);
    }
    const double d0 = sqrt(gsum(p0) * C.nt_inv);
    C.d1v = sqrt(gsum(p1) * C.nt_inv);
    C.h0v = S.h0(d0, C.d1v);
    if (C.hwarm > 0.0) {
      S.habs = fmin(C.hwarm, S.interval);
      S.plan();
      return S.done ? 99 : 1;
    }
    return S.done ? 99 : -1;
  }
  if (s == -1) {
    TA qa = 0;
#pragma unroll
    for (int v = 0; v < NA; v++) {
      const TA t = (outA[v] - c0A[v]) * wA * ff_t_rcp(ff_t_fma(ff_t_abs(yA[v]), rtA, atA));
      qa = ff_t_fma(t, t, qa);
    }
    double p2 = (double)qa;
#pragma unroll
    for (int v = 0; v < NB; v++) {
      const double t = (outB[v] - c0B[v]) * wgtB(v) * ff_rcp(fma(fabs(yB[v]), rtol, atol));
      p2 = fma(t, t, p2);
    }
    const double d2 = sqrt(gsum(p2) * C.nt_inv) / C.h0v;
    S.init_habs(C.h0v, C.d1v, d2);
    S.plan();
    return 1;
  }
  if (s == 0) {      // (three separate copies: a pointer selected at run time would force the register arrays into scratch)
#pragma unroll
    for (int v = 0; v < NA; v++) c0A[v] = outA[v];
#pragma unroll
    for (int v = 0; v < NB; v++) c0B[v] = outB[v];
    return 1;
  }
  if (s == 1) {
#pragma unroll
    for (int v = 0; v < NA; v++) c1A[v] = outA[v];
#pragma unroll
    for (int v = 0; v < NB; v++) c1B[v] = outB[v];
    return 2;
  }
  if (s == 2) {
#pragma unroll
    for (int v = 0; v < NA; v++) c2A[v] = outA[v];
#pragma unroll
    for (int v = 0; v < NB; v++) c2B[v] = outB[v];
    return 3;
  }
  if (s == 3) {
#pragma unroll
    for (int v = 0; v < NA; v++) {
      const TA k0v = c0A[v], k1v = c1A[v], k2v = c2A[v], k3v = outA[v], yv = yA[v];
      c0A[v] = ff_t_fma(hA, (TA)FF_A40 * k0v + (TA)FF_A41 * k1v + (TA)FF_A42 * k2v + (TA)FF_A43 * k3v, yv);
      c1A[v] = ff_t_fma(hA, (TA)FF_A50 * k0v + (TA)FF_A51 * k1v + (TA)FF_A52 * k2v + (TA)FF_A53 * k3v, yv);
      c2A[v] = ff_t_fma(hA, (TA)FF_B0 * k0v + (TA)FF_B2 * k2v + (TA)FF_B3 * k3v, yv);
      c3A[v] = hA * ((TA)FF_E0 * k0v + (TA)FF_E2 * k2v + (TA)FF_E3 * k3v);
      if constexpr (PIN) { FF_OPAQUE(c0A[v]); FF_OPAQUE(c1A[v]); FF_OPAQUE(c2A[v]); }
    }
#pragma unroll
    for (int v = 0; v < NB; v++) {
      const double k0v = c0B[v], k1v = c1B[v], k2v = c2B[v], k3v = outB[v], yv = yB[v];
      c0B[v] = fma(h, FF_A40 * k0v + FF_A41 * k1v + FF_A42 * k2v + FF_A43 * k3v, yv);
      c1B[v] = fma(h, FF_A50 * k0v + FF_A51 * k1v + FF_A52 * k2v + FF_A53 * k3v, yv);
      c2B[v] = fma(h, FF_B0 * k0v + FF_B2 * k2v + FF_B3 * k3v, yv);
      c3B[v] = h * (FF_E0 * k0v + FF_E2 * k2v + FF_E3 * k3v);
      if constexpr (PIN) { FF_OPAQUE(c0B[v]); FF_OPAQUE(c1B[v]); FF_OPAQUE(c2B[v]); }
    }
    return 4;
  }
  if (s == 4) {
#pragma unroll
    for (int v = 0; v < NA; v++) {
      c1A[v] = ff_t_fma(hA * (TA)FF_A54, outA[v], c1A[v]);
      c2A[v] = ff_t_fma(hA * (TA)FF_B4, outA[v], c2A[v]);
      // c0 -- the input of this stage -- is free from here on: where c3 lives in lane-private LDS columns the error accumulator moves
      // into it (one write at stage 3 and one read here instead of a read-modify-write per stage: -1.4 % of the pass at 12 particles;
      // with c3 in registers the longer life of c0 only costs -- the fp32 kernel of configs[4] spilled 92 B and lost 1.2 %)
      if constexpr (ACC0) c0A[v] = ff_t_fma(hA * (TA)FF_E4, outA[v], (TA)c3A[v]);
      else c3A[v] = ff_t_fma(hA * (TA)FF_E4, outA[v], (TA)c3A[v]);
      if constexpr (PIN) { FF_OPAQUE(c1A[v]); FF_OPAQUE(c2A[v]); if constexpr (ACC0) FF_OPAQUE(c0A[v]); }
    }
#pragma unroll
    for (int v = 0; v < NB; v++) {
      c1B[v] = fma(h * FF_A54, outB[v], c1B[v]);
      c2B[v] = fma(h * FF_B4, outB[v], c2B[v]);
      if constexpr (ACC0) c0B[v] = fma(h * FF_E4, outB[v], c3B[v]);
      else c3B[v] = fma(h * FF_E4, outB[v], c3B[v]);
      if constexpr (PIN) { FF_OPAQUE(c1B[v]); FF_OPAQUE(c2B[v]); if constexpr (ACC0) FF_OPAQUE(c0B[v]); }
    }
    return 5;
  }
  if (s == 5) {
#pragma unroll
    for (int v = 0; v < NA; v++) {
      c2A[v] = ff_t_fma(hA * (TA)FF_B5, outA[v], c2A[v]);
      if constexpr (ACC0) c0A[v] = ff_t_fma(hA * (TA)FF_E5, outA[v], c0A[v]);
      else c3A[v] = ff_t_fma(hA * (TA)FF_E5, outA[v], (TA)c3A[v]);
      if constexpr (PIN) { FF_OPAQUE(c2A[v]); if constexpr (ACC0) FF_OPAQUE(c0A[v]); }
    }
#pragma unroll
    for (int v = 0; v < NB; v++) {
      c2B[v] = fma(h * FF_B5, outB[v], c2B[v]);
      if constexpr (ACC0) c0B[v] = fma(h * FF_E5, outB[v], c0B[v]);
      else c3B[v] = fma(h * FF_E5, outB[v], c3B[v]);
      if constexpr (PIN) { FF_OPAQUE(c2B[v]); if constexpr (ACC0) FF_OPAQUE(c0B[v]); }
    }
    return 6;
  }
  // s == 6
  TA qa = 0;
#pragma unroll
  for (int v = 0; v < NA; v++) {
    const TA e = ff_t_fma(hA * (TA)FF_E6, outA[v], ACC0 ? c0A[v] : (TA)c3A[v]);
    const TA t = e * wA * ff_t_rcp(ff_t_fma(ff_t_max(ff_t_abs(yA[v]), ff_t_abs(c2A[v])), rtA, atA));
    qa = ff_t_fma(t, t, qa);
  }
  double pe = (double)qa;
#pragma unroll
  for (int v = 0; v < NB; v++) {
    const double e = fma(h * FF_E6, outB[v], ACC0 ? c0B[v] : (double)c3B[v]);
    const double t = e * wgtB(v) * ff_rcp(fma(fmax(fabs(yB[v]), fabs(c2B[v])), rtol, atol));
    pe = fma(t, t, pe);
  }
  const double err = sqrt(gsum(pe) * C.nt_inv);
  const bool acc = S.decide(err, C.max_steps);
  if (acc) {
    C.hmax_acc = fmax(C.hmax_acc, fabs(h));
#pragma unroll
    for (int v = 0; v < NA; v++) { yA[v] = c2A[v]; c0A[v] = outA[v]; }
#pragma unroll
    for (int v = 0; v < NB; v++) { yB[v] = c2B[v]; c0B[v] = outB[v]; }
  }
  S.plan();
  if (S.done) return 99;
  return acc ? 1 : 0;
}
