// ff_ode.h -- shared pieces of the fused CNF integrators (ff_cnf_fwd.hip, ff_cnf_adj.hip).
//
// Mapping (N particles, D dims, M = N*D coordinates): a 64-lane wave holds G = 64/M walkers; lane (g,i)
// owns coordinate i of walker g (and, for the local-energy kernel, the sensitivity column d z / d x_i).
// One workgroup = one wave, so __syncthreads() is a single-wave barrier and walkers with different adaptive
// step counts never wait for another wave.  Workgroups are persistent and stride over walker groups.
//
// Solver: Dormand-Prince 5(4) with per-walker step control (Hairer initial step; RMS error norm over the
// walker's own state, tolerance atol + rtol*max(|y|,|y_new|); factor 0.9*err^-1/5 clamped to [0.2,10], no
// growth right after a rejection; last step clipped to the end point).  These are the rules of the
// reference's scipy-RK45 backend (src/NeuralODE/nnModule.py:49-61) applied per walker instead of to the
// whole flattened batch -- SURVEY.md finding 3: E_loc is insensitive to the step sequence (1e-10 relative).
#pragma once
#include "ff_common.h"

template <int N, int D>
struct ff_geom {
  static constexpr int M = N * D;          // coordinates per walker
  static constexpr int G = FF_WAVE / M > 16 ? 16 : FF_WAVE / M;    // walkers per wave (at most 16: the radius ids carry 4 bits of it)
  static constexpr int P = N * (N - 1) / 2;  // electron pairs
  static constexpr int R = P + N;          // radii per walker: pairs then one-body
  static constexpr int RA = R > 0 ? R : 1;
};

// A per-lane vector of NV doubles that lives either in registers or in a lane-private LDS column (slot-major
// [NV][64] layout: conflict-free).  Indexing syntax is the same, so the integrator code does not care.
template <int NV, bool IN_LDS>
struct ff_lane_vec;
template <int NV>
struct ff_lane_vec<NV, false> {
  double r[NV];
  FF_D ff_lane_vec(double*, int) {}
  FF_D double& operator[](int v) { return r[v]; }
  FF_D const double& operator[](int v) const { return r[v]; }
};
template <int NV>
struct ff_lane_vec<NV, true> {
  double* col;
  FF_D ff_lane_vec(double* base, int lane) : col(base + lane) {}
  FF_D double& operator[](int v) { return col[v * FF_WAVE]; }
  FF_D const double& operator[](int v) const { return col[v * FF_WAVE]; }
};

// (a, b) of the p-th pair in the order a < b, a-major: compile-time table for statically unrolled sweeps
template <int N>
struct ff_pair_table {
  int a[N * (N - 1) / 2 + 1], b[N * (N - 1) / 2 + 1];
  constexpr ff_pair_table() : a(), b() {
    int p = 0;
    for (int i = 0; i < N; i++)
      for (int j = i + 1; j < N; j++) { a[p] = i; b[p] = j; p++; }
  }
};

// per-hidden-unit weight record staged in LDS (48 B, 16-B aligned -> three ds_read_b128)
// ff_ode.walker_cost (include/fermiflow.h): attempted steps + max(0, -log2(r_min^2) - 1), clamped to [0, 32)
FF_D int ff_cost_class(int steps, double rmin) {
  int e = 0;
  const double r2 = rmin * rmin;
  if (r2 > 0.0 && r2 < 1e300) (void)frexp(r2, &e); else e = r2 > 0.0 ? 2 : -64;
  const int rc = -e - 1 > 0 ? -e - 1 : 0;
  const int c = steps + rc;
  return c < 0 ? 0 : (c > 31 ? 31 : c);
}

struct __attribute__((aligned(16))) ff_wtab { double w1, b1, w2, w2w1, w2w1_2, w2w1_3; };

FF_D void ff_load_weights(ff_wtab (*s_w)[FF_HPAD], const ff_net& net, int lane) {
  for (int h = lane; h < FF_HPAD; h += FF_WAVE) {
    ff_wtab e = {0, 0, 0, 0, 0, 0}, m = {0, 0, 0, 0, 0, 0};
    if (h < net.He) {
      e.w1 = net.ew1[h]; e.b1 = net.eb1[h]; e.w2 = net.ew2[h];
      e.w2w1 = e.w2 * e.w1; e.w2w1_2 = e.w2w1 * e.w1; e.w2w1_3 = e.w2w1_2 * e.w1;
    }
    if (h < net.Hm) {
      m.w1 = net.mw1[h]; m.b1 = net.mb1[h]; m.w2 = net.mw2[h];
      m.w2w1 = m.w2 * m.w1; m.w2w1_2 = m.w2w1 * m.w1; m.w2w1_3 = m.w2w1_2 * m.w1;
    }
    s_w[0][h] = e; s_w[1][h] = m;
  }
}

// f(r) = sum_h w2 sigma(w1 r + b1) and its first NH-1 derivatives (MLP.forward / .grad, src/MLP.py:30-45,
// extended analytically: sigma' = s(1-s), sigma'' = s'(1-2s), sigma''' = s'(1-6s')).
// The exp/rcp chain of one hidden unit is ~45 dependent fp64 instructions; four units are evaluated side by
// side so that a single resident wave per SIMD still has independent work to issue every cycle.  The weight
// table is zero-padded to FF_HMAX, so running the loop to the next multiple of 4 adds exact zeros.
#define FF_HU 5
template <int NH, bool TAB>
FF_D void ff_heads(const ff_wtab* __restrict__ tab, const double* __restrict__ e2, int H, double r, double* hd) {
  double h0[FF_HU], h1[FF_HU], h2[FF_HU], h3[FF_HU];
#pragma unroll
  for (int q = 0; q < FF_HU; q++) h0[q] = h1[q] = h2[q] = h3[q] = 0.0;
  for (int h = 0; h < H; h += FF_HU) {
    ff_wtab w[FF_HU];
    double a[FF_HU], s[FF_HU];
#pragma unroll
    for (int q = 0; q < FF_HU; q++) { w[q] = tab[h + q]; a[q] = fma(w[q].w1, r, w[q].b1); }
    ff_sigmoid_n<FF_HU, TAB>(a, s, e2);
#pragma unroll
    for (int q = 0; q < FF_HU; q++) {
      h0[q] = fma(w[q].w2, s[q], h0[q]);
      if (NH >= 2) {
        const double s1 = s[q] * (1.0 - s[q]);
        h1[q] = fma(w[q].w2w1, s1, h1[q]);
        if (NH >= 3) h2[q] = fma(w[q].w2w1_2, s1 * fma(-2.0, s[q], 1.0), h2[q]);
        if (NH >= 4) h3[q] = fma(w[q].w2w1_3, s1 * fma(-6.0, s1, 1.0), h3[q]);
      }
    }
  }
  double t0 = h0[0], t1 = h1[0], t2 = h2[0], t3 = h3[0];
#pragma unroll
  for (int q = 1; q < FF_HU; q++) { t0 += h0[q]; t1 += h1[q]; t2 += h2[q]; t3 += h3[q]; }
  hd[0] = t0;
  if (NH >= 2) hd[1] = t1;
  if (NH >= 3) hd[2] = t2;
  if (NH >= 4) hd[3] = t3;
}

// Dormand-Prince tableau
#define FF_A10 (1.0 / 5)
#define FF_A20 (3.0 / 40)
#define FF_A21 (9.0 / 40)
#define FF_A30 (44.0 / 45)
#define FF_A31 (-56.0 / 15)
#define FF_A32 (32.0 / 9)
#define FF_A40 (19372.0 / 6561)
#define FF_A41 (-25360.0 / 2187)
#define FF_A42 (64448.0 / 6561)
#define FF_A43 (-212.0 / 729)
#define FF_A50 (9017.0 / 3168)
#define FF_A51 (-355.0 / 33)
#define FF_A52 (46732.0 / 5247)
#define FF_A53 (49.0 / 176)
#define FF_A54 (-5103.0 / 18656)
#define FF_B0 (35.0 / 384)
#define FF_B2 (500.0 / 1113)
#define FF_B3 (125.0 / 192)
#define FF_B4 (-2187.0 / 6784)
#define FF_B5 (11.0 / 84)
#define FF_E0 (-71.0 / 57600)
#define FF_E2 (71.0 / 16695)
#define FF_E3 (-71.0 / 1920)
#define FF_E4 (17253.0 / 339200)
#define FF_E5 (-22.0 / 525)
#define FF_E6 (1.0 / 40)

// x^(+-1/5) for the step-size controller: single-precision log/exp hardware ops (3 instructions instead of
// ~200 for the fp64 pow); 1e-6 relative accuracy is irrelevant for a step-size factor, and every lane of a
// walker's group computes the identical value.
FF_D double ff_pow02(double x, float e) {
  // v_log_f32 / v_exp_f32 (base 2, ~1e-7 relative): six instructions where the library powf is ~90
  return (double)__builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf((float)x));
}

// Wave-wide OR of per-lane flag words (bit 0: still integrating, bit 1: just rejected a step); wave-uniform result.
// The step a warm-started walker opens with (ff_ode.walker_h_init x walker_h_scale) -- with ff_ode.walker_h_equal rounded DOWN to
// interval / k, the equal steps that cover the interval in as many steps as it would: two steps of 1/2 pass where 0.42 + 0.58 needs the
// second one to grow by 1.4 x, which the step-size control only grants after an error below 0.11 of the tolerance (flow and adjoint
// kernels; the local-energy pass gets its opening steps rounded by ff_walker_schedule).
// (Selects, not a per-lane branch: a divergent join in a walker prologue is where ROCm 7.2's register allocator misplaces its AGPR
// copies -- docs/LOG.md, round 2 -- and `if (h > 0 && h < interval)` written as a branch did break ff_wide_adjtab_kernel<3, 4, .> at 20 particles:
// every walker failed, with walker_h_equal = 0.  The flag itself is uniform.)
FF_D double ff_open_step(double h, double ta, double tb, int equal) {
  if (equal) {
    const double interval = fabs(tb - ta);
    const bool round = h > 0.0 && h < interval;
    const double hs = round ? h : interval;
    const double r = interval / ceil(interval / hs - 1e-9);
    h = round ? r : h;
  }
  return h;
}

// Two ballots -- no LDS traffic, no barrier (64 lanes OR-ing into one LDS word serialise: that cost ~10 % of the
// local-energy kernel).
FF_D int ff_wave_or(int* s_any, int lane, int flags) {
  const unsigned long long b1 = __ballot((flags & 1) != 0), b2 = __ballot((flags & 2) != 0);
  return (b1 ? 1 : 0) | (b2 ? 2 : 0);
}

// the same for workgroups whose waves do not run in lockstep (ff_wave_ballot: the caller's own wave)
FF_D int ff_wave_or_w(int* s_any, int lane, int flags) {
  const unsigned long long b1 = ff_wave_ballot((flags & 1) != 0), b2 = ff_wave_ballot((flags & 2) != 0);
  return (b1 ? 1 : 0) | (b2 ? 2 : 0);
}

#ifndef FF_STEP_TRACE
#define FF_STEP_TRACE(t, h, err, acc) do { } while (0)
#endif
// per-walker step-size bookkeeping (identical on all lanes of a walker's group)
struct ff_stepper {
  double t, tb, dir, interval, habs, h, tnew;
  double hprev, eprev;   // size and error norm of the previous accepted step (0: none yet)
  int nacc, nrej, natt, rejected, fail;
  bool done;
  FF_D void begin(double ta_, double tb_, bool active) {
    t = ta_; tb = tb_; dir = tb_ > ta_ ? 1.0 : -1.0; interval = fabs(tb_ - ta_);
    habs = 0.0; h = 0.0; tnew = ta_; nacc = nrej = natt = rejected = fail = 0; hprev = eprev = 0.0;
    done = !active || interval == 0.0;
  }
  // Hairer initial step, part 1 (scipy select_initial_step)
  FF_D double h0(double d0, double d1) const {
    double v = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    return fmin(v, interval);
  }
  FF_D void init_habs(double h0v, double d1, double d2) {
    double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0v * 1e-3) : ff_pow02(0.01 / fmax(d1, d2), 0.2f);
    habs = fmin(fmin(100.0 * h0v, h1), interval);
  }
  FF_D void plan() {  // choose h for the next attempt
    if (done) { h = 0.0; tnew = t; return; }
    tnew = t + habs * dir;
    if (dir * (tnew - tb) > 0.0) tnew = tb;
    h = tnew - t;
    habs = fabs(h);
  }
  // returns true if the attempt is accepted
  FF_D bool decide(double err, int max_steps) {
    if (done) return false;
    natt++;
    bool acc = err < 1.0;
    FF_STEP_TRACE(t, h, err, acc);   // (a no-op; tests/hostsim/hip_shim.h prints every step decision when built with -DFF_HOSTSIM_TRACE)
    if (acc) {
      double f = (err == 0.0) ? 10.0 : fmin(10.0, 0.9 * ff_pow02(err, -0.2f));
#ifndef FF_NO_PREDICTIVE
      // Predictive bound (Gustafsson; Hairer & Wanner II, IV.8): the elementary rule assumes the error coefficient
      // C = err / h^5 of the next step equals this step's.  Where a trajectory runs towards a point at which the field is
      // only C^1 (a particle passing the origin: mu(|x|) x), C grows by a constant factor per step, the elementary rule
      // proposes a step that fails, and every accepted step is followed by a rejected one (7 wasted evaluations each).
      // Extrapolating the trend, C_next = C_n^2 / C_(n-1), gives h_next = h_el * (h_n / h_(n-1)) * (err_(n-1) / err_n)^(1/5);
      // taken only where it is the SMALLER of the two, so no step is ever larger than the elementary rule's.
      if (eprev > 0.0 && err > 0.0) f = fmax(0.2, fmin(f, f * (habs / hprev) * ff_pow02(eprev / err, 0.2f)));
      hprev = habs; eprev = err;
#endif
      if (rejected) f = fmin(1.0, f);
      habs *= f; t = tnew; rejected = 0; nacc++;
      if (dir * (t - tb) >= 0.0) done = true;
    } else {
      if (!(err == err)) { fail = 1; done = true; }
      else { habs *= fmax(0.2, 0.9 * ff_pow02(err, -0.2f)); rejected = 1; nrej++; }
    }
    if (!done && natt >= max_steps) { fail = 1; done = true; }
    return acc;
  }
};
