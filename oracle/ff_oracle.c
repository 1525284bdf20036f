/* ff_oracle.c -- ORACLE: CPU restatement of FermiFlow's VMC hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; fermiflow_amd/ never does.
 *
 * Parity status: PINNED, d = 2 and d = 3.  Checked (tests/test_oracle_golden.py) against golden vectors produced by
 * importing the reference itself (tests/golden/make_golden.py, groups g1..g6; g7: the reference's dimension-generic
 * code on (B, n, 3) walkers with 3-D orbitals that are products of its own HO2D closures) and against the
 * reference's own known-answer tests (tests/test_basedist.py:5-129: E_loc == sum of orbital energies).
 *
 * Third-party arithmetic that is not in /root/reference: the ODE solver.  The reference calls
 * torchdiffeq.odeint (dopri5; unpinned, .travis.yml:8; absent here) or, through its own second
 * backend, scipy.integrate.solve_ivp RK45 (src/NeuralODE/nnModule.py:49-61; scipy 1.15.3 is what the
 * golden vectors were produced with).  ffo_rk45() below restates scipy's published RK45 algorithm
 * (Dormand-Prince 5(4), Hairer initial step, RMS error norm over the whole flattened state, step
 * factor clamp [0.2, 10], safety 0.9, last step clipped to t_bound).
 *
 * Derivatives (grad / Laplacian of log p, needed by the local energy, src/VMC.py:48-49 via
 * src/utils.py:40-65) are obtained by propagating 2nd-order truncated Taylor numbers ("jets")
 * through exactly the arithmetic the reference runs -- forward-mode AD, the mathematical equivalent
 * of the reference's nested reverse-mode passes -- so this file contains no hand-derived
 * derivative formulas for the wavefunction or the flow.  (The parameter-gradient of the adjoint
 * solve, src/NeuralODE/nnModule.py:104-133, is the one place where d/dtheta of the MLP is written
 * out explicitly.)
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "ff_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define FFO_MAXN 24
#define FFO_MAXM 72

#define JN(x) x##_o0
#define FFO_ORDER 0
#include "ffo_core.inc"
#undef JN
#undef FFO_ORDER
#define JN(x) x##_o1
#define FFO_ORDER 1
#include "ffo_core.inc"
#undef JN
#undef FFO_ORDER
#define JN(x) x##_o2
#define FFO_ORDER 2
#include "ffo_core.inc"
#undef JN
#undef FFO_ORDER

int ffo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* =============================================================================================
 * scipy.integrate RK45 restated (scipy/integrate/_ivp/rk.py RK45 + common.py select_initial_step).
 * ============================================================================================= */
typedef void (*ffo_rhs)(void* ctx, double t, const double* y, double* dy);

static double rms(const double* a, const double* scale, int n) {
  double s = 0.0;
  for (int i = 0; i < n; i++) { double q = a[i] / scale[i]; s += q * q; }
  return sqrt(s) / sqrt((double)n);
}

static const double RK_C[6] = {0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1};
static const double RK_A[6][5] = {
    {0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
static const double RK_B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
static const double RK_E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};

static int ffo_rk45(ffo_rhs f, void* ctx, double t0, double t1, double* y, int n, double rtol, double atol,
                    int* nfev_out, int* nstep_out) {
  int nfev = 0, nstep = 0;
  if (n == 0 || t0 == t1) { if (nfev_out) *nfev_out = 0; if (nstep_out) *nstep_out = 0; return 0; }
  double* buf = (double*)malloc(sizeof(double) * (size_t)n * 11);
  if (!buf) return -1;
  double* K[7]; for (int s = 0; s < 7; s++) K[s] = buf + (size_t)s * n;
  double *ys = buf + (size_t)7 * n, *ynew = buf + (size_t)8 * n, *scale = buf + (size_t)9 * n, *tmp = buf + (size_t)10 * n;
  double dir = t1 > t0 ? 1.0 : -1.0, t = t0;
  f(ctx, t, y, K[0]); nfev++;
  /* select_initial_step, order = error_estimator_order = 4 */
  double h_abs;
  {
    double interval = fabs(t1 - t0);
    for (int i = 0; i < n; i++) scale[i] = atol + fabs(y[i]) * rtol;
    double d0 = rms(y, scale, n), d1 = rms(K[0], scale, n), h0;
    if (d0 < 1e-5 || d1 < 1e-5) h0 = 1e-6; else h0 = 0.01 * d0 / d1;
    if (h0 > interval) h0 = interval;
    for (int i = 0; i < n; i++) ys[i] = y[i] + h0 * dir * K[0][i];
    f(ctx, t0 + h0 * dir, ys, K[1]); nfev++;
    for (int i = 0; i < n; i++) tmp[i] = K[1][i] - K[0][i];
    double d2 = rms(tmp, scale, n) / h0, h1;
    if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
    else h1 = pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
    h_abs = fmin(fmin(100 * h0, h1), interval);
  }
  int status = 0;
  while (dir * (t - t1) < 0) {
    double min_step = 10.0 * fabs(nextafter(t, dir * INFINITY) - t);
    if (h_abs < min_step) h_abs = min_step;
    int accepted = 0, rejected = 0;
    double t_new = t, h = 0;
    while (!accepted) {
      if (h_abs < min_step) { status = -2; goto done; }
      h = h_abs * dir; t_new = t + h;
      if (dir * (t_new - t1) > 0) t_new = t1;
      h = t_new - t; h_abs = fabs(h);
      for (int s = 1; s < 6; s++) {
        for (int i = 0; i < n; i++) {
          double dy = 0.0;
          for (int j = 0; j < s; j++) dy += K[j][i] * RK_A[s][j];
          ys[i] = y[i] + dy * h;
        }
        f(ctx, t + RK_C[s] * h, ys, K[s]); nfev++;
      }
      for (int i = 0; i < n; i++) {
        double dy = 0.0;
        for (int j = 0; j < 6; j++) dy += K[j][i] * RK_B[j];
        ynew[i] = y[i] + h * dy;
      }
      f(ctx, t + h, ynew, K[6]); nfev++;
      for (int i = 0; i < n; i++) {
        scale[i] = atol + fmax(fabs(y[i]), fabs(ynew[i])) * rtol;
        double e = 0.0;
        for (int j = 0; j < 7; j++) e += K[j][i] * RK_E[j];
        tmp[i] = e * h;
      }
      double err = rms(tmp, scale, n);
      if (err < 1.0) {
        double factor = err == 0.0 ? 10.0 : fmin(10.0, 0.9 * pow(err, -0.2));
        if (rejected && factor > 1.0) factor = 1.0;
        h_abs *= factor; accepted = 1;
      } else {
        h_abs *= fmax(0.2, 0.9 * pow(err, -0.2)); rejected = 1;
      }
      if (!(err == err)) { status = -3; goto done; }   /* NaN */
    }
    t = t_new; nstep++;
    memcpy(y, ynew, sizeof(double) * n);
    memcpy(K[0], K[6], sizeof(double) * n);
  }
done:
  free(buf);
  if (nfev_out) *nfev_out = nfev;
  if (nstep_out) *nstep_out = nstep;
  return status;
}

/* =============================================================================================
 * orbitals / Slater / log_prob
 * ============================================================================================= */
int ffo_orbitals(const int* k, int nk, const double* pts, int npts, double* out) {
  for (int a = 0; a < nk; a++) {
    if (k[a] < 0 || k[a] >= 36) return 1;
    for (int p = 0; p < npts; p++)
      out[a * npts + p] = orbital2d_o0(k[a], jc_o0(pts[2 * p]), jc_o0(pts[2 * p + 1])).v;
  }
  return 0;
}

int ffo_orbitals3d(const int* k, int nk, const double* pts, int npts, double* out) {
  for (int a = 0; a < nk; a++) {
    if (k[a] < 0 || k[a] >= 120) return 1;
    for (int p = 0; p < npts; p++)
      out[a * npts + p] = orbital3d_o0(k[a], jc_o0(pts[3 * p]), jc_o0(pts[3 * p + 1]), jc_o0(pts[3 * p + 2])).v;
  }
  return 0;
}

static const int* row(const int* table, const int* wstate, int64_t b, int n) {
  return table + (size_t)(wstate ? wstate[b] : 0) * n;
}

/* value, gradient and Laplacian of logp0 = 2*(log|det up| + log|det dn|) at one point */
static void logprob_glap_d(int nup, int ndn, int d, const int* ou, const int* od, const double* x,
                           double* logp, double* grad, double* lap) {
  int n = nup + ndn, M = d * n;
  if (!grad && !lap) {
    jet_o0 xy[FFO_MAXM];
    for (int i = 0; i < M; i++) xy[i] = jc_o0(x[i]);
    *logp = logprob_d_o0(nup, ndn, d, ou, od, xy).v;
    return;
  }
  double l = 0.0;
  for (int i = 0; i < M; i++) {
    jet_o2 xy[FFO_MAXM];
    for (int k = 0; k < M; k++) xy[k] = jmk_o2(x[k], k == i ? 1.0 : 0.0, 0.0);
    jet_o2 r = logprob_d_o2(nup, ndn, d, ou, od, xy);
    if (logp) *logp = r.v;
    if (grad) grad[i] = r.d1;
    l += r.d2;
  }
  if (lap) *lap = l;
}
static void logprob_glap(int nup, int ndn, const int* ou, const int* od, const double* x,
                         double* logp, double* grad, double* lap) { logprob_glap_d(nup, ndn, 2, ou, od, x, logp, grad, lap); }

static int logprob_any(int64_t B, int nup, int ndn, int d, const int* tab_up, const int* tab_dn, const int* wstate,
                       const double* x, double* logp, double* grad, double* lap) {
  int n = nup + ndn;
  if (n <= 0 || nup > FFO_MAXN || ndn > FFO_MAXN || d * n > FFO_MAXM) return 1;
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; b++)
    logprob_glap_d(nup, ndn, d, nup ? row(tab_up, wstate, b, nup) : NULL, ndn ? row(tab_dn, wstate, b, ndn) : NULL,
                   x + b * d * n, logp + b, grad ? grad + b * d * n : NULL, lap ? lap + b : NULL);
  return 0;
}
int ffo_logprob(int64_t B, int nup, int ndn, const int* tab_up, const int* tab_dn, const int* wstate,
                const double* x, double* logp, double* grad, double* lap) {
  return logprob_any(B, nup, ndn, 2, tab_up, tab_dn, wstate, x, logp, grad, lap);
}
int ffo_logprob3d(int64_t B, int nup, int ndn, const int* tab_up, const int* tab_dn, const int* wstate,
                  const double* x, double* logp, double* grad, double* lap) {
  return logprob_any(B, nup, ndn, 3, tab_up, tab_dn, wstate, x, logp, grad, lap);
}

/* FreeFermion.sample loop body (src/base_dist.py:62-70) with explicit noise. */
static int mcmc_noise_any(int64_t B, int nup, int ndn, int d, const int* tab_up, const int* tab_dn, const int* wstate,
                          int steps, double tau, const double* g0, const double* g, const double* u,
                          double* x_out, double* logp_out, uint8_t* accept_out);
int ffo_mcmc_noise(int64_t B, int nup, int ndn, const int* tab_up, const int* tab_dn, const int* wstate,
                   int steps, double tau, const double* g0, const double* g, const double* u,
                   double* x_out, double* logp_out, uint8_t* accept_out) {
  return mcmc_noise_any(B, nup, ndn, 2, tab_up, tab_dn, wstate, steps, tau, g0, g, u, x_out, logp_out, accept_out);
}
int ffo_mcmc_noise3d(int64_t B, int nup, int ndn, const int* tab_up, const int* tab_dn, const int* wstate,
                     int steps, double tau, const double* g0, const double* g, const double* u,
                     double* x_out, double* logp_out, uint8_t* accept_out) {
  return mcmc_noise_any(B, nup, ndn, 3, tab_up, tab_dn, wstate, steps, tau, g0, g, u, x_out, logp_out, accept_out);
}
static int mcmc_noise_any(int64_t B, int nup, int ndn, int d, const int* tab_up, const int* tab_dn, const int* wstate,
                          int steps, double tau, const double* g0, const double* g, const double* u,
                          double* x_out, double* logp_out, uint8_t* accept_out) {
  int n = nup + ndn, M = d * n;
  if (n <= 0 || M > FFO_MAXM) return 1;
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; b++) {
    const int* ou = nup ? row(tab_up, wstate, b, nup) : NULL;
    const int* od = ndn ? row(tab_dn, wstate, b, ndn) : NULL;
    double x[FFO_MAXM], nx[FFO_MAXM], logp, nl;
    for (int i = 0; i < M; i++) x[i] = g0[b * M + i];
    logprob_glap_d(nup, ndn, d, ou, od, x, &logp, NULL, NULL);
    for (int s = 0; s < steps; s++) {
      const double* gs = g + ((size_t)s * B + b) * M;
      for (int i = 0; i < M; i++) { double t = tau * gs[i]; nx[i] = x[i] + t; }
      logprob_glap_d(nup, ndn, d, ou, od, nx, &nl, NULL, NULL);
      double p = exp(nl - logp);
      int acc = u[(size_t)s * B + b] < p;
      if (acc) { memcpy(x, nx, sizeof(double) * M); logp = nl; }
      if (accept_out) accept_out[(size_t)s * B + b] = (uint8_t)acc;
    }
    memcpy(x_out + b * M, x, sizeof(double) * M);
    if (logp_out) logp_out[b] = logp;
  }
  return 0;
}

/* =============================================================================================
 * backflow, potentials
 * ============================================================================================= */
int ffo_backflow(int64_t B, int n, int d, const ffo_net* net, const double* x, double* v, double* div) {
  if (n * d > FFO_MAXM || d > 3) return 1;
  int M = n * d;
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; b++) {
    jet_o0 z[FFO_MAXM], f[FFO_MAXM], dv;
    for (int i = 0; i < M; i++) z[i] = jc_o0(x[b * M + i]);
    backflow_o0(net, n, d, z, v ? f : NULL, div ? &dv : NULL);
    if (v) for (int i = 0; i < M; i++) v[b * M + i] = f[i].v;
    if (div) div[b] = dv.v;
  }
  return 0;
}

/* HO.V = 0.5*sum r^2 (src/potentials.py:13); Coulomb V = sum_{i<j} Z/|r_i-r_j| (:23-47). */
int ffo_potential(int64_t B, int n, int d, double Z, int use_ho, const double* x, double* V) {
  int M = n * d;
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; b++) {
    const double* xb = x + b * M;
    double pair = 0.0, ho = 0.0;
    for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++) {
        double r2 = 0.0;
        for (int c = 0; c < d; c++) { double t = xb[i * d + c] - xb[j * d + c]; r2 += t * t; }
        pair += Z / sqrt(r2);
      }
    for (int i = 0; i < M; i++) ho += xb[i] * xb[i];
    V[b] = pair + (use_ho ? 0.5 * ho : 0.0);
  }
  return 0;
}

/* =============================================================================================
 * CNF: generate / delta_logp with the reference's batch-global step control
 * (src/flow.py:42-55 -> src/NeuralODE/nnModule.py:49-61: the whole batch is ONE flat ODE state).
 * ============================================================================================= */
typedef struct { const ffo_net* net; int64_t B; int n, d; int with_logp; } batch_ctx;

static void rhs_batch(void* vctx, double t, const double* y, double* dy) {
  (void)t;
  batch_ctx* c = (batch_ctx*)vctx;
  int M = c->n * c->d;
  int64_t B = c->B;
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; b++) {
    jet_o0 z[FFO_MAXM], f[FFO_MAXM], dv;
    for (int i = 0; i < M; i++) z[i] = jc_o0(y[b * M + i]);
    backflow_o0(c->net, c->n, c->d, z, f, c->with_logp ? &dv : NULL);
    for (int i = 0; i < M; i++) dy[b * M + i] = f[i].v;
    if (c->with_logp) dy[B * M + b] = -dv.v;       /* flatten order: (x, logp) as torch.cat (NeuralODE/utils.py:4-5) */
  }
}

int ffo_cnf_generate(int64_t B, int n, int d, const ffo_net* net, double t0, double t1, double rtol, double atol,
                     const double* z, double* x_out, int* nfev) {
  if (n * d > FFO_MAXM) return 1;
  batch_ctx c = {net, B, n, d, 0};
  size_t len = (size_t)B * n * d;
  memcpy(x_out, z, sizeof(double) * len);
  return ffo_rk45(rhs_batch, &c, t0, t1, x_out, (int)len, rtol, atol, nfev, NULL);
}

/* integrates (x, 0) from t1 to t0 (t_span_reverse, src/flow.py:40,53) */
int ffo_cnf_delta_logp(int64_t B, int n, int d, const ffo_net* net, double t0, double t1, double rtol, double atol,
                       const double* x, double* z_out, double* dlogp_out, int* nfev) {
  if (n * d > FFO_MAXM) return 1;
  batch_ctx c = {net, B, n, d, 1};
  size_t M = (size_t)n * d, len = (size_t)B * M + B;
  double* y = (double*)malloc(sizeof(double) * len);
  memcpy(y, x, sizeof(double) * B * M);
  memset(y + B * M, 0, sizeof(double) * B);
  int st = ffo_rk45(rhs_batch, &c, t1, t0, y, (int)len, rtol, atol, nfev, NULL);
  memcpy(z_out, y, sizeof(double) * B * M);
  memcpy(dlogp_out, y + B * M, sizeof(double) * B);
  free(y);
  return st;
}

/* ---------------------------------------------------------------------------------------------
 * Adjoint of delta_logp: SolveIVP.backward with F_augFull (src/NeuralODE/nnModule.py:76-133).
 * Augmented state (xs, adjoint_xs, adjoint_params) = (z, D, a_z, a_D, a_theta), integrated over the
 * reversed span (t0 -> t1) starting from the forward result.  RHS = (f, vjp_xs, vjp_params) with
 * forward_value = -(a_z . v(z) - a_D * div v(z)).
 * d/dz of forward_value is formed from the dense Jacobian of (v, div) obtained with n*d first-order
 * jets (no symmetry assumed); d/dtheta is written out from src/MLP.py:30-45.
 * Parameter order = Backflow.parameters(): eta.fc1.weight, eta.fc1.bias, eta.fc2.weight, mu.* .   */
typedef struct { const ffo_net* net; int64_t B; int n, d; int P; } adj_ctx;

static void mlp_param_terms(int H, const double* w1, const double* b1, const double* w2, double r,
                            double ca /* coeff on f(r) */, double cb /* coeff on f'(r) */, double* g /* [3H]: w1,b1,w2 */) {
  for (int h = 0; h < H; h++) {
    double s = 1.0 / (1.0 + exp(-(w1[h] * r + b1[h])));
    double s1 = s * (1.0 - s), s2 = s1 * (1.0 - 2.0 * s);
    g[h] += ca * w2[h] * s1 * r + cb * w2[h] * (s1 + w1[h] * r * s2);
    g[H + h] += ca * w2[h] * s1 + cb * w2[h] * w1[h] * s2;
    g[2 * H + h] += ca * s + cb * w1[h] * s1;
  }
}

static void rhs_adjoint(void* vctx, double t, const double* y, double* dy) {
  (void)t;
  adj_ctx* c = (adj_ctx*)vctx;
  const ffo_net* net = c->net;
  int n = c->n, d = c->d, M = n * d, P = c->P;
  int64_t B = c->B;
  const double *Z = y, *AZ = y + B * M + B, *AD = y + 2 * B * M + B;
  double *dZ = dy, *dD = dy + B * M, *dAZ = dy + B * M + B, *dAD = dy + 2 * B * M + B, *dTH = dy + 2 * B * M + 2 * B;
  int nth = ffo_num_threads();
  double* gpart = (double*)calloc((size_t)nth * P, sizeof(double));
#pragma omp parallel
  {
#ifdef _OPENMP
    double* g = gpart + (size_t)omp_get_thread_num() * P;
#else
    double* g = gpart;
#endif
#pragma omp for schedule(static)
    for (int64_t b = 0; b < B; b++) {
      const double *z = Z + b * M, *az = AZ + b * M; double ad = AD[b];
      /* f values */
      jet_o0 z0[FFO_MAXM], f0[FFO_MAXM], dv0;
      for (int i = 0; i < M; i++) z0[i] = jc_o0(z[i]);
      backflow_o0(net, n, d, z0, f0, &dv0);
      for (int i = 0; i < M; i++) dZ[b * M + i] = f0[i].v;
      dD[b] = -dv0.v;
      /* vjp_x[i] = d/dz_i [-(a_z.v - a_D div)] */
      for (int i = 0; i < M; i++) {
        jet_o1 z1[FFO_MAXM], f1[FFO_MAXM], dv1;
        for (int k = 0; k < M; k++) z1[k] = jmk_o1(z[k], k == i ? 1.0 : 0.0, 0.0);
        backflow_o1(net, n, d, z1, f1, &dv1);
        double s = 0.0;
        for (int k = 0; k < M; k++) s += az[k] * f1[k].d1;
        dAZ[b * M + i] = -(s - ad * dv1.d1);
      }
      dAD[b] = 0.0;
      /* vjp_params */
      for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++) {
          double r2 = 0.0, al = 0.0;
          for (int k = 0; k < d; k++) {
            double rho = z[i * d + k] - z[j * d + k];
            r2 += rho * rho; al += (az[i * d + k] - az[j * d + k]) * rho;
          }
          double r = sqrt(r2);
          mlp_param_terms(net->He, net->ew1, net->eb1, net->ew2, r, -(al - 2.0 * d * ad), 2.0 * ad * r, g);
        }
      if (net->Hm > 0)
        for (int i = 0; i < n; i++) {
          double r2 = 0.0, al = 0.0;
          for (int k = 0; k < d; k++) { r2 += z[i * d + k] * z[i * d + k]; al += az[i * d + k] * z[i * d + k]; }
          double r = sqrt(r2);
          mlp_param_terms(net->Hm, net->mw1, net->mb1, net->mw2, r, -(al - d * ad), ad * r, g + 3 * net->He);
        }
    }
  }
  for (int k = 0; k < P; k++) { double s = 0.0; for (int tt = 0; tt < nth; tt++) s += gpart[(size_t)tt * P + k]; dTH[k] = s; }
  free(gpart);
}

int ffo_cnf_adjoint(int64_t B, int n, int d, const ffo_net* net, double t0, double t1, double rtol, double atol,
                    const double* z_t0, const double* dlogp_t0, const double* a_z, const double* a_d,
                    double* grad_x, double* grad_params, int* nfev) {
  if (n * d > FFO_MAXM) return 1;
  int M = n * d, P = 3 * net->He + 3 * (net->Hm > 0 ? net->Hm : 0);
  adj_ctx c = {net, B, n, d, P};
  size_t len = (size_t)2 * B * M + 2 * B + P;
  double* y = (double*)calloc(len, sizeof(double));
  memcpy(y, z_t0, sizeof(double) * B * M);
  memcpy(y + B * M, dlogp_t0, sizeof(double) * B);
  memcpy(y + B * M + B, a_z, sizeof(double) * B * M);
  memcpy(y + 2 * B * M + B, a_d, sizeof(double) * B);
  int st = ffo_rk45(rhs_adjoint, &c, t0, t1, y, (int)len, rtol, atol, nfev, NULL);
  if (grad_x) memcpy(grad_x, y + B * M + B, sizeof(double) * B * M);
  memcpy(grad_params, y + 2 * B * M + 2 * B, sizeof(double) * P);
  free(y);
  return st;
}

/* =============================================================================================
 * Local energy (src/VMC.py:46-55).  logp(x) = logp0(z(x)) - Delta(x);  for every coordinate
 * direction i a 2nd-order jet of the whole map x -> logp is pushed through the ODE solve.
 * ============================================================================================= */
typedef struct { const ffo_net* net; int n, d; } walker_ctx;
static void rhs_walker_o2(void* vctx, double t, const double* y, double* dy) {
  (void)t; walker_ctx* c = (walker_ctx*)vctx; rhs_xlogp_o2(c->net, c->n, c->d, y, dy);
}

static int eloc_any(int64_t B, int nup, int ndn, int d, const int* tab_up, const int* tab_dn, const int* wstate,
                    const ffo_net* net, double t0, double t1, double rtol, double atol, double Zc, int use_ho,
                    const double* x, double* logp, double* grad, double* lap, double* V, double* eloc);
int ffo_eloc(int64_t B, int nup, int ndn, const int* tab_up, const int* tab_dn, const int* wstate,
             const ffo_net* net, double t0, double t1, double rtol, double atol, double Zc, int use_ho,
             const double* x, double* logp, double* grad, double* lap, double* V, double* eloc) {
  return eloc_any(B, nup, ndn, 2, tab_up, tab_dn, wstate, net, t0, t1, rtol, atol, Zc, use_ho, x, logp, grad, lap, V, eloc);
}
/* the same in three dimensions (HO3D orbitals: no upstream orbital list; pinned to the reference's dimension-generic code by
 * tests/golden/g7_3d.npz) */
int ffo_eloc3d(int64_t B, int nup, int ndn, const int* tab_up, const int* tab_dn, const int* wstate,
               const ffo_net* net, double t0, double t1, double rtol, double atol, double Zc, int use_ho,
               const double* x, double* logp, double* grad, double* lap, double* V, double* eloc) {
  return eloc_any(B, nup, ndn, 3, tab_up, tab_dn, wstate, net, t0, t1, rtol, atol, Zc, use_ho, x, logp, grad, lap, V, eloc);
}
static int eloc_any(int64_t B, int nup, int ndn, int d, const int* tab_up, const int* tab_dn, const int* wstate,
                    const ffo_net* net, double t0, double t1, double rtol, double atol, double Zc, int use_ho,
                    const double* x, double* logp, double* grad, double* lap, double* V, double* eloc) {
  int n = nup + ndn, M = n * d, len = M + 1;
  if (M > FFO_MAXM) return 1;
  int fail = 0;
  double* Vloc = V ? V : (double*)malloc(sizeof(double) * B);
  ffo_potential(B, n, d, Zc, use_ho, x, Vloc);
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t b = 0; b < B; b++) {
    const int* ou = nup ? row(tab_up, wstate, b, nup) : NULL;
    const int* od = ndn ? row(tab_dn, wstate, b, ndn) : NULL;
    walker_ctx c = {net, n, d};
    double y[3 * (FFO_MAXM + 1)], l = 0.0, g2 = 0.0, lp = 0.0;
    for (int i = 0; i < M; i++) {
      memset(y, 0, sizeof(double) * 3 * len);
      for (int k = 0; k < M; k++) y[k] = x[b * M + k];
      y[len + i] = 1.0;
      int st = ffo_rk45(rhs_walker_o2, &c, t1, t0, y, 3 * len, rtol, atol, NULL, NULL);
      if (st) fail = 1;
      jet_o2 zj[FFO_MAXM + 1];
      unpack_o2(y, len, zj);
      jet_o2 r = jsub_o2(logprob_d_o2(nup, ndn, d, ou, od, zj), zj[M]);
      lp = r.v; l += r.d2; g2 += r.d1 * r.d1;
      if (grad) grad[b * M + i] = r.d1;
    }
    if (logp) logp[b] = lp;
    if (lap) lap[b] = l;
    if (eloc) eloc[b] = -0.25 * l - 0.125 * g2 + Vloc[b];
  }
  if (!V) free(Vloc);
  return fail ? -2 : 0;
}

/* =============================================================================================
 * One complete GSVMC iteration (src/VMC.py:40-59 + src/FermionHO2D.py:69-71), used as the CPU
 * baseline ("port") by bench.py.  MCMC noise comes from an internal xoshiro256** + Box-Muller
 * stream (timing only; parity of the MCMC step is checked through ffo_mcmc_noise).
 * ============================================================================================= */
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
typedef struct { uint64_t s[4]; } xo_t;
static uint64_t xo_next(xo_t* g) {
  uint64_t* s = g->s; uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45); return r;
}
static void xo_seed(xo_t* g, uint64_t seed) {
  for (int i = 0; i < 4; i++) { seed += 0x9E3779B97F4A7C15ull; uint64_t z = seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; g->s[i] = z ^ (z >> 31); }
}
static double xo_unif(xo_t* g) { return (double)(xo_next(g) >> 11) * (1.0 / 9007199254740992.0); }
static double xo_normal(xo_t* g) {
  double u1 = 1.0 - xo_unif(g), u2 = xo_unif(g);
  return sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
}

int ffo_gsvmc_sweep(int64_t B, int nup, int ndn, const ffo_net* net, double t0, double t1, double rtol, double atol,
                    double Zc, int use_ho, int mcmc_steps, double tau, uint64_t seed,
                    double* E_out, double* Estd_out, double* gradE_out, double* grad_params, double* stage_seconds /*[5]*/) {
  int n = nup + ndn, d = 2, M = n * d;
  int orb[FFO_MAXN]; for (int i = 0; i < FFO_MAXN; i++) orb[i] = i;   /* orbitals[:nup], orbitals[:ndown], src/VMC.py:22-23 */
  double *z = (double*)malloc(sizeof(double) * B * M), *x = (double*)malloc(sizeof(double) * B * M);
  double *zb = (double*)malloc(sizeof(double) * B * M), *dl = (double*)malloc(sizeof(double) * B);
  double *el = (double*)malloc(sizeof(double) * B), *lp0 = (double*)malloc(sizeof(double) * B);
  double *az = (double*)malloc(sizeof(double) * B * M), *ad = (double*)malloc(sizeof(double) * B);
  double tm[6];
#ifdef _OPENMP
#define NOW() omp_get_wtime()
#else
#define NOW() 0.0
#endif
  tm[0] = NOW();
  /* FreeFermion.sample */
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; b++) {
    xo_t g; xo_seed(&g, seed * 0x100000001B3ull + (uint64_t)b);
    double xs[FFO_MAXM], nx[FFO_MAXM], logp, nl;
    for (int i = 0; i < M; i++) xs[i] = xo_normal(&g);
    logprob_glap(nup, ndn, orb, orb, xs, &logp, NULL, NULL);
    for (int s = 0; s < mcmc_steps; s++) {
      for (int i = 0; i < M; i++) nx[i] = xs[i] + tau * xo_normal(&g);
      logprob_glap(nup, ndn, orb, orb, nx, &nl, NULL, NULL);
      if (xo_unif(&g) < exp(nl - logp)) { memcpy(xs, nx, sizeof(double) * M); logp = nl; }
    }
    memcpy(z + b * M, xs, sizeof(double) * M);
  }
  tm[1] = NOW();
  int st = ffo_cnf_generate(B, n, d, net, t0, t1, rtol, atol, z, x, NULL);
  tm[2] = NOW();
  /* logp_full = self.logp(x, params_require_grad=True) */
  st |= ffo_cnf_delta_logp(B, n, d, net, t0, t1, rtol, atol, x, zb, dl, NULL);
  tm[3] = NOW();
  st |= ffo_eloc(B, nup, ndn, orb, orb, NULL, net, t0, t1, rtol, atol, Zc, use_ho, x, NULL, NULL, NULL, NULL, el);
  tm[4] = NOW();
  double E = 0.0; for (int64_t b = 0; b < B; b++) E += el[b]; E /= (double)B;
  double var = 0.0; for (int64_t b = 0; b < B; b++) var += (el[b] - E) * (el[b] - E);
  double gradE = 0.0;
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; b++) {
    double w = (el[b] - E) / (double)B;
    logprob_glap(nup, ndn, orb, orb, zb + b * M, lp0 + b, az + b * M, NULL);
    for (int i = 0; i < M; i++) az[b * M + i] *= w;
    ad[b] = -w;
  }
  for (int64_t b = 0; b < B; b++) gradE += (lp0[b] - dl[b]) * (el[b] - E) / (double)B;
  st |= ffo_cnf_adjoint(B, n, d, net, t0, t1, rtol, atol, zb, dl, az, ad, NULL, grad_params, NULL);
  tm[5] = NOW();
  *E_out = E; *Estd_out = B > 1 ? sqrt(var / (double)(B - 1)) : 0.0; *gradE_out = gradE;
  if (stage_seconds) for (int i = 0; i < 5; i++) stage_seconds[i] = tm[i + 1] - tm[i];
  free(z); free(x); free(zb); free(dl); free(el); free(lp0); free(az); free(ad);
  return st;
}
