// ff_eloc_ws.h -- layout of the ff_eloc workspace between its launches (shared by ff_cnf_fwd.hip and ff_ho3d.hip):
//   z(t0) (B,M) | Jt (B,M,M) | kbar (B,M) | dDelta (B,M) | lap parts (B,M) | Delta (B) | Slater table Q (B, nq) | 2 work counters
// (the positions of z(t0) and Delta are documented in include/fermiflow.h: callers may read them in place; ff_eloc_nd with more than
// 24 coordinates uses the compact layout below)
// nq = M + d(d+1)/2 n + d n^2 + 2: g0, the same-particle Hessian sums S, the gradient-times-inverse tables T of both spins
// (at most d n^2 entries), 2 log|det| per spin.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct ff_eloc_ws { double *z0, *Jt, *kbar, *dD, *Lp, *dl, *Q; unsigned long long* queue; };

static inline size_t ff_eloc_nq_max(size_t n, size_t d) { return n * d + (d * (d + 1) / 2) * n + d * n * n + 2; }
// COMPACT layout of the single-call ff_eloc_nd for M = n d > 24:  z(t0) (B,M) | Delta (B) | 2 work counters.  Beyond 24 coordinates
// only the one-walker-per-workgroup kernels serve a walker, and they finish it in their epilogue (ff_fwd_args::fin): no sensitivity
// ever goes to HBM, and the workspace no longer scales with M^2 (3.8 GB at configs[4]).
static inline bool ff_eloc_ws_compact(size_t n, size_t d) { return n * d > 24; }
static inline size_t ff_eloc_ws_doubles(int64_t B, size_t n, size_t d, bool compact = false) {
  const size_t M = n * d;
  if (compact) return (size_t)B * (M + 1) + 2;
  return (size_t)B * (M * M + 4 * M + 1 + ff_eloc_nq_max(n, d)) + 2;
}
static inline ff_eloc_ws ff_eloc_carve(void* workspace, int64_t B, size_t n, size_t d, bool compact = false) {
  const size_t M = n * d;
  double* w = (double*)workspace;
  ff_eloc_ws o;
  o.z0 = w;   w += (size_t)B * M;
  if (compact) {
    o.Jt = o.kbar = o.dD = o.Lp = o.Q = nullptr;
    o.dl = w;
  } else {
    o.Jt = w;   w += (size_t)B * M * M;
    o.kbar = w; w += (size_t)B * M;
    o.dD = w;   w += (size_t)B * M;
    o.Lp = w;   w += (size_t)B * M;
    o.dl = w;   w += (size_t)B;
    o.Q = w;
  }
  o.queue = (unsigned long long*)((double*)workspace + ff_eloc_ws_doubles(B, n, d, compact) - 2);
  return o;
}
