// ff_cnf_fwd.hip -- fused forward CNF integrations:
//   MODE 0  CNF.generate      (src/flow.py:42-44)     state z                      heads eta
//   MODE 1  CNF.delta_logp    (src/flow.py:51-55)     state (z, Delta)             heads eta, eta'
//   MODE 2  local-energy pass (src/VMC.py:46-49, replaces the 2+2*n*d nested adjoint solves of
//           src/utils.py:40-65)  state (z, J = dz/dx, kbar = lap_x z, Delta, grad_x Delta, lap_x Delta)
//                                                                                heads eta .. eta'''
// All of one walker's stages, error control and accept/reject happen on chip; HBM sees the walker's
// coordinates once on the way in and the results once on the way out.
//
// Per RHS evaluation (all lanes of the wave):
//   1. every lane publishes its stage value z_i (and kbar_i) to LDS;
//   2. "radius phase": the wave's G*R radii (pairs r_ab, one-body r_a) are dealt one per lane (the assignment is fixed for
//      the launch); a lane gets the derivative heads of eta (or mu) at its radius -- from the per-launch table (TAB
//      instantiations) or by evaluating the H sigmoids (direct instantiations) -- and leaves in LDS what does not depend
//      on the direction: MODE 0/1 the heads, MODE 2 a record per radius plus the radius' contributions to its particles'
//      own rows (v, Dv[kbar], grad div);
//   3. MODE 2 "jet sweep": lane (g,i) gathers its own row, then pushes its direction u_i = dz/dx_i through every radius
//      term as a 2nd-order Taylor jet (first-order part -> dJ/dt column, quadratic part -> source of kbar and lap Delta);
//      MODE 0/1 "component phase": lane (g,i) assembles v_i (and div v) from the heads of its particle's radii;
//   4. Dormand-Prince stage bookkeeping; per-walker error norm and step-size control (ff_ode.h).
// n >= 8 uses ff_eloc_split_kernel (two lanes per direction) for MODE 2.  The local-energy finish (Slater table +
// contraction with the sensitivities) is at the end of the file.
#include <atomic>
#include "ff_common.h"
#include "ff_ode.h"
#include "ff_slater.h"
#define FF_RADIAL_BUILD_KERNELS
#include "ff_radial.h"

// FF_STAMPS: diagnostic build only (tools/kbench.py --stamps): per-phase s_memtime shares of the RHS loop,
// added into stats[8..] as 64-bit counters.  Never defined in the product build.
#ifdef FF_STAMPS
#define FF_STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += t_ - stamp_prev; stamp_prev = t_; } while (0)
#else
#define FF_STAMP(i) do { } while (0)
#endif

// which kernels use the LDS-table exp (ff_exp_tab): tuning knob, see tools/kbench.py A/B runs
#ifndef FF_TAB_MODE
#define FF_TAB_MODE(MODE) true
#endif

struct ff_fwd_args {
  int64_t B;
  ff_net net;
  double ta, tb, rtol, atol;
  int max_steps;
  const double* y_in;   // (B, M)
  double* y_out;        // (B, M)   z(tb)
  double* dl_out;       // (B)      Delta(tb)                  MODE >= 1
  double* Jt;           // (B, M, M) Jt[b][i][k] = dz_k/dx_i   MODE 2
  double* kbar;         // (B, M)
  double* dD;           // (B, M)   d Delta / d x_i
  double* Lpart;        // (B, M)   per-direction parts of lap_x Delta
  int32_t* stats;
  const double* h_init;    // optional (B): first step size to try for every walker (ff_ode.walker_h_init), times h_scale
  double h_scale;          // negative: h_init holds ONE entry used by every walker (ff_ode.walker_h_uniform), scale = -h_scale
  double* h_out;           // optional (B): largest step size accepted for every walker in this call (ff_ode.walker_h_out)
  int32_t* wcost;         // optional (B): attempted steps of every walker (ff_ode.walker_cost)
  const int32_t* order;    // optional (B): workgroups take walkers in this order (ff_ode.walker_order); results stay in place
  // Off-table protocol (TAB kernels): a kernel that meets a radius beyond the table, or an unusable table, stores
  // evt_id into *evt (a slot of the table header); the direct-evaluation kernel launched right behind it with the
  // same id returns at once unless it finds its id there.  ids are unique per process, so slots need no reset.
  double* evt;
  double evt_id;
  // Work queue (optional): with `queue` set the launch is a persistent grid and every workgroup takes its next walker
  // group from this counter (slot 0: table kernel, slot 1: direct kernel), zeroed by the host before the launch.
  unsigned long long* queue;
};

#ifndef FF_FORM_EARLY
#define FF_FORM_EARLY 1   // MODE 2: form the stage input between the two halves of the radius phase
#endif
#ifndef FF_SWEEP_CH
#define FF_SWEEP_CH 2   // records per look-ahead chunk of the jet sweep (measured at n = 6: 2 -> 1.49 ms, 3 -> 1.52, 4 spills)
#endif
#ifndef FF_FWD_WAVES_PER_SIMD
#define FF_FWD_WAVES_PER_SIMD 1
#endif
#include "ff_eloc_mfma.h"
#ifndef FF_TW
#define FF_TW 1
#endif
template __global__ void ff_eloc_mfma_kernel<FF_TN, 2, true, FF_TW>(ff_fwd_args);
