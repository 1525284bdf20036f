// ff_fwd_args.h -- launch arguments shared by every forward CNF kernel (ff_cnf_fwd.hip: one wave per walker group;
// ff_wide.hip: one walker per workgroup) and the hand-over between the two translation units.
#pragma once
#include "ff_common.h"

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
  int h_equal;             // flow kernels: the opening step rounded down to equal steps of the interval (ff_ode.walker_h_equal)
  double* h_out;           // optional (B): largest step size accepted for every walker in this call (ff_ode.walker_h_out)
  int32_t* wcost;         // optional (B): attempted steps of every walker (ff_ode.walker_cost)
  const int32_t* order;    // optional (B): workgroups take walkers in this order (ff_ode.walker_order); results stay in place
  // Off-table protocol (TAB kernels): a kernel that meets a radius beyond the table, or an unusable table, stores
  // evt_id into *evt (a slot of the table header); the direct-evaluation kernel launched right behind it with the
  // same id returns at once unless it finds its id there.  ids are unique per process, so slots need no reset.
  double* evt;
  double evt_id;
  // local-energy pass, optional (B): a cost class per walker (ff_ode.walker_class).  Walkers with class <= sens_class weigh
  // the sensitivity components (J, kbar, the Delta derivatives) with sens_w = 1 / ff_ode.sens_tol in the error norm and
  // open with h_init * h_scale_loose
  const int32_t* wclass;
  int sens_class;
  double sens_w, h_scale_loose;
  // Work queue (optional): with `queue` set the launch is a persistent grid and every workgroup takes its next walker
  // group from this counter (slot 0: table kernel, slot 1: direct kernel), zeroed by the host before the launch.
  unsigned long long* queue;
  // Routing of the local-energy pass by cost class (needs wclass): heavy_mode 1 = this launch integrates ONLY the walkers with
  // class >= heavy_class, 2 = only the others, 0 = every walker (ff_cnf_fwd.hip, launch_mfma)
  int heavy_mode, heavy_class;
  double heavy_tol;        // tolerances of the heavy launch: heavy_tol x (rtol, atol)  (ff_ode.heavy_tol)
  double sum_w;            // ff_ode.sum_weight: error-norm weight of Delta and lap Delta in the matrix-core kernel
  // Fused finish (ff_eloc; kernels that implement it: ff_eloc_mfma_kernel for nup = ndn): with fin.on the sensitivity kernel
  // contracts J, kbar, grad Delta and lap Delta with the Slater gradient / Hessian at z(t0) in its epilogue and writes logp, grad,
  // lap, V, E_loc (src/VMC.py:46-55) and grad_z logp0 itself -- J^T never goes to HBM (Jt, kbar, dD, Lpart are not written;
  // y_out = z(t0) and dl_out = Delta still are: the adjoint reads them).  Any output pointer may be NULL.
  struct ff_fin_args {
    int on;      // bit 0: the matrix-core kernel may fuse (nup = ndown <= 3), bit 1: the one-walker-per-workgroup kernels may (any shape)
    int nup, ndn, use_ho;
    const int32_t *tab_up, *tab_dn, *wstate;
    double Z;
    double *logp, *grad, *lap, *V, *eloc, *glogp0;
    void* workspace;      // host side only: the caller's ff_eloc workspace (finish of the walkers of the heavy route)
  } fin;
};

// Kernels for walkers that do not fit one wave's column / row layouts (n > 12 in d = 2, n > 4 in d = 3): ff_wide.hip.
// mode: 0 CNF.generate, 1 CNF.delta_logp, 2 local-energy sensitivities.  Returns FF_OK / FF_EUNSUPPORTED / FF_ELAUNCH.
int ff_wide_dispatch_fwd(int mode, void* stream, int n, int d, const ff_fwd_args& a);
// nonzero if the wide family serves (n, d)
int ff_wide_supported(int n, int d);
// the local-energy kernel of the family for the walkers a launch with heavy_mode = 1 selects: at most `max_groups` single-walker workgroups
int ff_wide_eloc_heavy(void* stream, int n, int d, const ff_fwd_args& a, int64_t max_groups);
// FF_WIDE=1 in the environment routes EVERY particle number to the wide family (A/B and parity testing)
bool ff_wide_forced();
