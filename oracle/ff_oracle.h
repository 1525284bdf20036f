/* ff_oracle.h -- ORACLE (test infrastructure): C API of the CPU restatement. See ff_oracle.c. */
#ifndef FF_ORACLE_H
#define FF_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* the two scalar MLPs of the backflow (src/MLP.py): He/Hm hidden sizes, Hm = 0 <=> mu=None */
typedef struct {
  int He; const double *ew1, *eb1, *ew2;
  int Hm; const double *mw1, *mb1, *mw2;
} ffo_net;

int ffo_num_threads(void);
int ffo_orbitals(const int* k, int nk, const double* pts, int npts, double* out);
int ffo_logprob(int64_t B, int nup, int ndn, const int* tab_up, const int* tab_dn, const int* wstate,
                const double* x, double* logp, double* grad, double* lap);
int ffo_mcmc_noise(int64_t B, int nup, int ndn, const int* tab_up, const int* tab_dn, const int* wstate,
                   int steps, double tau, const double* g0, const double* g, const double* u,
                   double* x_out, double* logp_out, uint8_t* accept_out);
/* HO3D counterparts (d = 3; no upstream orbital list: SURVEY 8(f).4 -- pinned by tests/golden/g7_3d.npz) */
int ffo_orbitals3d(const int* k, int nk, const double* pts, int npts, double* out);
int ffo_logprob3d(int64_t B, int nup, int ndn, const int* tab_up, const int* tab_dn, const int* wstate,
                  const double* x, double* logp, double* grad, double* lap);
int ffo_mcmc_noise3d(int64_t B, int nup, int ndn, const int* tab_up, const int* tab_dn, const int* wstate,
                     int steps, double tau, const double* g0, const double* g, const double* u,
                     double* x_out, double* logp_out, uint8_t* accept_out);
int ffo_eloc3d(int64_t B, int nup, int ndn, const int* tab_up, const int* tab_dn, const int* wstate,
               const ffo_net* net, double t0, double t1, double rtol, double atol, double Zc, int use_ho,
               const double* x, double* logp, double* grad, double* lap, double* V, double* eloc);
int ffo_backflow(int64_t B, int n, int d, const ffo_net* net, const double* x, double* v, double* div);
int ffo_potential(int64_t B, int n, int d, double Z, int use_ho, const double* x, double* V);
int ffo_cnf_generate(int64_t B, int n, int d, const ffo_net* net, double t0, double t1, double rtol, double atol,
                     const double* z, double* x_out, int* nfev);
int ffo_cnf_delta_logp(int64_t B, int n, int d, const ffo_net* net, double t0, double t1, double rtol, double atol,
                       const double* x, double* z_out, double* dlogp_out, int* nfev);
int ffo_cnf_adjoint(int64_t B, int n, int d, const ffo_net* net, double t0, double t1, double rtol, double atol,
                    const double* z_t0, const double* dlogp_t0, const double* a_z, const double* a_d,
                    double* grad_x, double* grad_params, int* nfev);
int ffo_eloc(int64_t B, int nup, int ndn, const int* tab_up, const int* tab_dn, const int* wstate,
             const ffo_net* net, double t0, double t1, double rtol, double atol, double Zc, int use_ho,
             const double* x, double* logp, double* grad, double* lap, double* V, double* eloc);
int ffo_gsvmc_sweep(int64_t B, int nup, int ndn, const ffo_net* net, double t0, double t1, double rtol, double atol,
                    double Zc, int use_ho, int mcmc_steps, double tau, uint64_t seed,
                    double* E_out, double* Estd_out, double* gradE_out, double* grad_params, double* stage_seconds);
#ifdef __cplusplus
}
#endif
#endif
