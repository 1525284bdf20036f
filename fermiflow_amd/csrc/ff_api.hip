// ff_api.hip -- library-wide pieces of the C ABI (version, error string) and the host-side set-up code that the
// reference keeps in Python: enumeration of the low-lying many-body states (src/orbitals.py:14-54).
#include "ff_common.h"
#include <string.h>
#include <algorithm>
#include <vector>

static thread_local char g_ff_err[256] = "";

void ff_set_error(const char* msg) {
  strncpy(g_ff_err, msg ? msg : "", sizeof(g_ff_err) - 1);
  g_ff_err[sizeof(g_ff_err) - 1] = 0;
}

namespace {

struct Subset { std::vector<int32_t> idx; double price; };

// All index subsets of length k of 0..n-1 with total price <= pmax, in lexicographic order (the generation order of
// Orbitals.subsets, src/orbitals.py:14-31).  Prices are non-decreasing in the index (orbitals are listed shell by
// shell), so the cheapest completion of a prefix is the next `need` consecutive orbitals: the same pruning rule.
void grow(const double* price, int n, int k, double pmax, std::vector<int32_t>& prefix, double total, int start,
          std::vector<Subset>& out) {
  const int need = k - (int)prefix.size();
  if (need == 0) { out.push_back({prefix, total}); return; }
  for (int idx = start; idx + need - 1 < n; idx++) {
    double cheapest = 0.0;
    for (int q = 0; q < need; q++) cheapest += price[idx + q];
    if (cheapest <= pmax - total) {
      prefix.push_back(idx);
      grow(price, n, k, pmax, prefix, total + price[idx], idx + 1, out);
      prefix.pop_back();
    }
  }
}

std::vector<Subset> subsets(const double* price, int n, int k, double pmax) {
  std::vector<Subset> out;
  std::vector<int32_t> prefix;
  if (k == 0) { out.push_back({{}, 0.0}); return out; }
  grow(price, n, k, pmax, prefix, 0.0, 0, out);
  return out;
}

}  // namespace

extern "C" {
int ff_version(void) { return 109; }   // 109: ff_walker_schedule takes shrink_at (no struct change); 108: ff_ode gained walker_h_equal; 107: ff_scale_counts takes the interval and fills 128 counts, ff_walker_schedule reads as many (no struct change); 106: ff_ode gained after_main_event; 105: ff_walker_schedule, ff_rng_fill3d (no struct change); 104: ff_ode gained compact_finish; 103: ff_ode gained heavy_class / heavy_tol / sum_weight (102: walker_class / sens_tol / walker_h_scale_loose / sens_tol_class)
const char* ff_last_error(void) { return g_ff_err; }

// Orbitals.fermion_states (src/orbitals.py:33-54), host code, no GPU involved.  See include/fermiflow.h.
int64_t ff_fermion_states(int n_orb, const double* orb_E, int nup, int ndn, double deltaE, int64_t capacity,
                          int32_t* states_up, int32_t* states_dn, double* states_E) {
  if (n_orb <= 0 || !orb_E || nup < 0 || ndn < 0 || nup + ndn == 0 || nup > n_orb || ndn > n_orb || !(deltaE >= 0.0)) {
    ff_set_error("ff_fermion_states: bad argument");
    return -1;
  }
  for (int i = 1; i < n_orb; i++)
    if (orb_E[i] < orb_E[i - 1]) { ff_set_error("ff_fermion_states: orbital energies must be non-decreasing"); return -1; }
  double e0u = 0.0, e0d = 0.0;
  for (int i = 0; i < nup; i++) e0u += orb_E[i];
  for (int i = 0; i < ndn; i++) e0d += orb_E[i];
  // each spin species alone can be excited by at most deltaE; the pair by deltaE in total
  const std::vector<Subset> U = subsets(orb_E, n_orb, nup, e0u + deltaE), Dn = subsets(orb_E, n_orb, ndn, e0d + deltaE);
  struct St { int32_t u, d; double E; };
  std::vector<St> all;
  for (size_t u = 0; u < U.size(); u++)
    for (size_t d = 0; d < Dn.size(); d++)
      if (U[u].price + Dn[d].price <= e0u + e0d + deltaE) all.push_back({(int32_t)u, (int32_t)d, U[u].price + Dn[d].price});
  // by total energy; ties keep the (up, down)-lexicographic generation order (stable), as `sorted(..., key=price)` does
  std::stable_sort(all.begin(), all.end(), [](const St& a, const St& b) { return a.E < b.E; });
  const int64_t ns = (int64_t)all.size();
  if (ns <= capacity) {
    for (int64_t s = 0; s < ns; s++) {
      if (states_up) for (int k = 0; k < nup; k++) states_up[s * nup + k] = U[all[s].u].idx[k];
      if (states_dn) for (int k = 0; k < ndn; k++) states_dn[s * ndn + k] = Dn[all[s].d].idx[k];
      if (states_E) states_E[s] = all[s].E;
    }
  }
  return ns;
}
}
