// Instantiations + launcher of the order-4 network integrator in the dense two-lanes-per-protein layout (pk_network_solve_arkp.hpp):
// arrow topologies 0 / 4, site classes 4 / 6 / 8 (3 / 4 / 5 rows per lane).
#include <type_traits>
#include "pk_network_solve_arkp.hpp"
#include <cstdlib>

namespace pk {

bool net_arkp_enabled() {
  static const bool on = [] { const char* v = getenv("PK_ARK_PAIR"); return !(v && v[0] == '0'); }();
  return on;
}

hipError_t launch_net_arkp(const NetDev& n, const NetSolveArgs& a, int nnzT, int max_sites, long long B, hipStream_t st) {
  const int threads = ((n.n_lanes + 63) / 64) * 64;
  const size_t lds = net_solve_arkp_lds_bytes(n, nnzT);
  if (threads > 512 || lds > 64 * 1024) return hipErrorInvalidValue;
  const int nrl = arkp_rows_per_lane(max_sites <= 4 ? 4 : max_sites <= 6 ? 6 : 8);
#define PK_ARKP(M, R) hipLaunchKernelGGL((net_solve_arkp_kernel<M, R>), dim3((unsigned)B), dim3(threads), lds, st, n, a)
  if (n.model == 0) { if (nrl == 3) PK_ARKP(0, 3); else if (nrl == 4) PK_ARKP(0, 4); else PK_ARKP(0, 5); }
  else              { if (nrl == 3) PK_ARKP(4, 3); else if (nrl == 4) PK_ARKP(4, 4); else PK_ARKP(4, 5); }
#undef PK_ARKP
  return hipSuccess;
}

}  // namespace pk
