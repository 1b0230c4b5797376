// Instantiations + launcher of the order-4 network integrator in the dense two-lanes-per-protein layout (pk_network_solve_arkp.hpp):
// arrow topologies 0 / 4 (here) and the sequential chain 1 (pk_inst_net_arkc.hip), site classes 4 / 6 / 8 (3 / 4 / 5 rows per lane).
#include <type_traits>
#include "pk_network_solve_arkp.hpp"
#include <cstdlib>

namespace pk {

bool net_arkp_enabled() {
  static const bool on = [] { const char* v = getenv("PK_ARK_PAIR"); return !(v && v[0] == '0'); }();
  return on;
}

// the sequential chain's instantiations live in their own translation unit (pk_inst_net_arkc.hip: parallel hipcc)
hipError_t launch_net_arkp_chain(const NetDev& n, const NetSolveArgs& a, int threads, size_t lds, bool park, int nrl, long long B, hipStream_t st);

// PK_ARK_PAIR: 0 = round 2's kernel (one thread per protein), 1 = the pair layout out of registers (256 VGPRs, 2 waves / SIMD),
// 3 = the pair layout on the register diet (168 VGPRs, 3 waves / SIMD)
static int arkp_mode() {
  static const int m = [] { const char* v = getenv("PK_ARK_PAIR"); return v ? atoi(v) : 3; }();
  return m;
}

bool net_arkp_fuses_loss() { return arkp_mode() == 3; }

// does the dense lane layout take this network?  (lane table present, <= 512 lanes, LDS of the chosen variant within 160 KB)
bool net_arkp_fits(const NetDev& n, int max_sites) {
  if (!n.lane_unit || n.n_lanes < 1 || n.n_lanes > 512 || !net_arkp_enabled()) return false;
  const int threads = ((n.n_lanes + 63) / 64) * 64;
  const int nrl = arkp_rows_per_lane(max_sites <= 4 ? 4 : max_sites <= 6 ? 6 : 8);
  return net_solve_arkp_lds_bytes(n, nrl, threads, arkp_mode() == 3) <= 160 * 1024;
}

hipError_t launch_net_arkp(const NetDev& n, const NetSolveArgs& a, int nnzT, int max_sites, long long B, hipStream_t st) {
  const int threads = ((n.n_lanes + 63) / 64) * 64;
  const int nrl = arkp_rows_per_lane(max_sites <= 4 ? 4 : max_sites <= 6 ? 6 : 8);
  const bool park = arkp_mode() == 3;
  const size_t lds = net_solve_arkp_lds_bytes(n, nrl, threads, park);
  if (threads > 512 || lds > 160 * 1024) return hipErrorInvalidValue;
  if (n.model == 1) return launch_net_arkp_chain(n, a, threads, lds, park, nrl, B, st);
#define PK_ARKP(K, M, R)                                                                                                              \
  do {                                                                                                                                \
    if (lds > 64 * 1024) {                                  /* beyond the default dynamic-LDS limit: raise it (idempotent, cheap) */  \
      hipError_t e_ = hipFuncSetAttribute((const void*)K<M, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);              \
      if (e_ != hipSuccess) return e_;                                                                                                \
    }                                                                                                                                 \
    hipLaunchKernelGGL((K<M, R>), dim3((unsigned)B), dim3(threads), lds, st, n, a);                                                   \
  } while (0)
#define PK_ARKP_ALL(K)                                                                                          \
  do {                                                                                                          \
    if (n.model == 0) { if (nrl == 3) PK_ARKP(K, 0, 3); else if (nrl == 4) PK_ARKP(K, 0, 4); else PK_ARKP(K, 0, 5); } \
    else              { if (nrl == 3) PK_ARKP(K, 4, 3); else if (nrl == 4) PK_ARKP(K, 4, 4); else PK_ARKP(K, 4, 5); } \
  } while (0)
  if (park) PK_ARKP_ALL(net_solve_arkp3_kernel); else PK_ARKP_ALL(net_solve_arkp_kernel);
#undef PK_ARKP_ALL
#undef PK_ARKP
  return hipSuccess;
}

}  // namespace pk
