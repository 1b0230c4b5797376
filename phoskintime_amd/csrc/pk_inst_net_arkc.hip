// Instantiations of the dense two-lanes-per-protein integrator for the sequential topology (model 1: tridiagonal block, twisted
// factorisation across the lane pair; pk_network_solve_arkp.hpp), site classes 4 / 6 / 8.
#include <type_traits>
#include "pk_network_solve_arkp.hpp"

namespace pk {

hipError_t launch_net_arkp_chain(const NetDev& n, const NetSolveArgs& a, int threads, size_t lds, bool park, int nrl, long long B, hipStream_t st) {
#define PK_ARKC(K, R)                                                                                                                 \
  do {                                                                                                                                \
    if (lds > 64 * 1024) {                                                                                                            \
      hipError_t e_ = hipFuncSetAttribute((const void*)K<1, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);              \
      if (e_ != hipSuccess) return e_;                                                                                                \
    }                                                                                                                                 \
    hipLaunchKernelGGL((K<1, R>), dim3((unsigned)B), dim3(threads), lds, st, n, a);                                                   \
  } while (0)
  if (park) { if (nrl == 3) PK_ARKC(net_solve_arkp3_kernel, 3); else if (nrl == 4) PK_ARKC(net_solve_arkp3_kernel, 4); else PK_ARKC(net_solve_arkp3_kernel, 5); }
  else      { if (nrl == 3) PK_ARKC(net_solve_arkp_kernel, 3);  else if (nrl == 4) PK_ARKC(net_solve_arkp_kernel, 4);  else PK_ARKC(net_solve_arkp_kernel, 5); }
#undef PK_ARKC
  return hipSuccess;
}

}  // namespace pk
