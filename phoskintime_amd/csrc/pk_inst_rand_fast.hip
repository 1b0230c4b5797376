// Instantiations + launcher of the random-model throughput kernel (pk_rand_fast.hpp).
#include "pk_rand_fast.hpp"
#include "pk_rand_fast2.hpp"
#include <cstdlib>
#include "pk_launch.hpp"

namespace pk {

template <int NB>
static void launch_nb(const SolveArgs& a, int method, hipStream_t st) {
  constexpr int G = (1 << NB) < 4 ? 4 : (1 << NB);
  const long long rpb = 256 / G;
  const long long nblk = (a.B + rpb - 1) / rpb;
  if (method == PK_METHOD_LRP8) hipLaunchKernelGGL((rand_fast_kernel<NB, PK_METHOD_LRP8>), dim3((unsigned)nblk), dim3(256), 0, st, a);
  else                          hipLaunchKernelGGL((rand_fast_kernel<NB, PK_METHOD_RODAS4>), dim3((unsigned)nblk), dim3(256), 0, st, a);
}

void launch_rand_fast(const SolveArgs& a, int method, hipStream_t st) {
  switch (a.n_sites) {
    case 1: launch_nb<1>(a, method, st); break;
    case 2: launch_nb<2>(a, method, st); break;
    case 3: launch_nb<3>(a, method, st); break;
    case 4: launch_nb<4>(a, method, st); break;
    default: {
      // n = 5: two rows per lane in 16-lane groups (DPP broadcasts); PK_RAND5_G32=1 selects the one-row-per-lane 32-lane kernel (A/B)
      static const bool g32 = getenv("PK_RAND5_G32") && atoi(getenv("PK_RAND5_G32")) != 0;
      if (g32) { launch_nb<5>(a, method, st); break; }
      const long long nblk = (a.B + 15) / 16;
      if (method == PK_METHOD_LRP8) hipLaunchKernelGGL((rand_fast2_kernel<PK_METHOD_LRP8>), dim3((unsigned)nblk), dim3(256), 0, st, a);
      else                          hipLaunchKernelGGL((rand_fast2_kernel<PK_METHOD_RODAS4>), dim3((unsigned)nblk), dim3(256), 0, st, a);
    } break;
  }
}

}  // namespace pk
