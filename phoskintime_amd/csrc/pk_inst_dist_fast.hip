// Instantiations + launcher of the distributive-model throughput kernel (pk_dist_fast.hpp).
#include "pk_dist_fast.hpp"
#include "pk_launch.hpp"

namespace pk {

template <int G, int RPL>
static void launch_one(const SolveArgs& a, int method, hipStream_t st) {
  const long long rpb = 256 / G;
  const long long nblk = (a.B + rpb - 1) / rpb;
  if (method == PK_METHOD_LRP12)     hipLaunchKernelGGL((dist_fast_kernel<G, RPL, PK_METHOD_LRP12>), dim3((unsigned)nblk), dim3(256), 0, st, a);
  else if (method == PK_METHOD_LRP8) hipLaunchKernelGGL((dist_fast_kernel<G, RPL, PK_METHOD_LRP8>), dim3((unsigned)nblk), dim3(256), 0, st, a);
  else                          hipLaunchKernelGGL((dist_fast_kernel<G, RPL, PK_METHOD_RODAS4>), dim3((unsigned)nblk), dim3(256), 0, st, a);
}

// lanes per replica x site rows per lane, chosen so that G * RPL >= n_sites with the fewest idle slots
void launch_dist_fast(const SolveArgs& a, int method, hipStream_t st) {
  const int n = a.n_sites;
  if (n <= 4) launch_one<4, 1>(a, method, st);
  else if (n <= 8) launch_one<4, 2>(a, method, st);
  else if (n <= 16) launch_one<8, 2>(a, method, st);
  else if (n <= 32) launch_one<8, 4>(a, method, st);        // 4 lanes x 8 rows measured equal (65.9 vs 65.2 M/s) at half the occupancy
  else launch_one<16, 4>(a, method, st);
}

}  // namespace pk
