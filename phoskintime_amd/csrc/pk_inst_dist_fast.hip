// Instantiations + launcher of the distributive-model throughput kernel (pk_dist_fast.hpp): RODAS4 and LRP8.
// LRP12 (the default method) has its own, finer layout table in pk_inst_dist_fast12.hip.
#include "pk_dist_fast.hpp"
#include "pk_launch.hpp"

namespace pk {

void launch_dist_fast12(const SolveArgs& a, hipStream_t st);

template <int G, int RPL>
static void launch_one(const SolveArgs& a, int method, hipStream_t st) {
  const long long rpb = 256 / G;
  const long long nblk = (a.B + rpb - 1) / rpb;
  if (method == PK_METHOD_LRP8) hipLaunchKernelGGL((dist_fast_kernel<G, RPL, PK_METHOD_LRP8>), dim3((unsigned)nblk), dim3(256), 0, st, a);
  else                          hipLaunchKernelGGL((dist_fast_kernel<G, RPL, PK_METHOD_RODAS4>), dim3((unsigned)nblk), dim3(256), 0, st, a);
}

// lanes per replica x site rows per lane, chosen so that G * RPL >= n_sites with few idle slots
void launch_dist_fast(const SolveArgs& a, int method, hipStream_t st) {
  if (method == PK_METHOD_LRP12) { launch_dist_fast12(a, st); return; }
  const int n = a.n_sites;
  if (n <= 4) launch_one<4, 1>(a, method, st);
  else if (n <= 8) launch_one<4, 2>(a, method, st);
  else if (n <= 16) launch_one<4, 4>(a, method, st);       // 4 x 4 beats 8 x 2 by 1.4x (fewer reduction levels, less shadow-row redundancy)
  else if (n <= 32) launch_one<8, 4>(a, method, st);
  else launch_one<16, 4>(a, method, st);
}

}  // namespace pk
