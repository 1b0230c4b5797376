// Instantiations + launcher of the distributive-model throughput kernel (pk_dist_fast.hpp).
#include "pk_dist_fast.hpp"
#include "pk_launch.hpp"
#include <cstdlib>

namespace pk {

template <int G, int RPL>
static void launch_one(const SolveArgs& a, int method, hipStream_t st) {
  const long long rpb = 256 / G;
  const long long nblk = (a.B + rpb - 1) / rpb;
  if (method == PK_METHOD_LRP12)     hipLaunchKernelGGL((dist_fast_kernel<G, RPL, PK_METHOD_LRP12>), dim3((unsigned)nblk), dim3(256), 0, st, a);
  else if (method == PK_METHOD_LRP8) hipLaunchKernelGGL((dist_fast_kernel<G, RPL, PK_METHOD_LRP8>), dim3((unsigned)nblk), dim3(256), 0, st, a);
  else                          hipLaunchKernelGGL((dist_fast_kernel<G, RPL, PK_METHOD_RODAS4>), dim3((unsigned)nblk), dim3(256), 0, st, a);
}

// 4 lanes x 8 rows with the once-per-step values parked in LDS (two waves / SIMD): LRP12 only (the default method)
static void launch_48_parked(const SolveArgs& a, hipStream_t st) {
  const long long rpb = 256 / 4;
  const long long nblk = (a.B + rpb - 1) / rpb;
  constexpr size_t lds = dist_fast_lds_bytes<8, true>();
  static const bool once = [] {
    (void)hipFuncSetAttribute((const void*)dist_fast_kernel<4, 8, PK_METHOD_LRP12, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return true;
  }();
  (void)once;
  hipLaunchKernelGGL((dist_fast_kernel<4, 8, PK_METHOD_LRP12, true>), dim3((unsigned)nblk), dim3(256), lds, st, a);
}

// lanes per replica x site rows per lane, chosen so that G * RPL >= n_sites with the fewest idle slots
void launch_dist_fast(const SolveArgs& a, int method, hipStream_t st) {
  const int n = a.n_sites;
  if (n <= 4) launch_one<4, 1>(a, method, st);
  else if (n <= 8) launch_one<4, 2>(a, method, st);
  else if (n <= 16) launch_one<8, 2>(a, method, st);
  else if (n <= 32) {
    static const bool parked = !(getenv("PK_DIST_PARK") && atoi(getenv("PK_DIST_PARK")) == 0);      // PK_DIST_PARK=0: A/B against 8 x 4
    if (parked && method == PK_METHOD_LRP12) launch_48_parked(a, st); else launch_one<8, 4>(a, method, st);
  }        // 4 lanes x 8 rows measured equal (65.9 vs 65.2 M/s) at half the occupancy
  else launch_one<16, 4>(a, method, st);
}

}  // namespace pk
