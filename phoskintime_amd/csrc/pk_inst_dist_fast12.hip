// LRP12 instantiations of the distributive-model throughput kernel (pk_dist_fast.hpp) -- the default method gets the fine-grained
// layout table: G lanes per replica x RPL site rows per lane with G * RPL >= n_sites and as few idle rows as possible.
//   * fewer lanes per replica = fewer DPP reduction levels per solve and less redundant work on the shadowed (R, P) rows;
//   * RPL >= 5 needs more than 256 VGPRs with everything in registers: those layouts park the once-per-step values (site rates) and the
//     once-per-output values (metric bookkeeping) in LDS (Parked<RPL, true>) and run two waves per SIMD.
// Measured, B = 65 536, theta ~ U(0, 20) (tools/gpu_bench_dev.py layouts): n = 14: 4x4 0.287 ms vs 8x2 0.410; n = 30: 4x8 parked 0.422 vs
// 8x4 0.528; n = 62: 8x8 parked 0.913 vs 16x4 1.206.
#include "pk_dist_fast.hpp"
#include "pk_launch.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace pk {

template <int G, int RPL>
static void launch_plain(const SolveArgs& a, hipStream_t st) {
  const long long rpb = 256 / G;
  const long long nblk = (a.B + rpb - 1) / rpb;
  hipLaunchKernelGGL((dist_fast_kernel<G, RPL, PK_METHOD_LRP12>), dim3((unsigned)nblk), dim3(256), 0, st, a);
}

template <int G, int RPL>
static void launch_parked(const SolveArgs& a, hipStream_t st) {
  const long long rpb = 256 / G;
  const long long nblk = (a.B + rpb - 1) / rpb;
  constexpr size_t lds = dist_fast_lds_bytes<RPL, true>();
  static const bool once = [] {
    (void)hipFuncSetAttribute((const void*)dist_fast_kernel<G, RPL, PK_METHOD_LRP12, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return true;
  }();
  (void)once;
  hipLaunchKernelGGL((dist_fast_kernel<G, RPL, PK_METHOD_LRP12, true, 2>), dim3((unsigned)nblk), dim3(256), lds, st, a);
}

void launch_dist_fast12(const SolveArgs& a, hipStream_t st) {
  const int n = a.n_sites;
  // dev A/B: PK_DIST_LAYOUT=8x4 forces the register-only 8 x 4 layout for 17 <= n <= 32
  static const bool force84 = getenv("PK_DIST_LAYOUT") && !strcmp(getenv("PK_DIST_LAYOUT"), "8x4");
  if (n <= 4) launch_plain<4, 1>(a, st);
  else if (n <= 8) launch_plain<4, 2>(a, st);
  else if (n <= 12) launch_plain<4, 3>(a, st);
  else if (n <= 16) launch_plain<4, 4>(a, st);
  else if (n <= 32 && force84) launch_plain<8, 4>(a, st);
  else if (n <= 20) launch_parked<4, 5>(a, st);
  else if (n <= 24) launch_parked<4, 6>(a, st);
  else if (n <= 28) launch_parked<4, 7>(a, st);
  else if (n <= 32) launch_parked<4, 8>(a, st);
  else if (n <= 40) launch_parked<8, 5>(a, st);
  else if (n <= 48) launch_parked<8, 6>(a, st);
  else if (n <= 56) launch_parked<8, 7>(a, st);
  else launch_parked<8, 8>(a, st);
}

}  // namespace pk
