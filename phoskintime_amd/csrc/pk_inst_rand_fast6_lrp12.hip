// Instantiation of the one-wave-per-replica random-model kernel at n = 6 (pk_rand_fast.hpp), method PK_METHOD_LRP12: its own translation unit
// (the unrolled 64-row elimination is the longest compile of the library).
#include "pk_rand_fast.hpp"
#include "pk_launch.hpp"

namespace pk {

void launch_rand_fast6_lrp12(const SolveArgs& a, hipStream_t st) {
  const long long rpb = 256 / 64;
  const long long nblk = (a.B + rpb - 1) / rpb;
  hipLaunchKernelGGL((rand_fast_kernel<6, PK_METHOD_LRP12>), dim3((unsigned)nblk), dim3(256), 0, st, a);
}

}  // namespace pk
