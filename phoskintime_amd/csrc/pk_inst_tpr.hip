// Instantiations + launcher of the thread-per-replica kernels (pk_tpr.hpp): LRP12 only (the default method).
#include "pk_tpr.hpp"
#include "pk_tpr_rand.hpp"
#include "pk_launch.hpp"
#include <atomic>

namespace pk {

// These kernels need 69-106 KB of dynamic LDS: above the 64 KB a kernel gets by default, so the limit is raised per (kernel, DEVICE) --
// a process may hold contexts on several GPUs and the attribute belongs to the device that is current when it is set.
template <int MODEL, int NS>
static hipError_t launch_tpr_one(const SolveArgs& a, hipStream_t st) {
  const long long nblk = (a.B + 255) / 256;
  constexpr size_t lds = tpr_lds_bytes<NS>();
  static std::atomic<uint64_t> ready{0};                      // bit d: attribute set on device d (devices >= 64: set on every launch)
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = dev < 64 ? (1ull << dev) : 0;
  if (!(ready.load(std::memory_order_acquire) & bit) || !bit) {
    e = hipFuncSetAttribute((const void*)tpr_kernel<MODEL, NS, PK_METHOD_LRP12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    ready.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((tpr_kernel<MODEL, NS, PK_METHOD_LRP12>), dim3((unsigned)nblk), dim3(256), lds, st, a);
  return hipSuccess;
}

template <int NB>
static hipError_t launch_tpr_rand(const SolveArgs& a, hipStream_t st) {
  const long long nblk = (a.B + 255) / 256;
  constexpr size_t lds = tpr_rand_lds_bytes<NB>();
  static_assert(lds <= 64 * 1024, "fits the default dynamic-LDS limit");
  hipLaunchKernelGGL((tpr_rand_kernel<NB, PK_METHOD_LRP12>), dim3((unsigned)nblk), dim3(256), lds, st, a);
  return hipSuccess;
}

// true if a thread-per-replica kernel exists for (model, n_sites)
bool tpr_available(int model, int n_sites) {
  if (model == M_DIST) return n_sites <= 12;
  if (model == M_SUCC) return n_sites <= 14;
  return n_sites <= 3;                                       // random model: 2^n <= 8 coupled rows in one lane
}

hipError_t launch_tpr(const SolveArgs& a, int model, hipStream_t st) {
  const int n = a.n_sites;
  if (model == M_DIST) {
    if (n <= 4) return launch_tpr_one<M_DIST, 4>(a, st);
    else if (n <= 8) return launch_tpr_one<M_DIST, 8>(a, st);
    else return launch_tpr_one<M_DIST, 12>(a, st);
  } else if (model == M_RAND) {
    if (n == 1) return launch_tpr_rand<1>(a, st);
    else if (n == 2) return launch_tpr_rand<2>(a, st);
    else return launch_tpr_rand<3>(a, st);
  } else {
    if (n <= 4) return launch_tpr_one<M_SUCC, 4>(a, st);
    else if (n <= 8) return launch_tpr_one<M_SUCC, 8>(a, st);
    else return launch_tpr_one<M_SUCC, 14>(a, st);
  }
}

}  // namespace pk
