// Instantiations + launcher of the thread-per-replica kernels (pk_tpr.hpp): LRP12 only (the default method).
#include "pk_tpr.hpp"
#include "pk_tpr_rand.hpp"
#include "pk_launch.hpp"
#include <atomic>
#include <cstdlib>

namespace pk {

// These kernels need 69-106 KB of dynamic LDS: above the 64 KB a kernel gets by default, so the limit is raised per (kernel, DEVICE) --
// a process may hold contexts on several GPUs and the attribute belongs to the device that is current when it is set.
template <int MODEL, int NS, bool STAGE = false>
static hipError_t launch_tpr_one(const SolveArgs& a, hipStream_t st) {
  const long long nblk = (a.B + 255) / 256;
  constexpr size_t lds = tpr_lds_bytes<NS, STAGE>();
  static std::atomic<uint64_t> ready{0};                      // bit d: attribute set on device d (devices >= 64: set on every launch)
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = dev < 64 ? (1ull << dev) : 0;
  if (!(ready.load(std::memory_order_acquire) & bit) || !bit) {
    e = hipFuncSetAttribute((const void*)tpr_kernel<MODEL, NS, PK_METHOD_LRP12, STAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    ready.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((tpr_kernel<MODEL, NS, PK_METHOD_LRP12, STAGE>), dim3((unsigned)nblk), dim3(256), lds, st, a);
  return hipSuccess;
}
// PK_TPR_STAGE=1 (read once): trajectories of the smallest systems go through the thread-private line buffer (pk_tpr.hpp, STAGE).
// Measured at distmod n = 4, B = 524 288 (profiles/r03_d_tpr_*): staged WRITE_SIZE = 1.001 x the algorithmic bytes at 0.597 ms per launch;
// direct 16-byte stores 1.35 x at 0.499 ms (round 2's 8-byte stores: 1.44 x, 0.512 ms).  The kernel is bound by VALU issue, not by HBM (14 % of
// peak), and the buffer's 32 KB of LDS cost an occupancy step (3 -> 2 workgroups per CU): the faster path is the default, the lean one
// is there for callers who share the HBM with something that needs it.
static bool tpr_stage(const SolveArgs& a) {
  static const bool on = [] { const char* v = getenv("PK_TPR_STAGE"); return v && v[0] == '1'; }();
  return on && a.sol != nullptr;
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
    if (n <= 4) return tpr_stage(a) ? launch_tpr_one<M_DIST, 4, true>(a, st) : launch_tpr_one<M_DIST, 4>(a, st);
    else if (n <= 8) return launch_tpr_one<M_DIST, 8>(a, st);
    else return launch_tpr_one<M_DIST, 12>(a, st);
  } else if (model == M_RAND) {
    if (n == 1) return launch_tpr_rand<1>(a, st);
    else if (n == 2) return launch_tpr_rand<2>(a, st);
    else return launch_tpr_rand<3>(a, st);
  } else {
    if (n <= 4) return tpr_stage(a) ? launch_tpr_one<M_SUCC, 4, true>(a, st) : launch_tpr_one<M_SUCC, 4>(a, st);
    else if (n <= 8) return launch_tpr_one<M_SUCC, 8>(a, st);
    else return launch_tpr_one<M_SUCC, 14>(a, st);
  }
}

}  // namespace pk
