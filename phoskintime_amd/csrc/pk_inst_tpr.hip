// Instantiations + launcher of the thread-per-replica kernels (pk_tpr.hpp): LRP12 only (the default method).
#include "pk_tpr.hpp"
#include "pk_tpr_rand.hpp"
#include "pk_launch.hpp"

namespace pk {

template <int MODEL, int NS>
static void launch_tpr_one(const SolveArgs& a, hipStream_t st) {
  const long long nblk = (a.B + 255) / 256;
  constexpr size_t lds = tpr_lds_bytes<NS>();
  static const bool once = [] {
    (void)hipFuncSetAttribute((const void*)tpr_kernel<MODEL, NS, PK_METHOD_LRP12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return true;
  }();
  (void)once;
  hipLaunchKernelGGL((tpr_kernel<MODEL, NS, PK_METHOD_LRP12>), dim3((unsigned)nblk), dim3(256), lds, st, a);
}

template <int NB>
static void launch_tpr_rand(const SolveArgs& a, hipStream_t st) {
  const long long nblk = (a.B + 255) / 256;
  constexpr size_t lds = tpr_rand_lds_bytes<NB>();
  hipLaunchKernelGGL((tpr_rand_kernel<NB, PK_METHOD_LRP12>), dim3((unsigned)nblk), dim3(256), lds, st, a);
}

// true if a thread-per-replica kernel exists for (model, n_sites)
bool tpr_available(int model, int n_sites) {
  if (model == M_DIST) return n_sites <= 12;
  if (model == M_SUCC) return n_sites <= 14;
  return n_sites <= 3;                                       // random model: 2^n <= 8 coupled rows in one lane
}

void launch_tpr(const SolveArgs& a, int model, hipStream_t st) {
  const int n = a.n_sites;
  if (model == M_DIST) {
    if (n <= 4) launch_tpr_one<M_DIST, 4>(a, st);
    else if (n <= 8) launch_tpr_one<M_DIST, 8>(a, st);
    else launch_tpr_one<M_DIST, 12>(a, st);
  } else if (model == M_RAND) {
    if (n == 1) launch_tpr_rand<1>(a, st);
    else if (n == 2) launch_tpr_rand<2>(a, st);
    else launch_tpr_rand<3>(a, st);
  } else {
    if (n <= 4) launch_tpr_one<M_SUCC, 4>(a, st);
    else if (n <= 8) launch_tpr_one<M_SUCC, 8>(a, st);
    else launch_tpr_one<M_SUCC, 14>(a, st);
  }
}

}  // namespace pk
