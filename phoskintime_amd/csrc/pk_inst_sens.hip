// Instantiations + launcher of the forward-sensitivity kernels (pk_sens.hpp): LRP12, one column per lane.
#include "pk_sens.hpp"
#include "pk_launch.hpp"

namespace pk {

// sizes with a sensitivity kernel: distmod / succmod n <= 14 (S <= 16 rows in one lane, 1 + P = 5 + 2 n <= 64 columns in one wave),
// randmod n <= 3 (2^n <= 8 coupled rows inverted in registers; 1 + P <= 16 columns) and n = 4, 5 (the inverse shared by the group in LDS)
bool sens_available(int model, int n_sites) {
  if (model == M_RAND) return n_sites <= 5;
  return n_sites <= 14;
}

template <class Sys, int GP>
static hipError_t launch_sens_one(const SensArgs& a, hipStream_t st) {
  constexpr int NG = 64 / GP;
  const long long nblk = (a.s.B + NG - 1) / NG;
  constexpr size_t lds = sens_lds_bytes<Sys, GP>();
  static_assert(lds <= 64 * 1024, "fits the default dynamic-LDS limit");
  hipLaunchKernelGGL((sens_kernel<Sys, GP>), dim3((unsigned)nblk), dim3(64), lds, st, a);
  return hipGetLastError();
}

template <int MODEL>
static hipError_t launch_sens_chain(const SensArgs& a, hipStream_t st) {
  const int n = a.s.n_sites;                                 // columns: 1 + P = 5 + 2 n
  if (n <= 1) return launch_sens_one<ChainSys<MODEL, 1>, 8>(a, st);
  if (n <= 3) return launch_sens_one<ChainSys<MODEL, 3>, 16>(a, st);
  if (n <= 5) return launch_sens_one<ChainSys<MODEL, 5>, 16>(a, st);
  if (n <= 9) return launch_sens_one<ChainSys<MODEL, 9>, 32>(a, st);
  if (n <= 13) return launch_sens_one<ChainSys<MODEL, 13>, 32>(a, st);
  return launch_sens_one<ChainSys<MODEL, 14>, 64>(a, st);
}

hipError_t launch_sens(const SensArgs& a, int model, hipStream_t st) {
  if (model == M_DIST) return launch_sens_chain<M_DIST>(a, st);
  if (model == M_SUCC) return launch_sens_chain<M_SUCC>(a, st);
  const int n = a.s.n_sites;
  if (n == 1) return launch_sens_one<CubeSys<1>, 8>(a, st);          // 1 + P = 7
  if (n == 2) return launch_sens_one<CubeSys<2>, 16>(a, st);         // 10
  if (n == 3) return launch_sens_one<CubeSys<3>, 16>(a, st);         // 15
  if (n == 4) return launch_sens_one<CubeLdsSys<4, 32>, 32>(a, st);  // 1 + P = 24 columns, 17 rows
  return launch_sens_one<CubeLdsSys<5, 64>, 64>(a, st);              // 41 columns, 33 rows
}

}  // namespace pk
