// Instantiations of the random-model kernels with several bit-mask rows per lane (pk_rand_fastr.hpp): n = 3, 4, 5.
#include "pk_rand_fastr.hpp"
#include "pk_launch.hpp"

namespace pk {

template <int NB, int RPL>
static void launch_r(const SolveArgs& a, int method, hipStream_t st) {
  constexpr int G = (1 << NB) / RPL;
  const long long rpb = 256 / G;
  const long long nblk = (a.B + rpb - 1) / rpb;
  constexpr size_t lds = rand_fastr_lds_bytes<RPL>();
  if (method == PK_METHOD_LRP12)     hipLaunchKernelGGL((rand_fastr_kernel<NB, RPL, PK_METHOD_LRP12>), dim3((unsigned)nblk), dim3(256), lds, st, a);
  else if (method == PK_METHOD_LRP8) hipLaunchKernelGGL((rand_fastr_kernel<NB, RPL, PK_METHOD_LRP8>), dim3((unsigned)nblk), dim3(256), lds, st, a);
  else                               hipLaunchKernelGGL((rand_fastr_kernel<NB, RPL, PK_METHOD_RODAS4>), dim3((unsigned)nblk), dim3(256), lds, st, a);
}

void launch_rand_fastr(const SolveArgs& a, int method, hipStream_t st) {
  if (a.n_sites == 3) launch_r<3, 2>(a, method, st);
  else if (a.n_sites == 4) launch_r<4, 4>(a, method, st);
  else launch_r<5, 2>(a, method, st);
}

}  // namespace pk
