// Instantiations + launcher of the random-model throughput kernel (pk_rand_fast.hpp).
#include "pk_rand_fast.hpp"
#include <cstring>
#include <cstdlib>
#include "pk_launch.hpp"

namespace pk {

template <int NB>
static void launch_nb(const SolveArgs& a, int method, hipStream_t st) {
  constexpr int G = (1 << NB) < 4 ? 4 : (1 << NB);
  const long long rpb = 256 / G;
  const long long nblk = (a.B + rpb - 1) / rpb;
  if (method == PK_METHOD_LRP12)     hipLaunchKernelGGL((rand_fast_kernel<NB, PK_METHOD_LRP12>), dim3((unsigned)nblk), dim3(256), 0, st, a);
  else if (method == PK_METHOD_LRP8) hipLaunchKernelGGL((rand_fast_kernel<NB, PK_METHOD_LRP8>), dim3((unsigned)nblk), dim3(256), 0, st, a);
  else                          hipLaunchKernelGGL((rand_fast_kernel<NB, PK_METHOD_RODAS4>), dim3((unsigned)nblk), dim3(256), 0, st, a);
}

// separate translation units (their unrolled Gauss-Jordan eliminations take minutes to compile: n = 6 one unit per method)
void launch_rand_fastr(const SolveArgs& a, int method, hipStream_t st);        // n = 3, 4, 5: several bit-mask rows per lane (pk_rand_fastr.hpp)
void launch_rand_fast6_lrp12(const SolveArgs& a, hipStream_t st);
void launch_rand_fast6_lrp8(const SolveArgs& a, hipStream_t st);
void launch_rand_fast6_rodas4(const SolveArgs& a, hipStream_t st);

void launch_rand_fast(const SolveArgs& a, int method, hipStream_t st) {
  // dev A/B: PK_RAND_ROWS=1 keeps the one-row-per-lane kernels for n = 3, 4
  static const bool one_row = getenv("PK_RAND_ROWS") && !strcmp(getenv("PK_RAND_ROWS"), "1");
  if (!one_row && (a.n_sites == 3 || a.n_sites == 4)) { launch_rand_fastr(a, method, st); return; }
  switch (a.n_sites) {
    case 1: launch_nb<1>(a, method, st); break;
    case 2: launch_nb<2>(a, method, st); break;
    case 3: launch_nb<3>(a, method, st); break;
    case 4: launch_nb<4>(a, method, st); break;
    case 6:                                                // 64 masks, one wave per replica, v_readlane broadcasts
      if (method == PK_METHOD_LRP12) launch_rand_fast6_lrp12(a, st);
      else if (method == PK_METHOD_LRP8) launch_rand_fast6_lrp8(a, st);
      else launch_rand_fast6_rodas4(a, st);
      break;
    default: launch_rand_fastr(a, method, st); break;     // n = 5: two rows per lane, 16-lane groups (DPP broadcasts; one row per lane: 1.8x slower)
  }
}

// ---- n = 6 (S = 65 states > one wave): right-hand side and Jacobian with one thread per (replica, row); y is read from memory.
__global__ void rand_rhs_wide_kernel(const double* __restrict__ theta, const double* __restrict__ y, double* __restrict__ dydt,
                                     const long long B, const int n, const int S, const int P) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= B * S) return;
  const long long rep = gid / S;
  const int row = (int)(gid - rep * S);
  const RowCoef c = load_row<M_RAND>(theta + rep * P, n, S, row);
  const double* yr = y + rep * S;
  double f = __builtin_fma(c.dg, yr[row], c.bias);
  if (row >= 1) {
    const int m = row - 1;
    f = __builtin_fma(c.c2, yr[0], f);
    for (int j = 0; j < n; ++j) {
      const int bit = 1 << j;
      f = __builtin_fma((m & bit) ? c.c1 : 1.0, yr[(m ^ bit) + 1], f);
    }
  }
  dydt[gid] = f;
}

__global__ void rand_jac_wide_kernel(const double* __restrict__ theta, double* __restrict__ J, const long long B, const int n,
                                     const int S, const int P) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= B * S) return;
  const long long rep = gid / S;
  const int row = (int)(gid - rep * S);
  const RowCoef c = load_row<M_RAND>(theta + rep * P, n, S, row);
  double* Jr = J + gid * S;
  for (int j = 0; j < S; ++j) Jr[j] = 0.0;
  Jr[row] = c.dg;
  if (row >= 1) {
    const int m = row - 1;
    if (row == 1) Jr[0] = c.c2;
    for (int j = 0; j < n; ++j) {
      const int bit = 1 << j;
      Jr[(m ^ bit) + 1] = (m & bit) ? c.c1 : 1.0;
    }
  }
}

void launch_rand_rhs_wide(const double* theta, const double* y, double* dydt, long long B, int n, int S, int P, hipStream_t st) {
  const long long nblk = (B * S + 255) / 256;
  hipLaunchKernelGGL(rand_rhs_wide_kernel, dim3((unsigned)nblk), dim3(256), 0, st, theta, y, dydt, B, n, S, P);
}
void launch_rand_jac_wide(const double* theta, double* J, long long B, int n, int S, int P, hipStream_t st) {
  const long long nblk = (B * S + 255) / 256;
  hipLaunchKernelGGL(rand_jac_wide_kernel, dim3((unsigned)nblk), dim3(256), 0, st, theta, J, B, n, S, P);
}

}  // namespace pk
