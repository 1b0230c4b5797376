// Instantiations + launchers of the workgroup-per-replica kernels (pk_wide.hpp) for systems beyond one wavefront's lane groups.
#include "pk_wide.hpp"
#include "pk_rand_dense.hpp"
#include "pk_rand_level.hpp"
#include "pk_rand_parity.hpp"
#include "pk_rand_sens.hpp"
#include "pk_launch.hpp"
#include <atomic>
#include <cstdlib>

namespace pk {

namespace {
constexpr size_t kLdsMax = 160 * 1024;       // gfx950: 160 KB of LDS per workgroup

// raise the dynamic-LDS limit of `fn` on the current device once (per kernel and device)
template <class K>
hipError_t allow_lds(K fn, std::atomic<uint64_t>& ready) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = dev < 64 ? (1ull << dev) : 0;
  if (!bit || !(ready.load(std::memory_order_acquire) & bit)) {
    e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    if (e != hipSuccess) return e;
    ready.fetch_or(bit, std::memory_order_release);
  }
  return hipSuccess;
}
}  // namespace

// randmod n = 7 only: the 2^n x 2^n inverse must fit the register file of one workgroup (128 KB of a CU's 512 KB; n = 8 would need all of
// it).  PK_WIDE_RAND_DENSE=0 (read once) sends n = 7 back to the approximate-factorisation kernel (tests exercise both).
// randmod n = 8 (n = 6 on request): twisted block elimination over the popcount levels (pk_rand_level.hpp)
hipError_t launch_rand_level(const SolveArgs& a_in, hipStream_t st) {
  static const int dbg = [] { const char* v = getenv("PK_LEVEL_DEBUG"); return v ? atoi(v) : 0; }();     // dev timing switches (results are garbage when set)
  SolveArgs a = a_in;
  a.stage_form = dbg;
  if (a.n_sites == 8) return launch_rand_level_one<8>(a, st);
  if (a.n_sites == 6) return launch_rand_level_one<6>(a, st);
  return hipErrorInvalidValue;
}

// randmod n = 8: odd-popcount states eliminated exactly, the 128 x 128 even Schur complement inverted in registers (pk_rand_parity.hpp)
hipError_t launch_rand_parity(const SolveArgs& a, hipStream_t st, bool pinned_family) {
  if (a.n_sites == 8) {
    // PK_RAND_PARITY8_GRID=512 (dev, read once): the 16 x 32 thread grid (8 x 4 blocks per thread) instead of 16 x 16 (8 x 8 blocks)
    static const int grid8 = [] { const char* v = getenv("PK_RAND_PARITY8_GRID"); return v ? atoi(v) : 256; }();
    if (grid8 == 512) hipLaunchKernelGGL((rand_parity_kernel<8, 16, 32>), dim3((unsigned)a.B), dim3(512), rand_parity_lds_bytes(8), st, a);
    else              hipLaunchKernelGGL((rand_parity_kernel<8, 16, 16>), dim3((unsigned)a.B), dim3(256), rand_parity_lds_bytes(8), st, a);
  }
  else if (a.n_sites == 7) {
    // One wave per replica (8 x 8 lanes, 8 x 8 blocks) at large batches: 386-429 k replicas/s against 336-370 k of the 256-thread grid at
    // B = 1 024 ... 8 192; below that the grid finishes a launch sooner (1.13-1.30 ms against 2.0-2.3 ms at B = 1 ... 256).  The two sum in
    // different orders: a caller that pinned the kernel family (opts->kernel, sharded runs that must reproduce one-process bits) keeps
    // the grid at every size.  PK_RAND_PARITY7_TB=8 / 16 (dev, read once) forces one of them.
    static const int tb7 = [] { const char* v = getenv("PK_RAND_PARITY7_TB"); return v ? atoi(v) : 0; }();
    const bool one7 = tb7 == 8 || (tb7 != 16 && !pinned_family && a.B >= 1024);
    if (one7) hipLaunchKernelGGL((rand_parity_kernel<7, 8>), dim3((unsigned)a.B), dim3(64), rand_parity_lds_bytes(7), st, a);
    else      hipLaunchKernelGGL((rand_parity_kernel<7, 16>), dim3((unsigned)a.B), dim3(256), rand_parity_lds_bytes(7), st, a);
  }
  else if (a.n_sites == 6) {
    // PK_RAND_PARITY6_TB=16 (dev, read once): the 256-thread grid (2 x 2 blocks) instead of one wave per replica (8 x 8 lanes, 4 x 4 blocks)
    static const int tb6 = [] { const char* v = getenv("PK_RAND_PARITY6_TB"); return v ? atoi(v) : 8; }();
    const bool wide6 = tb6 == 16;
    // (measured and dropped: 4 x 8 lanes per replica, two replicas per wave -- 1.64-1.68 M replicas/s against 2.09-2.19 M)
    if (wide6) hipLaunchKernelGGL((rand_parity_kernel<6, 16>), dim3((unsigned)a.B), dim3(256), rand_parity_lds_bytes(6), st, a);
    else       hipLaunchKernelGGL((rand_parity_kernel<6, 8>), dim3((unsigned)a.B), dim3(64), rand_parity_lds_bytes(6), st, a);
  }
  else if (a.n_sites == 5)      // 4 x 4 lanes per replica, four replicas per wave
    hipLaunchKernelGGL((rand_parity_kernel<5, 4>), dim3((unsigned)((a.B + 3) / 4)), dim3(64), 4 * rand_parity_lds_bytes(5), st, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// forward sensitivities of randmod n = 6, 7: the parity-eliminated inverse serves eight columns per workgroup (pk_rand_sens.hpp)
hipError_t launch_rand_sens(const SensArgs& a, hipStream_t st) {
  // n = 6: one wave per column chunk (8 x 8 lanes, 4 x 4 blocks; 34.7 against 44.5 ms per 1 024 Jacobians on the 256-thread grid, same box).
  // PK_RAND_SENS6_TB=16 (dev, read once): the 256-thread grid
  static const int tb6 = [] { const char* v = getenv("PK_RAND_SENS6_TB"); return v ? atoi(v) : 8; }();
  if (a.s.n_sites == 6) return tb6 == 8 ? launch_rand_sens_one<6, 8>(a, st) : launch_rand_sens_one<6, 16>(a, st);
  if (a.s.n_sites == 7) return launch_rand_sens_one<7>(a, st);
  return hipErrorInvalidValue;
}

bool rand_dense_available(int n_sites) {
  static const bool on = [] { const char* v = getenv("PK_WIDE_RAND_DENSE"); return !(v && v[0] == '0'); }();
  return on && n_sites == 7;
}
hipError_t launch_rand_dense(const SolveArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((rand_dense_kernel<7>), dim3((unsigned)a.B), dim3(rand_dense_threads<7>()), rand_dense_lds_bytes(a.n_sites), st, a);
  return hipGetLastError();
}

bool wide_chain_fits(int S, int n) { return wide_chain_lds_bytes(S, n) <= kLdsMax; }

hipError_t launch_wide_chain(const SolveArgs& a, int model, hipStream_t st) {
  const size_t lds = wide_chain_lds_bytes(a.S, a.n_sites);
  const int nt = a.S <= 64 ? 64 : a.S <= 128 ? 128 : 256;
  hipError_t e;
  if (model == M_DIST) {
    static std::atomic<uint64_t> ready{0};
    if ((e = allow_lds(wide_chain_kernel<M_DIST>, ready)) != hipSuccess) return e;
    hipLaunchKernelGGL((wide_chain_kernel<M_DIST>), dim3((unsigned)a.B), dim3(nt), lds, st, a);
  } else {
    static std::atomic<uint64_t> ready{0};
    if ((e = allow_lds(wide_chain_kernel<M_SUCC>, ready)) != hipSuccess) return e;
    hipLaunchKernelGGL((wide_chain_kernel<M_SUCC>), dim3((unsigned)a.B), dim3(nt), lds, st, a);
  }
  return hipSuccess;
}

// scratch == nullptr: the vectors live in LDS (caller checked wide_rand_in_lds); else one row of `stride` doubles per replica in HBM
// the order-4 additive method is the default (PK_WIDE_RAND_ROSW=1, read once, selects round-2's first version: ROS34PW2-W)
// PK_WIDE_RAND_DRIFT=0 (read once) switches the drift removal of the n-cube kernel off (A/B runs, tests of the plain path)
static int wide_rand_drift() { static const int on = [] { const char* v = getenv("PK_WIDE_RAND_DRIFT"); return (v && v[0] == '0') ? 0 : 1; }(); return on; }
static bool wide_rand_ark() { static const bool rosw = [] { const char* v = getenv("PK_WIDE_RAND_ROSW"); return v && atoi(v) == 1; }(); return !rosw; }
bool wide_rand_in_lds(int n) { return n <= 16 && wide_rand_lds_bytes(n, true, wide_rand_ark()) <= kLdsMax; }
size_t wide_rand_scratch_bytes(int n, long long B) { return wide_rand_in_lds(n) ? 0 : (size_t)B * wide_rand_scratch_doubles(n, wide_rand_ark()) * sizeof(double); }

template <bool ARK>
static hipError_t launch_wide_rand_m(const SolveArgs& a, double* scratch, hipStream_t st) {
  const int n = a.n_sites;
  const int NM = 1 << n;
  const int nt = NM <= 128 ? 64 : NM <= 256 ? 128 : 256;
  hipError_t e;
  if (!scratch) {
    static std::atomic<uint64_t> ready{0};
    if ((e = allow_lds(wide_rand_kernel<true, ARK>, ready)) != hipSuccess) return e;
    hipLaunchKernelGGL((wide_rand_kernel<true, ARK>), dim3((unsigned)a.B), dim3(nt), wide_rand_lds_bytes(n, true, ARK), st, a, (double*)nullptr, (size_t)0, wide_rand_drift());
  } else {
    hipLaunchKernelGGL((wide_rand_kernel<false, ARK>), dim3((unsigned)a.B), dim3(256), wide_rand_lds_bytes(n, false, ARK), st, a, scratch, wide_rand_scratch_doubles(n, ARK), wide_rand_drift());
  }
  return hipSuccess;
}
hipError_t launch_wide_rand(const SolveArgs& a, double* scratch, hipStream_t st) {
  return wide_rand_ark() ? launch_wide_rand_m<true>(a, scratch, st) : launch_wide_rand_m<false>(a, scratch, st);
}

// steady state beyond 64 states; returns hipErrorInvalidValue when the system does not fit LDS (randmod n_sites >= 13)
hipError_t launch_wide_steady(int model, const double* theta, double* yss, int32_t* status, long long B, int n, int S, int P, hipStream_t st) {
  hipError_t e;
  if (model == M_RAND) {
    const size_t lds = wide_steady_rand_lds_bytes(n);
    if (lds > kLdsMax) return hipErrorInvalidValue;
    static std::atomic<uint64_t> ready{0};
    if ((e = allow_lds(wide_steady_rand_kernel, ready)) != hipSuccess) return e;
    hipLaunchKernelGGL(wide_steady_rand_kernel, dim3((unsigned)B), dim3(S <= 129 ? 64 : 256), lds, st, theta, yss, status, B, n, S, P);
    return hipSuccess;
  }
  const size_t lds = wide_steady_chain_lds_bytes(S);
  if (lds > kLdsMax) return hipErrorInvalidValue;
  const int nt = S <= 64 ? 64 : S <= 128 ? 128 : 256;
  if (model == M_DIST) {
    static std::atomic<uint64_t> ready{0};
    if ((e = allow_lds(wide_steady_chain_kernel<M_DIST>, ready)) != hipSuccess) return e;
    hipLaunchKernelGGL((wide_steady_chain_kernel<M_DIST>), dim3((unsigned)B), dim3(nt), lds, st, theta, yss, status, B, n, S, P);
  } else {
    static std::atomic<uint64_t> ready{0};
    if ((e = allow_lds(wide_steady_chain_kernel<M_SUCC>, ready)) != hipSuccess) return e;
    hipLaunchKernelGGL((wide_steady_chain_kernel<M_SUCC>), dim3((unsigned)B), dim3(nt), lds, st, theta, yss, status, B, n, S, P);
  }
  return hipSuccess;
}

void launch_chain_rhs_wide(int model, const double* theta, const double* y, double* dydt, long long B, int n, int S, int P, hipStream_t st) {
  const long long nblk = (B * S + 255) / 256;
  if (model == M_DIST) hipLaunchKernelGGL((chain_rhs_wide_kernel<M_DIST>), dim3((unsigned)nblk), dim3(256), 0, st, theta, y, dydt, B, n, S, P);
  else                 hipLaunchKernelGGL((chain_rhs_wide_kernel<M_SUCC>), dim3((unsigned)nblk), dim3(256), 0, st, theta, y, dydt, B, n, S, P);
}
void launch_chain_jac_wide(int model, const double* theta, double* J, long long B, int n, int S, int P, hipStream_t st) {
  const long long nblk = (B * S + 255) / 256;
  if (model == M_DIST) hipLaunchKernelGGL((chain_jac_wide_kernel<M_DIST>), dim3((unsigned)nblk), dim3(256), 0, st, theta, J, B, n, S, P);
  else                 hipLaunchKernelGGL((chain_jac_wide_kernel<M_SUCC>), dim3((unsigned)nblk), dim3(256), 0, st, theta, J, B, n, S, P);
}

}  // namespace pk
