// Instantiations + launcher of the order-4 network integrator (pk_network_solve_ark.hpp): topologies 0 / 1 / 4, site classes 4 / 6 / 8.
#include "pk_network_solve_ark.hpp"
#include <atomic>
#include <cstdlib>

namespace pk {

static int site_class(int max_sites) { return max_sites <= 4 ? 4 : max_sites <= 6 ? 6 : 8; }

// rows of a block vector: 2 + site class (topologies 0 / 1 / 4), 1 + 2^NB with NB = 2 or 3 (combinatorial)
size_t net_ark_lds_bytes(const NetDev& n, int nnzT, int max_sites, int threads) {
  const int rows = n.model == 2 ? 1 + (max_sites <= 2 ? 4 : 8) : 2 + site_class(max_sites);
  // PK_ARK_LDS_PAD (bytes, dev switch, read once): unused LDS per workgroup, to measure the kernel at fewer resident workgroups per CU
  static const size_t pad = [] { const char* v = getenv("PK_ARK_LDS_PAD"); return v ? (size_t)atol(v) : (size_t)0; }();
  return net_solve_ark_lds_bytes(n, nnzT, rows, threads) + pad;
}

template <int M, int MS>
static hipError_t launch_one(const NetDev& n, const NetSolveArgs& a, long long B, int threads, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {                                  // beyond the default dynamic-LDS limit: raise it per (kernel, device)
    static std::atomic<uint64_t> ready{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = dev < 64 ? (1ull << dev) : 0;
    if (!bit || !(ready.load(std::memory_order_acquire) & bit)) {
      e = hipFuncSetAttribute((const void*)net_solve_ark_kernel<M, MS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      ready.fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL((net_solve_ark_kernel<M, MS>), dim3((unsigned)B), dim3(threads), lds, st, n, a);
  return hipSuccess;
}

template <int NB, bool EXACT>
static hipError_t launch_comb(const NetDev& n, const NetSolveArgs& a, long long B, int threads, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    static std::atomic<uint64_t> ready{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = dev < 64 ? (1ull << dev) : 0;
    if (!bit || !(ready.load(std::memory_order_acquire) & bit)) {
      e = hipFuncSetAttribute((const void*)net_solve_ark2_kernel<NB, EXACT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      ready.fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL((net_solve_ark2_kernel<NB, EXACT>), dim3((unsigned)B), dim3(threads), lds, st, n, a);
  return hipSuccess;
}

hipError_t launch_net_ark(const NetDev& n, const NetSolveArgs& a, int max_sites, long long B, int threads, size_t lds, hipStream_t st) {
  if (n.model == 2) {
    // PK_ARK2_EXACT=0 (read once): round 2's approximate factorisation as the implicit operator (the A/B twin of the exact block solve)
    static const bool exact = [] { const char* v = getenv("PK_ARK2_EXACT"); return !(v && v[0] == '0'); }();
    if (max_sites <= 2) return exact ? launch_comb<2, true>(n, a, B, threads, lds, st) : launch_comb<2, false>(n, a, B, threads, lds, st);
    return exact ? launch_comb<3, true>(n, a, B, threads, lds, st) : launch_comb<3, false>(n, a, B, threads, lds, st);
  }
  const int cls = site_class(max_sites);
#define PK_ARK(M)                                                                   \
  (cls == 4 ? launch_one<M, 4>(n, a, B, threads, lds, st) : cls == 6 ? launch_one<M, 6>(n, a, B, threads, lds, st) : launch_one<M, 8>(n, a, B, threads, lds, st))
  if (n.model == 0) return PK_ARK(0);
  if (n.model == 1) return PK_ARK(1);
  return PK_ARK(4);
#undef PK_ARK
}

}  // namespace pk
