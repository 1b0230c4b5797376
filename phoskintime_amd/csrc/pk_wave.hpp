// pk_wave.hpp -- sub-wavefront ("replica group") primitives for gfx950 (wave64).
//
// A replica (one parameter vector = one ODE system) is owned by a group of G consecutive lanes of a
// wavefront, G in {8, 16, 32, 64}; lane r of the group owns state / matrix row r.  64 / G replicas share
// a wavefront, so S = 32 runs two replicas per wave and S = 6 runs eight.  All cross-lane traffic stays
// inside the group and never touches LDS memory:
//   * G == 64 : v_readlane_b32 (value lands in SGPRs and feeds v_fma_f64 as a scalar operand)
//   * G <  64 : ds_swizzle_b32 in bit-mask mode (LDS crossbar, no LDS storage, no address VGPR)
//   * runtime source lane: ds_bpermute_b32
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>

namespace pk {

// ---- compile-time loops (register arrays must only ever be indexed by constants: guide rule 20) ----
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// broadcast lane K of each group to every lane of that group (K compile-time)
template <int G, int K>
__device__ __forceinline__ double bcast(double v) {
  static_assert(K >= 0 && K < G, "lane out of group");
  int lo = __double2loint(v), hi = __double2hiint(v);
  if constexpr (G == 64) {
    lo = __builtin_amdgcn_readlane(lo, K);
    hi = __builtin_amdgcn_readlane(hi, K);
  } else {
    // bit-mask mode: src_lane = ((lane & and_mask) | or_mask) ^ xor_mask within each 32-lane half
    constexpr int pat = ((32 - G) & 0x1f) | (K << 5);
    lo = __builtin_amdgcn_ds_swizzle(lo, pat);
    hi = __builtin_amdgcn_ds_swizzle(hi, pat);
  }
  return __hiloint2double(hi, lo);
}

// value of group lane `src` (runtime, may differ per lane)
template <int G>
__device__ __forceinline__ double gshfl(double v, int src, int lane) {
  const int addr = ((lane & ~(G - 1)) | (src & (G - 1))) << 2;
  int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
  int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

template <int MASK>
__device__ __forceinline__ double xor_lane(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  if constexpr (MASK < 32) {
    constexpr int pat = 0x1f | (MASK << 10);
    lo = __builtin_amdgcn_ds_swizzle(lo, pat);
    hi = __builtin_amdgcn_ds_swizzle(hi, pat);
  } else {
    const int addr = (lane ^ 32) << 2;
    lo = __builtin_amdgcn_ds_bpermute(addr, lo);
    hi = __builtin_amdgcn_ds_bpermute(addr, hi);
  }
  return __hiloint2double(hi, lo);
}

// all-reduce over the group (xor butterfly: every lane ends with the bit-identical result)
template <int G>
__device__ __forceinline__ double gsum(double v, int lane) {
  if constexpr (G >= 2)  v += xor_lane<1>(v, lane);
  if constexpr (G >= 4)  v += xor_lane<2>(v, lane);
  if constexpr (G >= 8)  v += xor_lane<4>(v, lane);
  if constexpr (G >= 16) v += xor_lane<8>(v, lane);
  if constexpr (G >= 32) v += xor_lane<16>(v, lane);
  if constexpr (G >= 64) v += xor_lane<32>(v, lane);
  return v;
}
template <int G>
__device__ __forceinline__ double gmax(double v, int lane) {
  // NaN-propagating max: a NaN anywhere in the group must surface (fmax would swallow it)
  auto mx = [](double a, double b) { return (a > b || a != a) ? a : b; };
  if constexpr (G >= 2)  v = mx(v, xor_lane<1>(v, lane));
  if constexpr (G >= 4)  v = mx(v, xor_lane<2>(v, lane));
  if constexpr (G >= 8)  v = mx(v, xor_lane<4>(v, lane));
  if constexpr (G >= 16) v = mx(v, xor_lane<8>(v, lane));
  if constexpr (G >= 32) v = mx(v, xor_lane<16>(v, lane));
  if constexpr (G >= 64) v = mx(v, xor_lane<32>(v, lane));
  return v;
}

// ------------------------------------------------------------------ dense in-register LU
// Row-per-lane storage: lane `row` holds a[0..G) = row `row` of W.  No pivoting: for every model on this
// path W = g I - J with J a compartmental (Metzler, column-sum <= 0) matrix, so W is a column-diagonally-
// dominant M-matrix and Gaussian elimination without pivoting is backward stable.
// On exit a[] holds L (unit lower, multipliers) below the diagonal and U on/above it; dinv = 1 / U[row][row].
template <int G>
__device__ __forceinline__ void lu_factor(double (&a)[G], const int row, double& dinv) {
  static_for<G>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const double piv = bcast<G, k>(a[k]);
    const double rp = 1.0 / piv;
    if (row == k) dinv = rp;
    if constexpr (k + 1 < G) {
      const double l = (row > k) ? a[k] * rp : 0.0;   // l == 0 leaves rows <= k untouched
      if (row > k) a[k] = l;
      static_for<G - 1 - k>([&](auto jc) {
        constexpr int j = k + 1 + decltype(jc)::value;
        const double u = bcast<G, k>(a[j]);
        a[j] = __builtin_fma(-l, u, a[j]);
      });
    }
  });
}

// x <- W^{-1} x using the factors above (x distributed one element per lane)
template <int G>
__device__ __forceinline__ double lu_solve(const double (&a)[G], const int row, const double dinv, double x) {
  // forward: L z = x (unit lower)
  static_for<G - 1>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const double xk = bcast<G, k>(x);
    if (row > k) x = __builtin_fma(-a[k], xk, x);
  });
  // backward: U x = z ; lane k's value is final once every k' > k has been eliminated
  static_for<G - 1>([&](auto kc) {
    constexpr int k = G - 1 - decltype(kc)::value;
    const double xk = bcast<G, k>(x * dinv);
    if (row < k) x = __builtin_fma(-a[k], xk, x);
  });
  return x * dinv;
}

}  // namespace pk
