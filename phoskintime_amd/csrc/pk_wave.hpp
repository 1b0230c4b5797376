// pk_wave.hpp -- sub-wavefront ("replica group") primitives for gfx950 (wave64).
//
// A replica (one parameter vector = one ODE system) is owned by a group of G consecutive lanes of a
// wavefront, G in {8, 16, 32, 64}; lane r of the group owns state / matrix row r.  64 / G replicas share
// a wavefront, so S = 32 runs two replicas per wave and S = 6 runs eight.  All cross-lane traffic stays
// inside the group and never touches LDS memory:
//   * G == 64 : v_readlane_b32 (value lands in SGPRs and feeds v_fma_f64 as a scalar operand)
//   * G == 4, 8, 16 : DPP quad_perm / row_newbcast (VALU only)
//   * G == 32 : ds_swizzle_b32 in bit-mask mode (LDS crossbar, no LDS storage, no address VGPR)
//   * reductions: DPP (quad_perm / row_half_mirror / row_mirror) up to 16 lanes, then swizzle / bpermute
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

// inf or NaN, from the exponent bits.  (The arithmetic idiom `x - x != 0` is NOT safe here: when x is a fresh product a * b the compiler
// contracts x - x into fma(a, b, -x), the rounding residual of the product, which is non-zero for finite x.)
__device__ __forceinline__ bool nonfinite(double x) { return (__double2hiint(x) & 0x7ff00000) == 0x7ff00000; }

// pairwise (tree) sum of a register array: ceil(log2 N) dependent adds instead of N - 1
template <int LO, int HI, int N>
__device__ __forceinline__ double tree_sum_range(const double (&v)[N]) {
  if constexpr (HI - LO == 1) return v[LO];
  else { constexpr int MID = (LO + HI) / 2; return tree_sum_range<LO, MID>(v) + tree_sum_range<MID, HI>(v); }
}
template <int N>
__device__ __forceinline__ double tree_sum(const double (&v)[N]) { return tree_sum_range<0, N>(v); }

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// broadcast lane K of each group to every lane of that group (K compile-time)
template <int G, int K>
__device__ __forceinline__ double bcast(double v) {
  static_assert(K >= 0 && K < G, "lane out of group");
  int lo = __double2loint(v), hi = __double2hiint(v);
  if constexpr (G == 64) {
    lo = __builtin_amdgcn_readlane(lo, K);
    hi = __builtin_amdgcn_readlane(hi, K);
  } else if constexpr (G == 16) {
    // DPP row_newbcast:K -- lane K of each 16-lane row to the whole row; stays in the VALU
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + K, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + K, 0xf, 0xf, true);
  } else if constexpr (G == 8) {
    // two bank-masked row_newbcasts: lanes 0-7 of a row take lane K, lanes 8-15 take lane 8 + K
    const int l0 = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xf, 0x3, false);
    const int h0 = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xf, 0x3, false);
    lo = __builtin_amdgcn_update_dpp(l0, lo, 0x150 + 8 + K, 0xf, 0xc, false);
    hi = __builtin_amdgcn_update_dpp(h0, hi, 0x150 + 8 + K, 0xf, 0xc, false);
  } else if constexpr (G == 4) {
    lo = __builtin_amdgcn_update_dpp(lo, lo, K * 0x55, 0xf, 0xf, true);      // quad_perm [K,K,K,K]
    hi = __builtin_amdgcn_update_dpp(hi, hi, K * 0x55, 0xf, 0xf, true);
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

// value of group lane (row - D) [D > 0] or (row + |D|) [D < 0]; lanes whose source falls outside the group get 0.
// G <= 16: DPP row_shr / row_shl (zero fill at the 16-lane row edge; for G == 8 the half-row edge is masked explicitly so that
// nothing -- in particular no NaN of a failed neighbour replica -- leaks across groups).  G >= 32: ds_bpermute.
template <int G, int D>
__device__ __forceinline__ double gshift(double v, int row, int lane) {
  constexpr int AD = D > 0 ? D : -D;
  static_assert(AD >= 1 && AD < G, "shift distance");
  const bool inside = (D > 0) ? (row - AD >= 0) : (row + AD < G);
  if constexpr (G <= 16 && AD < 16) {
    constexpr int ctrl = (D > 0 ? 0x110 : 0x100) + AD;              // row_shr:AD / row_shl:AD
    const int l_ = __double2loint(v), h_ = __double2hiint(v);
    const int lo = __builtin_amdgcn_update_dpp(l_, l_, ctrl, 0xf, 0xf, true);       // bound_ctrl: lanes shifted in from outside the row read 0
    const int hi = __builtin_amdgcn_update_dpp(h_, h_, ctrl, 0xf, 0xf, true);
    const double r = __hiloint2double(hi, lo);
    if constexpr (G == 16) return r;                                  // DPP already zero-fills at the row edge
    else return inside ? r : 0.0;
  } else {
    const double r = gshfl<G>(v, row - D, lane);
    return inside ? r : 0.0;
  }
}

// ---- butterfly partners.  Levels 1..8 stay in the VALU (DPP: quad_perm / row_half_mirror / row_mirror, ~1 issue slot
// per dword, no LDS-crossbar trip); level 16 uses ds_swizzle, level 32 ds_bpermute.  Mirrors instead of XORs at levels
// 4 and 8 pair the same sub-groups, and every lane of a group still ends with the bit-identical result (fp add and
// max are commutative, and the tree shape is the same for all lanes).
// `old` = the source itself and bound_ctrl = 1: every lane of these permutations has a valid source, so no zero-initialised
// destination register (an extra v_mov per dword) is needed
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int l = __double2loint(v), h = __double2hiint(v);
  const int lo = __builtin_amdgcn_update_dpp(l, l, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(h, h, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

template <int LEVEL>
__device__ __forceinline__ double partner(double v, int lane) {
  if constexpr (LEVEL == 1) return dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
  else if constexpr (LEVEL == 2) return dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  else if constexpr (LEVEL == 4) return dpp_mov<0x141>(v);  // row_half_mirror: i <-> 7 - i
  else if constexpr (LEVEL == 8) return dpp_mov<0x140>(v);  // row_mirror: i <-> 15 - i
  else if constexpr (LEVEL == 16) {
    constexpr int pat = 0x1f | (16 << 10);                  // ds_swizzle bit-mask mode, xor 16
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), pat);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pat);
    return __hiloint2double(hi, lo);
  } else {
    const int addr = (lane ^ 32) << 2;
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
  }
}

// value of lane (lane ^ MASK): exact XOR partner (hypercube neighbours of the randmod bit-mask states).
// MASK 1, 2, 3: quad_perm; 4 and 8: two DPP mirrors composed (xor 7 . xor 3, xor 15 . xor 7) -- VALU only; 16: ds_swizzle; 32: ds_bpermute
template <int MASK>
__device__ __forceinline__ double xor_partner(double v) {
  static_assert(MASK > 0 && MASK < 64, "xor mask");
  if constexpr (MASK == 1) return dpp_mov<0xB1>(v);
  else if constexpr (MASK == 2) return dpp_mov<0x4E>(v);
  else if constexpr (MASK == 3) return dpp_mov<0x1B>(v);     // quad_perm [3,2,1,0]
  else if constexpr (MASK == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));     // row_half_mirror (i -> 7 - i) then quad reverse (i -> i ^ 3)
  else if constexpr (MASK == 8) return dpp_mov<0x141>(dpp_mov<0x140>(v));    // row_mirror (i -> 15 - i) then row_half_mirror
  else if constexpr (MASK == 32) {
    const int addr = (lane_id() ^ 32) << 2;
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
  } else {
    constexpr int pat = 0x1f | (MASK << 10);
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), pat);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pat);
    return __hiloint2double(hi, lo);
  }
}

// all-reduce over the group: every lane ends with the bit-identical result
template <int G>
__device__ __forceinline__ double gsum(double v, int lane) {
  if constexpr (G >= 2)  v += partner<1>(v, lane);
  if constexpr (G >= 4)  v += partner<2>(v, lane);
  if constexpr (G >= 8)  v += partner<4>(v, lane);
  if constexpr (G >= 16) v += partner<8>(v, lane);
  if constexpr (G >= 32) v += partner<16>(v, lane);
  if constexpr (G >= 64) v += partner<32>(v, lane);
  return v;
}
template <int G>
__device__ __forceinline__ double gmax(double v, int lane) {
  // NaN-propagating max: a NaN anywhere in the group must surface (fmax would swallow it)
  auto mx = [](double a, double b) { return (a > b || a != a) ? a : b; };
  if constexpr (G >= 2)  v = mx(v, partner<1>(v, lane));
  if constexpr (G >= 4)  v = mx(v, partner<2>(v, lane));
  if constexpr (G >= 8)  v = mx(v, partner<4>(v, lane));
  if constexpr (G >= 16) v = mx(v, partner<8>(v, lane));
  if constexpr (G >= 32) v = mx(v, partner<16>(v, lane));
  if constexpr (G >= 64) v = mx(v, partner<32>(v, lane));
  return v;
}

}  // namespace pk
