// pk_linsolve.hpp -- solvers for the implicit stage systems  (g I - J) x = r,  one replica per lane group.
//
//   DenseInv<G>         any model: W row-per-lane in VGPRs, Gauss-Jordan inverse by cross-lane broadcast (pk_wave.hpp)
//   Arrow<G>  (DIST)    J is an arrow matrix: eliminate the site rows, one group reduction per solve
//   Tridiag<G> (SUCC)   J is tridiagonal: parallel cyclic reduction, log2(G) neighbour exchanges per solve
#pragma once
#include "pk_models.hpp"

namespace pk {

// 1/x to full double precision from v_rcp_f64 + two Newton steps (x normal, non-zero)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}

// 1/x from v_rcp_f64 alone (relative error ~1e-8): for quantities that only steer the step-size controller (error ratios), where one
// instruction instead of seven is worth more than the last eight digits
__device__ __forceinline__ double approx_rcp(double x) { return __builtin_amdgcn_rcp(x); }

// Dense, explicit inverse: in-register Gauss-Jordan (no pivoting, same M-matrix argument), then every solve is a
// mat-vec whose G broadcasts are independent of each other.  The first version of this solver (LU + two triangular solves per
// right-hand side) was bound by the 2G-long dependent chain of cross-lane broadcasts: 363 ms vs 29 ms on config 3.
// The price is ~2x the factorisation flops; with 6-8 solves per factorisation (resolvent-form steps) the inverse wins clearly.
template <int MODEL, int G>
struct DenseInvSolver {
  double a[G];           // row `row` of W, then of W^{-1}
  __device__ __forceinline__ void factor(const RowCoef& c, const double g, const int S, const int row, const int lane) {
    fill_w_row<MODEL, G>(a, c, g, S, row);
    static_for<G>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const double rp = fast_rcp(bcast<G, k>(a[k]));
      // row k: a_kj <- a_kj * rp ; other rows: a_ij <- a_ij - (a_ik rp) a_kj ; both as a_ij - m * a_kj(old)
      const double m = (row == k) ? 1.0 - rp : a[k] * rp;
      static_for<G>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j != k) a[j] = __builtin_fma(-m, bcast<G, k>(a[j]), a[j]);
      });
      a[k] = (row == k) ? rp : -m;
    });
  }
  __device__ __forceinline__ double solve(const double r, const int S, const int row, const int lane) const {
    double x0 = 0.0, x1 = 0.0;           // two accumulators: halves the fma dependency chain
    static_for<G / 2>([&](auto jc) {
      constexpr int j = 2 * decltype(jc)::value;
      x0 = __builtin_fma(a[j], bcast<G, j>(r), x0);
      x1 = __builtin_fma(a[j + 1], bcast<G, j + 1>(r), x1);
    });
    return x0 + x1;
  }
};

// DIST: rows >= 2 couple only to row 1 (column 1 and the diagonal), row 1 couples to row 0 and all sites.
template <int G>
struct ArrowSolver {
  double winv;   // 1 / (g - J[row][row])
  double cw;     // rows >= 2: S_i * winv
  double sinv;   // 1 / Schur complement of row 1 (uniform)
  double C;      // J[1][0] (uniform)
  __device__ __forceinline__ void factor(const RowCoef& c, const double g, const int S, const int row, const int lane) {
    const double w = (row < S) ? g - c.dg : 1.0;
    winv = fast_rcp(w);
    const bool site = (row >= 2 && row < S);
    cw = site ? c.c1 * winv : 0.0;
    const double sum = gsum<G>(cw, lane);               // sum_i S_i / w_i  (J[1][i] = 1)
    const double w1 = bcast<G, 1>(w);
    sinv = fast_rcp(w1 - sum);
    C = bcast<G, 1>(c.c2);
  }
  __device__ __forceinline__ double solve(double r, const int S, const int row, const int lane) const {
    const double t = r * winv;
    const double x0 = bcast<G, 0>(t);
    const double r1 = bcast<G, 1>(r);
    const bool site = (row >= 2 && row < S);
    const double sum = gsum<G>(site ? t : 0.0, lane);
    const double x1 = (__builtin_fma(C, x0, r1) + sum) * sinv;
    const double xs = __builtin_fma(cw, x1, t);
    return (row == 1) ? x1 : xs;                        // row 0: cw == 0 -> t
  }
};

// SUCC: parallel cyclic reduction.  Level d (1, 2, 4, ...): row i eliminates its couplings to i-d and i+d.
// Neighbour values come through gshift (DPP row shifts for G <= 16: no LDS-crossbar trip), which returns 0 outside the group.
template <int G>
struct TridiagSolver {
  static constexpr int LV = (G == 8) ? 3 : (G == 16) ? 4 : (G == 32) ? 5 : 6;
  double kl[LV], ku[LV];   // multipliers per level
  double binv;             // 1 / final diagonal
  __device__ __forceinline__ void factor(const RowCoef& c, const double g, const int S, const int row, const int lane) {
    // row i:  lo * x[i-1] + b * x[i] + up * x[i+1] = r[i]
    double lo = (row < S && row >= 1) ? -c.c1 : 0.0;
    double up = (row + 1 < S) ? -c.c2 : 0.0;
    double b = (row < S) ? g - c.dg : 1.0;
    static_for<LV>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      constexpr int d = 1 << l;
      const bool hl = (row - d >= 0), hu = (row + d < G);
      const double b_m = gshift<G, d>(b, row, lane), b_p = gshift<G, -d>(b, row, lane);
      const double lo_m = gshift<G, d>(lo, row, lane), up_m = gshift<G, d>(up, row, lane);
      const double lo_p = gshift<G, -d>(lo, row, lane), up_p = gshift<G, -d>(up, row, lane);
      const double k1 = hl ? lo * fast_rcp(b_m) : 0.0;
      const double k2 = hu ? up * fast_rcp(b_p) : 0.0;
      kl[l] = k1; ku[l] = k2;
      b = b - k1 * up_m - k2 * lo_p;                     // shifted-in values are 0 outside the group
      lo = -k1 * lo_m;
      up = -k2 * up_p;
    });
    binv = fast_rcp(b);
  }
  __device__ __forceinline__ double solve(double r, const int S, const int row, const int lane) const {
    static_for<LV>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      constexpr int d = 1 << l;
      const double r_m = gshift<G, d>(r, row, lane), r_p = gshift<G, -d>(r, row, lane);
      r = __builtin_fma(-kl[l], r_m, r);
      r = __builtin_fma(-ku[l], r_p, r);
    });
    return r * binv;
  }
};

template <int MODEL, int G, bool STRUCTURED> struct SolverFor { using type = DenseInvSolver<MODEL, G>; };
template <int G> struct SolverFor<M_DIST, G, true> { using type = ArrowSolver<G>; };
template <int G> struct SolverFor<M_SUCC, G, true> { using type = TridiagSolver<G>; };

}  // namespace pk
