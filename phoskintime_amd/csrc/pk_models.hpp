// pk_models.hpp -- the three per-protein kinetic models of the reference, lane-per-state.
//
//   DIST  models/distmod.py:7-65    dR = A - B R ; dP = C R - (D + sum S) P + sum P_i ; dP_i = S_i P - (1 + D_i) P_i
//   SUCC  models/succmod.py:9-90    chain P -> P_1 -> ... -> P_n (tridiagonal), n == 1 special case (:59-63)
//   RAND  models/randmod.py:122-247 2^n - 1 bit-mask states, incl. the quirk at randmod.py:201: every inflow into
//                                   state `tgt` (from P or from tgt ^ bit) carries the rate S[lowest set bit of tgt]
//
// All three are affine in y (dy/dt = J y + b, J constant), so each lane keeps the few coefficients of ITS row in
// registers (loaded once from the [B, P] parameter matrix) and both the right-hand side and the row of
// W = g I - J are produced from them with no further memory traffic.
#pragma once
#include "pk_wave.hpp"

namespace pk {

enum { M_DIST = 0, M_SUCC = 1, M_RAND = 2 };

__host__ __device__ inline int n_states(int model, int n) { return 2 + (model == M_RAND ? (1 << n) - 1 : n); }
__host__ __device__ inline int n_params(int model, int n) { return 4 + n + (model == M_RAND ? (1 << n) - 1 : n); }

// Coefficients of one row of the affine system, as seen by the lane that owns the row.
struct RowCoef {
  double bias;  // constant term b_row (A for the mRNA row, else 0)
  double dg;    // J[row][row]
  double c1;    // DIST: J[row][1] (rows >= 2: S_i) | SUCC: J[row][row-1] | RAND: inflow rate S[lsb(mask)]
  double c2;    // DIST: row 1: J[1][0] = C          | SUCC: J[row][row+1] | RAND: row 1: C
};

template <int MODEL>
__device__ __forceinline__ RowCoef load_row(const double* __restrict__ th, const int n, const int S, const int row) {
  RowCoef c{0.0, 0.0, 0.0, 0.0};
  if (row >= S) return c;               // padding rows: dy = 0, W row = identity
  const double* Sr = th + 4;            // S_1..S_n
  const double* Dr = th + 4 + n;        // D_1..D_m
  if (row == 0) { c.bias = th[0]; c.dg = -th[1]; return c; }
  if constexpr (MODEL == M_DIST) {
    if (row == 1) {
      double sumS = 0.0;
      for (int i = 0; i < n; ++i) sumS += Sr[i];        // same order as distmod.py:41-43
      c.dg = -(th[3] + sumS);
      c.c2 = th[2];
    } else {
      c.c1 = Sr[row - 2];
      c.dg = -(1.0 + Dr[row - 2]);
    }
  } else if constexpr (MODEL == M_SUCC) {
    if (row == 1) {
      c.c1 = th[2];                                      // C * R
      c.dg = -th[3];
      if (n > 0) { c.dg -= Sr[0]; c.c2 = 1.0; }          // succmod.py:42-48
    } else {
      const int i = row - 2;
      c.c1 = Sr[i];
      const bool last = (i == n - 1);
      c.dg = last ? -(1.0 + Dr[i]) : -(1.0 + Sr[i + 1] + Dr[i]);
      c.c2 = last ? 0.0 : 1.0;
    }
  } else {
    const int m = row - 1;                               // bit mask of this state; 0 = unphosphorylated P
    if (m == 0) {
      double sumS = 0.0;
      for (int i = 0; i < n; ++i) sumS += Sr[i];
      c.dg = -(th[3] + sumS);
      c.c2 = th[2];
    } else {
      const int lsb = __builtin_ctz(m);
      c.c1 = Sr[lsb];
      double out = 0.0;
      for (int j = 0; j < n; ++j) {
        if (m & (1 << j)) out += 1.0;                                    // dephosphorylation of bit j, unit rate
        else              out += Sr[j < lsb ? j : lsb];                  // forward, rate S[lsb(m | 1<<j)]
      }
      c.dg = -(out + Dr[m - 1]);
    }
  }
  return c;
}

// dy_row/dt for the lane's row.  y is the lane's own state value; every lane of the group must call.
template <int MODEL, int G>
__device__ __forceinline__ double rhs(const RowCoef& c, const double y, const int n, const int S,
                                      const int row, const int lane) {
  if constexpr (MODEL == M_DIST) {
    const double R = bcast<G, 0>(y);
    const double P = bcast<G, 1>(y);
    const double sumsites = gsum<G>((row >= 2 && row < S) ? y : 0.0, lane);
    double f = __builtin_fma(c.dg, y, c.bias);
    f = __builtin_fma(c.c1, P, f);
    f = __builtin_fma(c.c2, R, f);
    return (row == 1) ? f + sumsites : f;
  } else if constexpr (MODEL == M_SUCC) {
    const double lo = gshift<G, 1>(y, row, lane);       // y[row - 1], 0 outside the group
    const double hi = gshift<G, -1>(y, row, lane);      // y[row + 1]
    double f = __builtin_fma(c.dg, y, c.bias);
    f = __builtin_fma(c.c1, lo, f);
    f = __builtin_fma(c.c2, (row + 1 < S) ? hi : 0.0, f);
    return f;
  } else {
    const int m = row - 1;
    const double R = bcast<G, 0>(y);
    double f = __builtin_fma(c.dg, y, c.bias);
    f = __builtin_fma(c.c2, R, f);
    for (int j = 0; j < n; ++j) {                       // n is uniform across the launch
      const int bit = 1 << j;
      const double v = gshfl<G>(y, (m ^ bit) + 1, lane);
      const double coef = (row >= 1 && row < S) ? ((m & bit) ? c.c1 : 1.0) : 0.0;
      f = __builtin_fma(coef, v, f);
    }
    return f;
  }
}

// J[row][j] for compile-time column j (row runtime).  Padding rows/columns give 0.
template <int MODEL, int J>
__device__ __forceinline__ double jac_entry(const RowCoef& c, const int S, const int row) {
  if (row >= S || J >= S) return 0.0;
  if (J == row) return c.dg;
  if constexpr (MODEL == M_DIST) {
    if constexpr (J == 0) return c.c2;                         // only row 1 has c2 != 0
    else if constexpr (J == 1) return c.c1;                    // rows >= 2: S_i
    else return (row == 1) ? 1.0 : 0.0;
  } else if constexpr (MODEL == M_SUCC) {
    if (J == row - 1) return c.c1;
    if (J == row + 1) return c.c2;
    return 0.0;
  } else {
    if constexpr (J == 0) return c.c2;                         // only row 1 (P) has c2 = C
    else {
      if (row == 0) return 0.0;
      const int m = row - 1, d = m ^ (J - 1);
      const bool one = (d & (d - 1)) == 0;                     // d != 0 here because J != row
      return one ? ((m & d) ? c.c1 : 1.0) : 0.0;
    }
  }
}

// Row of W = g I - J (identity on padding rows) into the lane's register array.
template <int MODEL, int G>
__device__ __forceinline__ void fill_w_row(double (&a)[G], const RowCoef& c, const double g, const int S, const int row) {
  static_for<G>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    double v = -jac_entry<MODEL, j>(c, S, row);
    if (row == j) v = (row < S) ? g - c.dg : 1.0;
    a[j] = v;
  });
}

}  // namespace pk
