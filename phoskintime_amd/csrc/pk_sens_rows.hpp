// pk_sens_rows.hpp -- forward parameter sensitivities of the distributive / successive models at the sizes the column-per-lane kernel
// of pk_sens.hpp cannot hold (n_sites = 15 .. 62: 1 + P = 35 .. 129 columns of 17 .. 64 rows): sens_rows_kernel<MODEL, G, KC>.
//
// What it replaces: the 1 + P calls of models.solve_ode per Jacobian that scipy.optimize.curve_fit's '2-point' rule makes under
// paramest/normest.py:167-326 -- at BASELINE config 3's size (distmod n = 30, P = 64) 65 solves per Jacobian.
//
// Same mathematics as pk_sens.hpp (internal differentiation of the LRP12 resolvent step: exact derivative of the discrete solution):
//     M = I - g h A = g h W,   W = gW I - A,  gW = 1 / (g h);      z_1 = W^-1 f(y) / g,     z_{j+1} = gW W^-1 z_j
//     z'_1 = W^-1 (A y' + b' + A' (y + g z_1)) / g,                z'_{j+1} = W^-1 (gW z'_j + A' z_{j+1})
// Mapping, transposed with respect to pk_sens.hpp: a group of G lanes (G = 32 or 64) owns the ROWS of the system, lane r <-> state r, as
// the generic solve_kernel does -- and every lane carries KC COLUMNS of its row: the state itself and KC - 1 tangents.  The factors of W
// (arrow elimination / parallel cyclic reduction, pk_linsolve.hpp) are computed once per step and serve all KC right-hand sides of a
// stage; the cross-lane traffic of the KC solves is independent and interleaves.  The 1 + P columns of a replica are cut into
// ceil(P / (KC - 1)) chunks, each chunk a group of its own that integrates the state again next to its tangents: nothing is exchanged
// between the chunks (no barrier, no LDS), a chunk controls its step size on the maximum norm over ITS columns, and only chunk 0 writes
// flat / n_steps.  The redundancy is 1 / KC of the work; what it buys is ceil(P / (KC - 1)) independent wave-level work items per replica
// -- the fits this kernel serves have tens to hundreds of rows, far too few to fill 256 CUs one wave per replica.
// A'_c v has at most three non-zeros per row for these models (d diag, d c1, d c2 of RowCoef), so the derivative coefficients of a
// lane are three doubles per column; the cross-lane operands of A' v (the P / R entries of v for distmod, the row above for succmod) are
// fetched once per stage for all columns.
#pragma once
#include "pk_sens.hpp"

namespace pk {

// d RowCoef / d theta_p for the row this lane owns (load_row<MODEL> in pk_models.hpp is affine in theta: these are its slopes)
template <int MODEL>
__device__ __forceinline__ void row_coef_deriv(const int p, const int n, const int S, const int row, double& dbias, double& ddg, double& dc1, double& dc2) {
  dbias = 0.0; ddg = 0.0; dc1 = 0.0; dc2 = 0.0;
  if (row >= S || p < 0) return;
  if (row == 0) { dbias = (p == 0) ? 1.0 : 0.0; ddg = (p == 1) ? -1.0 : 0.0; return; }
  if constexpr (MODEL == M_DIST) {
    if (row == 1) { ddg = (p == 3 || (p >= 4 && p < 4 + n)) ? -1.0 : 0.0; dc2 = (p == 2) ? 1.0 : 0.0; }
    else { const int j = row - 2; dc1 = (p == 4 + j) ? 1.0 : 0.0; ddg = (p == 4 + n + j) ? -1.0 : 0.0; }
  } else {
    static_assert(MODEL == M_SUCC, "chain models only");
    if (row == 1) { dc1 = (p == 2) ? 1.0 : 0.0; ddg = ((p == 3) || (n > 0 && p == 4)) ? -1.0 : 0.0; }
    else {
      const int j = row - 2;
      const bool last = (j == n - 1);
      dc1 = (p == 4 + j) ? 1.0 : 0.0;
      ddg = ((p == 4 + n + j) || (!last && p == 4 + j + 1)) ? -1.0 : 0.0;
    }
  }
}

template <int MODEL, int G, int KC>
__global__ __launch_bounds__(256) void sens_rows_kernel(const SensArgs SA) {
  using Tab = ResolventTab<PK_METHOD_LRP12>;
  using Solver = typename SolverFor<MODEL, G, true>::type;
  constexpr int WPB = 256 / G, KT = KC - 1;
  constexpr double GAMMA = Tab::GAM;
  const SolveArgs& A = SA.s;
  const int lane = lane_id();
  const int row = threadIdx.x & (G - 1);
  const int P = A.P, S = A.S, n = A.n_sites, T = A.T, F = A.F;
  const int nch = (P + KT - 1) / KT;
  const long long item = (long long)blockIdx.x * WPB + (threadIdx.x / G);
  const long long rep = item / nch;
  const int ch = (int)(item - rep * nch);
  if (rep >= A.B) return;                                 // whole groups leave together

  const RowCoef c = load_row<MODEL>(A.theta + rep * P, n, S, row);
  RowCoef c0 = c; c0.bias = 0.0;                          // A w without the constant term
  double ddg[KT], dc1[KT], dc2[KT], dbias[KT];
  static_for<KT>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const int p = ch * KT + k;
    row_coef_deriv<MODEL>(p < P ? p : -1, n, S, row, dbias[k], ddg[k], dc1[k], dc2[k]);
  });

  const double y_init = (row < S) ? A.y0[(A.y0_batched ? rep * S : 0) + row] : 0.0;
  double y[KC];
  y[0] = y_init;
  static_for<KT>([&](auto kc) { y[1 + decltype(kc)::value] = 0.0; });     // the initial condition is data: zero tangents

  const int T5 = T > 5 ? T - 5 : 0;
  const bool observed = row < S && row < 2 + n;
  const double yscale = (A.normalize && row < S) ? 1.0 / y_init : 1.0;
  double* const fl = A.flat + rep * F;
  double* const dfl = SA.dflat + rep * (long long)F * P + (long long)ch * KT;
  auto emit = [&](const int kk, const bool nan_fill) {
    if (!observed) return;
    const int fi = (row == 0) ? (kk >= 5 ? kk - 5 : -1) : (row == 1 ? T5 + kk : T5 + T + (row - 2) * T + kk);
    if (fi < 0) return;
    const bool clipped = A.clip && (y[0] < 0.0);
    const double qnan = __builtin_nan("");
    if (ch == 0) fl[fi] = nan_fill ? qnan : (clipped ? 0.0 : y[0] * yscale);
    static_for<KT>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if (ch * KT + k < P) dfl[(long long)fi * P + k] = nan_fill ? qnan : (clipped ? 0.0 : y[1 + k] * yscale);
    });
  };
  auto fail_from = [&](int kk) {
#pragma unroll 1
    for (; kk < T; ++kk) emit(kk, true);
  };
  auto finish = [&](const int status, const int acc, const int rej) {
    if (row != 0) return;
    if (A.status && status) atomicOr(&A.status[rep], status);          // zeroed by the launcher: a failure in ANY chunk flags the replica
    if (ch == 0 && A.n_steps) { A.n_steps[2 * rep] = acc; A.n_steps[2 * rep + 1] = rej; }
  };

  emit(0, false);
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  if (T < 2) { finish(status, 0, 0); return; }

  const double rtol = A.rtol, atol = A.atol;
  Solver ls;
  // cross-lane operands of A'_c v for all columns: distmod (v_P, v_R), succmod (v of the row above, unused)
  auto operands = [&](const double v, double& vA, double& vB) {
    if constexpr (MODEL == M_DIST) { vA = bcast<G, 1>(v); vB = bcast<G, 0>(v); }
    else { vA = gshift<G, 1>(v, row, lane); vB = 0.0; }
  };

  double tc = A.t[0];
  int k = 1;
  double te = A.t[1];
  double h;
  {
    const double f0 = rhs<MODEL, G>(c, y[0], n, S, row, lane);
    const double sc = __builtin_fma(rtol, fabs(y[0]), atol);
    const double d0 = gmax<G>(fabs(y[0]) / sc, lane), d1 = gmax<G>(fabs(f0) / sc, lane);
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
    if (A.h0 > 0.0) h = A.h0;
    if (!(h > 0.0) || h != h) h = 1e-6;
  }
  const double* const kB = Tab::B;                         // indexed by the (uniform) stage counter of the rolled loop
  const double* const kE = Tab::E;
  bool after_reject = false;
  while (true) {
    if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; fail_from(k); break; }
    const bool last = (tc + 1.0001 * h >= te);
    const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
    if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; fail_from(k); break; }
    const double gW = (1.0 / hs) * (1.0 / GAMMA);
    ls.factor(c, gW, S, row, lane);

    double z[KC], yn[KC], e[KC];
    {                                                      // stage 1
      z[0] = ls.solve(rhs<MODEL, G>(c, y[0], n, S, row, lane), S, row, lane) * (1.0 / GAMMA);
      const double v = __builtin_fma(GAMMA, z[0], y[0]);
      double vA, vB;
      operands(v, vA, vB);
      static_for<KT>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        double r = rhs<MODEL, G>(c0, y[1 + kk], n, S, row, lane) + dbias[kk];
        r = __builtin_fma(ddg[kk], v, r); r = __builtin_fma(dc1[kk], vA, r); r = __builtin_fma(dc2[kk], vB, r);
        z[1 + kk] = ls.solve(r, S, row, lane) * (1.0 / GAMMA);
      });
      static_for<KC>([&](auto kc) { constexpr int kk = decltype(kc)::value; yn[kk] = __builtin_fma(Tab::B[0], z[kk], y[kk]); e[kk] = 0.0; });
    }
#pragma unroll 1
    for (int j = 1; j < Tab::NS; ++j) {                    // stages 2 .. 12: one copy of the KC solves in the code
      z[0] = ls.solve(z[0], S, row, lane) * gW;
      double vA, vB;
      operands(z[0], vA, vB);
      static_for<KT>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        double r = __builtin_fma(ddg[kk], z[0], gW * z[1 + kk]);
        r = __builtin_fma(dc1[kk], vA, r); r = __builtin_fma(dc2[kk], vB, r);
        z[1 + kk] = ls.solve(r, S, row, lane);
      });
      const double bj = kB[j], ej = kE[j];
      static_for<KC>([&](auto kc) { constexpr int kk = decltype(kc)::value; yn[kk] = __builtin_fma(bj, z[kk], yn[kk]); e[kk] = __builtin_fma(ej, z[kk], e[kk]); });
    }

    double em = 0.0;
    static_for<KC>([&](auto kc) {
      constexpr int kk = decltype(kc)::value;
      const double q = fabs(e[kk]) * approx_rcp(__builtin_fma(rtol, fmax(fabs(y[kk]), fabs(yn[kk])), atol));
      em = (q > em || q != q) ? q : em;
    });
    const double err = gmax<G>(em, lane);
    if (err != err || err > 1e300) {
      ++nrej; after_reject = true; h = 0.1 * hs;
      const double bad = gmax<G>((nonfinite(y[0]) || nonfinite(c.dg) || nonfinite(c.c1) || nonfinite(c.c2) || nonfinite(c.bias)) ? 1.0 : 0.0, lane);
      if (bad != 0.0) { status |= PK_ST_NONFINITE; fail_from(k); break; }
      continue;
    }
    double fac = root_q(err, Tab::Q) * (1.0 / 0.9);
    fac = fmax(1.0 / 6.0, fmin(5.0, fac));
    double hnew = hs * fast_rcp(fac);
    if (err <= 1.0) {
      ++nacc;
      static_for<KC>([&](auto kc) { constexpr int kk = decltype(kc)::value; y[kk] = yn[kk]; });
      tc += hs;
      if (after_reject) hnew = fmin(hnew, hs);
      after_reject = false;
      if (last) {
        tc = te;
        emit(k, false);
        ++k;
        h = (hs < h) ? fmax(hnew, h) : hnew;
        if (k >= T) break;
        te = A.t[k];
      } else {
        h = hnew;
      }
    } else {
      ++nrej; after_reject = true;
      h = hnew;
    }
  }
  finish(status, nacc, nrej);
}

constexpr int kSensRowsKC = 8;                             // 1 state + 7 tangent columns per lane

template <int MODEL, int G>
static hipError_t launch_sens_rows_one(const SensArgs& a, hipStream_t st) {
  constexpr int KT = kSensRowsKC - 1;
  const long long nch = (a.s.P + KT - 1) / KT;
  const long long items = a.s.B * nch, wpb = 256 / G;
  const long long nblk = (items + wpb - 1) / wpb;
  if (nblk > 0x7fffffffLL) return hipErrorInvalidValue;
  if (a.s.status) {                                        // chunks OR their flags into it
    hipError_t e = hipMemsetAsync(a.s.status, 0, (size_t)a.s.B * sizeof(int32_t), st);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((sens_rows_kernel<MODEL, G, kSensRowsKC>), dim3((unsigned)nblk), dim3(256), 0, st, a);
  return hipGetLastError();
}

template <int MODEL>
static hipError_t launch_sens_rows_model(const SensArgs& a, hipStream_t st) {
  if (a.s.S <= 16) return launch_sens_rows_one<MODEL, 16>(a, st);      // only reached on request (PK_SENS_ROWS=1: A/B against pk_sens.hpp)
  if (a.s.S <= 32) return launch_sens_rows_one<MODEL, 32>(a, st);
  return launch_sens_rows_one<MODEL, 64>(a, st);
}

}  // namespace pk
