// pk_rand_fast.hpp -- throughput kernel for the random (2^n - 1 phospho-state) model, models/randmod.py:122-247.
//
// The generic dense kernel maps state r to lane r, so S = 2^n + 1 = 17 / 33 states need 32 / 64 lanes, pads the matrix to
// 32x32 / 64x64 and inverts the padding as well.  Here the mRNA row R -- which no other row feeds back into -- is carried
// as a per-group uniform scalar ("shadow"), leaving exactly 2^n coupled rows, one per lane, indexed by the bit mask itself
// (mask 0 = unphosphorylated protein P): G = 2^n lanes per replica, 64 / G replicas per wave, no padding, and for n <= 4
// every broadcast of the Gauss-Jordan inversion is a DPP move (no LDS crossbar).
//
// Row m of J (reference semantics incl. the lowest-set-bit rate rule, randmod.py:201):
//   neighbour m ^ (1 << j):  bit j set in m  -> inflow S[lsb(m)]     (from the state lacking bit j, or from P if that is 0)
//                            bit j clear     -> inflow 1              (de-phosphorylation of the state that has bit j)
//   diagonal:  -( sum_{j clear} S[min(j, lsb(m))] + popcount(m) + Ddeg[m-1] )      (m > 0)
//              -( D + sum_j S[j] )                                                 (m = 0), plus C * R from the mRNA row.
#pragma once
#include "pk_solve_kernel.hpp"

namespace pk {

template <int NB, int METHOD>
__global__ __launch_bounds__(256) void rand_fast_kernel(const SolveArgs A) {
  using Tab = ResolventTab<METHOD>;
  constexpr int NM = 1 << NB;                        // coupled rows (bit masks 0 .. 2^n - 1)
  constexpr int G = NM < 4 ? 4 : NM;                 // lanes per replica
  constexpr int RPB = 256 / G;
  const int lane = lane_id();
  const int m = threadIdx.x & (G - 1);               // this lane's bit mask
  const long long rep = (long long)blockIdx.x * RPB + (threadIdx.x / G);
  if (rep >= A.B) return;
  const int n = NB, S = A.S, T = A.T;
  const bool live = m < NM;                          // only n = 1 leaves padding lanes (G = 4 > 2)
  const double* __restrict__ th = A.theta + rep * A.P;
  const double* Sr = th + 4;
  const double* Dd = th + 4 + n;

  // ---- coefficients
  const double cA = th[0], cB = th[1], cC = th[2];
  double dgn, cin = 0.0;                             // -J[m][m], inflow rate of this row
  if (!live) dgn = 0.0;
  else if (m == 0) {
    double sumS = 0.0;
    for (int j = 0; j < n; ++j) sumS += Sr[j];
    dgn = th[3] + sumS;
  } else {
    const int lsb = __builtin_ctz(m);
    cin = Sr[lsb];
    double out = 0.0;
    for (int j = 0; j < n; ++j) out += (m & (1 << j)) ? 1.0 : Sr[j < lsb ? j : lsb];
    dgn = out + Dd[m - 1];
  }
  // neighbour coefficients, one per bit
  double nb[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) nb[j] = live ? ((m & (1 << j)) ? cin : 1.0) : 0.0;

  // ---- state: y (this lane's mask state), yR (shadow)
  const double* y0p = A.y0 + (A.y0_batched ? rep * S : 0);
  double y = live ? y0p[1 + m] : 0.0;
  double yR = y0p[0];

  // ---- output / fused metric.  Observables (sensitivity/analysis.py:90-176): R, P, and the first n phospho columns,
  // i.e. masks 1..n.
  const bool obs = live && m <= n;
  const int T5 = T > 5 ? T - 5 : 0;
  double m1 = 0.0, m2 = 0.0, mdyn = 0.0, shift = 0.0, prev = 0.0, prevR = 0.0;
  auto emit = [&](const int k, const double v, const double vRaw, const bool nan_fill) {
    auto val = [&](double x, int state) {
      if (nan_fill) return __builtin_nan("");
      double r = A.clip ? ((x < 0.0) ? 0.0 : x) : x;
      if (A.normalize) r *= 1.0 / y0p[state];
      return r;
    };
    const double vs = live ? val(v, 1 + m) : 0.0;
    const double vR = val(vRaw, 0);
    if (A.sol) {
      double* solp = A.sol + (rep * T + k) * S;
      if (live) solp[1 + m] = vs;
      if (m == 0) solp[0] = vR;
    }
    if (A.flat) {
      double* fl = A.flat + rep * A.F;
      if (m == 0) { if (k >= 5) fl[k - 5] = vR; fl[T5 + k] = vs; }
      else if (obs) fl[T5 + T + (m - 1) * T + k] = vs;
    }
    if (A.metric) {
      const double x = obs ? vs : 0.0;
      const double xr = (m == 0) ? vR : 0.0;
      if (k == 0) { shift = gsum<G>(x + xr, lane) / (2 + n); prev = x; prevR = xr; }
      m1 += x + xr;
      const double a = obs ? x - shift : 0.0, b = (m == 0) ? xr - shift : 0.0;
      m2 = __builtin_fma(a, a, m2); m2 = __builtin_fma(b, b, m2);
      const double d = x - prev, dr = xr - prevR;
      mdyn = __builtin_fma(d, d, mdyn); mdyn = __builtin_fma(dr, dr, mdyn);
      prev = x; prevR = xr;
    }
  };
  auto finish = [&](const int status, const int acc, const int rej) {
    if (A.metric) {
      const double L = 2.0 * T + (double)T * n;
      const double tot = gsum<G>(m1, lane);
      double mm;
      switch (A.metric_id) {
        case PK_METRIC_TOTAL_SIGNAL: mm = tot; break;
        case PK_METRIC_MEAN_ACTIVITY: mm = tot / L; break;
        case PK_METRIC_VARIANCE: { const double q = gsum<G>(m2, lane); const double ms = tot / L - shift; mm = q / L - ms * ms; } break;
        case PK_METRIC_DYNAMICS: mm = gsum<G>(mdyn, lane); break;
        default: { const double q = gsum<G>(m2, lane); mm = sqrt(fmax(q + 2.0 * shift * tot - L * shift * shift, 0.0)); } break;
      }
      if (m == 0) A.metric[rep] = mm;
    }
    if (m == 0) {
      if (A.status) A.status[rep] = status;
      if (A.n_steps) { A.n_steps[2 * rep] = acc; A.n_steps[2 * rep + 1] = rej; }
    }
  };
  auto fail_from = [&](int k) { for (; k < T; ++k) emit(k, y, yR, true); };

  emit(0, y, yR, false);
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  if (T < 2) { finish(status, 0, 0); return; }

  const double rtol = A.rtol, atol = A.atol;
  auto ratio = [&](double e, double ya, double yb) { return fabs(e) * approx_rcp(__builtin_fma(rtol, fmax(fabs(ya), fabs(yb)), atol)); };
  auto mxn = [](double p, double r) { return (p > r || p != p) ? p : r; };
  // f(Y) for this lane's row and for the shadow row
  auto rhs_row = [&](const double Y, const double YR) {
    double f = -dgn * Y;
    static_for<NB>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      f = __builtin_fma(nb[j], xor_partner<(1 << j)>(Y), f);
    });
    return (m == 0) ? __builtin_fma(cC, YR, f) : f;
  };

  double tc = A.t[0];
  int k = 1;
  double te = A.t[1];
  double h;
  {
    const double f0 = rhs_row(y, yR), fR = __builtin_fma(-cB, yR, cA);
    const double d0 = gmax<G>(mxn(ratio(y, y, y), ratio(yR, yR, yR)), lane);
    const double d1 = gmax<G>(mxn(ratio(f0, y, y), ratio(fR, yR, yR)), lane);
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
    if (A.h0 > 0.0) h = A.h0;
    if (!(h > 0.0) || h != h) h = 1e-6;
  }

  // M = I - q J on the 2^n coupled rows, inverted in registers (Gauss-Jordan, no pivoting: M is an M-matrix)
  double a[G];
  double winvR, qC;
  auto factor = [&](const double q) {
    winvR = fast_rcp(__builtin_fma(q, cB, 1.0));
    qC = q * cC;
    static_for<G>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      const int d = m ^ c;
      double v = 0.0;
      if (live && c < NM && d != 0 && (d & (d - 1)) == 0) v = -q * ((m & d) ? cin : 1.0);
      if (c == m) v = live ? __builtin_fma(q, dgn, 1.0) : 1.0;
      a[c] = v;
    });
    static_for<G>([&](auto kc) {
      constexpr int kk = decltype(kc)::value;
      const double rp = fast_rcp(bcast<G, kk>(a[kk]));
      const double mlt = (m == kk) ? 1.0 - rp : a[kk] * rp;
      static_for<G>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j != kk) a[j] = __builtin_fma(-mlt, bcast<G, kk>(a[j]), a[j]);
      });
      a[kk] = (m == kk) ? rp : -mlt;
    });
  };
  // (z, zR) = M^{-1} (r, rR)
  auto solve = [&](const double r, const double rR, double& zR) {
    zR = rR * winvR;
    const double rr = (m == 0) ? __builtin_fma(qC, zR, r) : r;       // move the -q C z_R coupling of row P to the right
    double x0 = 0.0, x1 = 0.0;
    static_for<G / 2>([&](auto jc) {
      constexpr int j = 2 * decltype(jc)::value;
      x0 = __builtin_fma(a[j], bcast<G, j>(rr), x0);
      x1 = __builtin_fma(a[j + 1], bcast<G, j + 1>(rr), x1);
    });
    return x0 + x1;
  };

  bool after_reject = false;
  while (true) {
    if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; fail_from(k); break; }
    const bool last = (tc + 1.0001 * h >= te);
    const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
    if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; fail_from(k); break; }
    factor(Tab::GAM * hs);

    double zR;
    double z = solve(hs * rhs_row(y, yR), hs * __builtin_fma(-cB, yR, cA), zR);
    double yn = __builtin_fma(Tab::B[0], z, y), ynR = __builtin_fma(Tab::B[0], zR, yR);
    double e = 0.0, eR = 0.0;
    static_for<Tab::NS - 1>([&](auto kc) {
      constexpr int kk = 1 + decltype(kc)::value;
      double zRn;
      z = solve(z, zR, zRn);
      zR = zRn;
      yn = __builtin_fma(Tab::B[kk], z, yn); ynR = __builtin_fma(Tab::B[kk], zR, ynR);
      e = __builtin_fma(Tab::E[kk], z, e); eR = __builtin_fma(Tab::E[kk], zR, eR);
    });

    const double err = gmax<G>(mxn(ratio(e, y, yn), ratio(eR, yR, ynR)), lane);
    if (err != err || err > 1e300) {
      ++nrej; after_reject = true; h = 0.1 * hs;
      const double bad = gmax<G>(((nonfinite(y)) || (nonfinite(yR)) || (nonfinite(dgn)) || (nonfinite(cin)) ||
                                  (nonfinite(cA)) || (nonfinite(cB)) || (nonfinite(cC))) ? 1.0 : 0.0, lane);
      if (bad != 0.0) { status |= PK_ST_NONFINITE; fail_from(k); break; }
      continue;
    }
    double fac = root_q(err, Tab::Q) * (1.0 / 0.9);
    fac = fmax(1.0 / 6.0, fmin(5.0, fac));
    double hnew = hs * fast_rcp(fac);
    if (err <= 1.0) {
      ++nacc;
      y = yn; yR = ynR; tc += hs;
      if (after_reject) hnew = fmin(hnew, hs);
      after_reject = false;
      if (last) {
        tc = te;
        emit(k, y, yR, false);
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

}  // namespace pk
