// pk_rand_fastr.hpp -- random model, SEVERAL bit-mask rows per lane: rand_fastr_kernel<NB, RPL, METHOD>.
//
// The 2^NB coupled rows are spread over G = 2^NB / RPL lanes, lane l owning the masks l + G * s, s = 0 .. RPL - 1.  Versus one row per
// lane (pk_rand_fast.hpp) every broadcast of the Gauss-Jordan inversion and of the mat-vec solves feeds RPL rows instead of one, and
// more replicas share a wave:
//   * n = 5: G = 16, RPL = 2 -- broadcasts are DPP row_newbcast moves instead of ds_swizzle trips through the LDS crossbar (1.8x);
//   * n = 4: G = 4, RPL = 4 and n = 3: G = 4, RPL = 2 -- quad_perm broadcasts, 16 replicas per wave, half the instructions per replica.
// The neighbour of a mask across bit j is an XOR-partner lane if 2^j < G, otherwise another slot of the same lane.
// Method, coefficients, controller and outputs are those of pk_rand_fast.hpp (reference: models/randmod.py:122-247).
#pragma once
#include "pk_solve_kernel.hpp"

namespace pk {

template <int NB, int RPL, int METHOD>
__global__ __launch_bounds__(256, 2) void rand_fastr_kernel(const SolveArgs A) {
  using Tab = ResolventTab<METHOD>;
  constexpr int NM = 1 << NB, G = NM / RPL, RPB = 256 / G;
  static_assert(G == 4 || G == 16, "group width: quad_perm or row_newbcast broadcasts");
  constexpr int LG = (G == 4) ? 2 : 4;                       // bits that address the lane inside the group
  const int lane = lane_id();
  const int l = threadIdx.x & (G - 1);
  const long long rep = (long long)blockIdx.x * RPB + (threadIdx.x / G);
  if (rep >= A.B) return;
  const int n = NB, S = A.S, T = A.T;
  const double* __restrict__ th = A.theta + rep * A.P;
  const double* Sr = th + 4;
  const double* Dd = th + 4 + n;
  const double cA = th[0], cB = th[1], cC = th[2];

  // Values touched once per step (row coefficients) or once per output (metric bookkeeping) live in LDS, slot-major (slot * 256 + thread:
  // conflict-free, thread-private, no barrier): the registers belong to the RPL x 2^NB matrix.
  extern __shared__ __align__(16) double park_lds[];
  double* const park = park_lds + threadIdx.x;
  constexpr int K_DG = 0, K_CI = RPL, K_PV = 2 * RPL, K_PR = 3 * RPL, K_M1 = 3 * RPL + 1, K_M2 = 3 * RPL + 2, K_MD = 3 * RPL + 3, K_SH = 3 * RPL + 4;
  auto ld = [&](int k) { return park[k * 256]; };
  auto st = [&](int k, double v) { park[k * 256] = v; };

  // ---- coefficients of the lane's rows: mask of slot s is l + G * s
  static_for<RPL>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    const int m = l + G * s;
    if (m == 0) {
      double sumS = 0.0;
      for (int j = 0; j < n; ++j) sumS += Sr[j];
      st(K_DG + s, th[3] + sumS); st(K_CI + s, 0.0);
    } else {
      const int lsb = __builtin_ctz(m);
      st(K_CI + s, Sr[lsb]);
      double out = 0.0;
      for (int j = 0; j < n; ++j) out += (m & (1 << j)) ? 1.0 : Sr[j < lsb ? j : lsb];
      st(K_DG + s, out + Dd[m - 1]);
    }
  });

  const double* y0p = A.y0 + (A.y0_batched ? rep * S : 0);
  double y[RPL];
  static_for<RPL>([&](auto sc) { constexpr int s = decltype(sc)::value; y[s] = y0p[1 + l + G * s]; });
  double yR = y0p[0];

  // ---- outputs / fused metric.  Observables (sensitivity/analysis.py:90-176): R, P (mask 0) and the masks 1..n
  const int T5 = T > 5 ? T - 5 : 0;
  static_for<RPL + 5>([&](auto kc) { st(K_PV + decltype(kc)::value, 0.0); });
  auto emit = [&](const int k, const double (&v)[RPL], const double vRaw, const bool nan_fill) {
    auto val = [&](double x, int state) {
      if (nan_fill) return __builtin_nan("");
      double r = A.clip ? ((x < 0.0) ? 0.0 : x) : x;
      if (A.normalize) r *= 1.0 / y0p[state];
      return r;
    };
    const double vR = val(vRaw, 0);
    double* solp = A.sol ? A.sol + (rep * T + k) * S : nullptr;
    double* fl = A.flat ? A.flat + rep * A.F : nullptr;
    if (l == 0) {
      if (solp) solp[0] = vR;
      if (fl && k >= 5) fl[k - 5] = vR;
    }
    double x[RPL];                                           // observable values of this lane (0 for non-observables)
    static_for<RPL>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      const int m = l + G * s;
      const double vv = val(v[s], 1 + m);
      if (solp) solp[1 + m] = vv;
      if (fl) {
        if (m == 0) fl[T5 + k] = vv;
        else if (m <= n) fl[T5 + T + (m - 1) * T + k] = vv;
      }
      x[s] = (m <= n) ? vv : 0.0;
    });
    if (A.metric) {
      const double xr = (l == 0) ? vR : 0.0;
      double loc = xr;
      static_for<RPL>([&](auto sc) { loc += x[decltype(sc)::value]; });
      st(K_M1, ld(K_M1) + loc);
      if (!(A.metric_id == PK_METRIC_TOTAL_SIGNAL || A.metric_id == PK_METRIC_MEAN_ACTIVITY)) {       // uniform across the launch
        double m2 = ld(K_M2), mdyn = ld(K_MD), shift = ld(K_SH), prevR = ld(K_PR);
        if (k == 0) {
          shift = gsum<G>(loc, lane) / (2 + n);
          st(K_SH, shift);
          prevR = xr;
          static_for<RPL>([&](auto sc) { constexpr int s = decltype(sc)::value; st(K_PV + s, x[s]); });
        }
        const double b = (l == 0) ? xr - shift : 0.0;
        m2 = __builtin_fma(b, b, m2);
        const double dr = xr - prevR;
        mdyn = __builtin_fma(dr, dr, mdyn);
        static_for<RPL>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          const int m = l + G * s;
          const double a = (m <= n) ? x[s] - shift : 0.0;
          m2 = __builtin_fma(a, a, m2);
          const double d = x[s] - ld(K_PV + s);
          mdyn = __builtin_fma(d, d, mdyn);
          st(K_PV + s, x[s]);
        });
        st(K_PR, xr); st(K_M2, m2); st(K_MD, mdyn);
      }
    }
  };
  auto finish = [&](const int status, const int acc, const int rej) {
    if (A.metric) {
      const double m1 = ld(K_M1), m2 = ld(K_M2), mdyn = ld(K_MD), shift = ld(K_SH);
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
      if (l == 0) A.metric[rep] = mm;
    }
    if (l == 0) {
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
  auto norm = [&](const double (&e)[RPL], const double eR, const double (&ya)[RPL], const double yRa, const double (&yb)[RPL], const double yRb) {
    double m = ratio(eR, yRa, yRb);
    static_for<RPL>([&](auto sc) { constexpr int s = decltype(sc)::value; m = mxn(m, ratio(e[s], ya[s], yb[s])); });
    return gmax<G>(m, lane);
  };
  // f(Y) of the lane's rows.  Bit j < LG: XOR-partner lane, same slot; bit j >= LG: slot s ^ (1 << (j - LG)) of the same lane.
  auto rhs_rows = [&](const double (&Y)[RPL], const double YR, double (&f)[RPL]) {
    static_for<RPL>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      const int m = l + G * s;
      const double ci = ld(K_CI + s);
      double a = -ld(K_DG + s) * Y[s];
      static_for<NB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const double w = (m & (1 << j)) ? ci : 1.0;
        double nbv;
        if constexpr (j < LG) nbv = xor_partner<(1 << j)>(Y[s]);
        else nbv = Y[s ^ (1 << (j - LG))];
        a = __builtin_fma(w, nbv, a);
      });
      f[s] = a;
    });
    if (l == 0) f[0] = __builtin_fma(cC, YR, f[0]);
  };

  double tc = A.t[0];
  int k = 1;
  double te = A.t[1];
  double h;
  {
    double f0[RPL];
    rhs_rows(y, yR, f0);
    const double fR = __builtin_fma(-cB, yR, cA);
    const double d0 = norm(y, yR, y, yR, y, yR);
    const double d1 = norm(f0, fR, y, yR, y, yR);
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
    if (A.h0 > 0.0) h = A.h0;
    if (!(h > 0.0) || h != h) h = 1e-6;
  }

  // M = I - q J (NM x NM), rows l + G * s per lane, inverted in registers (Gauss-Jordan, no pivoting: M-matrix)
  double a[RPL][NM];
  double winvR, qC;
  auto factor = [&](const double q) {
    winvR = fast_rcp(__builtin_fma(q, cB, 1.0));
    qC = q * cC;
    static_for<RPL>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      const int m = l + G * s;
      const double ci = ld(K_CI + s), dgs = ld(K_DG + s);
      static_for<NM>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        const int d = m ^ c;
        double v = 0.0;
        if (d != 0 && (d & (d - 1)) == 0) v = -q * ((m & d) ? ci : 1.0);
        if (c == m) v = __builtin_fma(q, dgs, 1.0);
        a[s][c] = v;
      });
    });
    static_for<NM>([&](auto kc) {
      constexpr int kk = decltype(kc)::value;
      constexpr int kl = kk & (G - 1), ks = kk / G;              // pivot row lives in lane kl, slot ks
      const double rp = fast_rcp(bcast<G, kl>(a[ks][kk]));
      double ml[RPL];
      static_for<RPL>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        ml[s] = (s == ks && l == kl) ? 1.0 - rp : a[s][kk] * rp;
      });
      static_for<NM>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j != kk) {
          const double u = bcast<G, kl>(a[ks][j]);
          static_for<RPL>([&](auto sc) { constexpr int s = decltype(sc)::value; a[s][j] = __builtin_fma(-ml[s], u, a[s][j]); });
        }
      });
      static_for<RPL>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        a[s][kk] = (s == ks && l == kl) ? rp : -ml[s];
      });
    });
  };
  auto solve = [&](const double (&r)[RPL], const double rR, double (&z)[RPL], double& zR) {
    zR = rR * winvR;
    double rr[RPL];
    static_for<RPL>([&](auto sc) { constexpr int s = decltype(sc)::value; rr[s] = r[s]; });
    if (l == 0) rr[0] = __builtin_fma(qC, zR, rr[0]);            // move the -q C z_R coupling of row P to the right
    double x[RPL];
    static_for<RPL>([&](auto sc) { x[decltype(sc)::value] = 0.0; });
    static_for<NM>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const double bj = bcast<G, (j & (G - 1))>(rr[j / G]);      // entry j of the right-hand side
      static_for<RPL>([&](auto sc) { constexpr int s = decltype(sc)::value; x[s] = __builtin_fma(a[s][j], bj, x[s]); });
    });
    static_for<RPL>([&](auto sc) { constexpr int s = decltype(sc)::value; z[s] = x[s]; });
  };

  bool after_reject = false;
  while (true) {
    if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; fail_from(k); break; }
    const bool last = (tc + 1.0001 * h >= te);
    const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
    if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; fail_from(k); break; }
    factor(Tab::GAM * hs);

    double f[RPL], z[RPL], zR;
    rhs_rows(y, yR, f);
    static_for<RPL>([&](auto sc) { f[decltype(sc)::value] *= hs; });
    solve(f, hs * __builtin_fma(-cB, yR, cA), z, zR);
    double yn[RPL], e[RPL], ynR = __builtin_fma(Tab::B[0], zR, yR), eR = 0.0;
    static_for<RPL>([&](auto sc) { constexpr int s = decltype(sc)::value; yn[s] = __builtin_fma(Tab::B[0], z[s], y[s]); e[s] = 0.0; });
    static_for<Tab::NS - 1>([&](auto kc) {
      constexpr int kk = 1 + decltype(kc)::value;
      double zn[RPL], zRn;
      solve(z, zR, zn, zRn);
      zR = zRn;
      ynR = __builtin_fma(Tab::B[kk], zR, ynR); eR = __builtin_fma(Tab::E[kk], zR, eR);
      static_for<RPL>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        z[s] = zn[s];
        yn[s] = __builtin_fma(Tab::B[kk], z[s], yn[s]);
        e[s] = __builtin_fma(Tab::E[kk], z[s], e[s]);
      });
    });

    const double err = norm(e, eR, y, yR, yn, ynR);
    if (err != err || err > 1e300) {
      ++nrej; after_reject = true; h = 0.1 * hs;
      bool nf = nonfinite(yR) || nonfinite(cA) || nonfinite(cB) || nonfinite(cC);
      static_for<RPL>([&](auto sc) { constexpr int s = decltype(sc)::value; nf = nf || nonfinite(y[s]) || nonfinite(ld(K_DG + s)) || nonfinite(ld(K_CI + s)); });
      if (gmax<G>(nf ? 1.0 : 0.0, lane) != 0.0) { status |= PK_ST_NONFINITE; fail_from(k); break; }
      continue;
    }
    double fac = root_q(err, Tab::Q) * (1.0 / 0.9);
    fac = fmax(1.0 / 6.0, fmin(5.0, fac));
    double hnew = hs * fast_rcp(fac);
    if (err <= 1.0) {
      ++nacc;
      static_for<RPL>([&](auto sc) { constexpr int s = decltype(sc)::value; y[s] = yn[s]; });
      yR = ynR; tc += hs;
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

template <int RPL> constexpr size_t rand_fastr_lds_bytes() { return (size_t)(3 * RPL + 5) * 256 * sizeof(double); }

}  // namespace pk
