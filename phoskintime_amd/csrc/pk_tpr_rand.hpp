// pk_tpr_rand.hpp -- thread-per-replica kernel for the random model with n <= 3 sites (2^n <= 8 coupled bit-mask rows):
// tpr_rand_kernel<NB, METHOD>.  Companion of pk_tpr.hpp: the whole replica lives in one lane -- the 2^n x 2^n matrix I - q J is inverted in
// that lane's registers by a fully unrolled Gauss-Jordan (no pivoting: M-matrix), every solve is a register mat-vec, there is no
// cross-lane instruction, and a wave integrates 64 replicas.  Same method, controller, outputs and flags as pk_rand_fastr.hpp
// (reference: models/randmod.py:122-247, incl. the lowest-set-bit rate quirk at :201).
#pragma once
#include "pk_solve_kernel.hpp"

namespace pk {

template <int NB> constexpr size_t tpr_rand_lds_bytes() { return (size_t)(3 * (1 << NB) + 5) * 256 * sizeof(double); }

template <int NB, int METHOD>
__global__ __launch_bounds__(256) void tpr_rand_kernel(const SolveArgs A) {
  using Tab = ResolventTab<METHOD>;
  constexpr int NM = 1 << NB;                                // masks 0 .. NM - 1 (mask 0 = unphosphorylated protein P); plus the mRNA row R
  const long long rep = (long long)blockIdx.x * 256 + threadIdx.x;
  if (rep >= A.B) return;
  const int n = NB, S = A.S, T = A.T;
  const double* __restrict__ th = A.theta + rep * A.P;
  const double* Sr = th + 4;
  const double* Dd = th + 4 + n;
  const double cA = th[0], cB = th[1], cC = th[2];

  extern __shared__ __align__(16) double park_lds[];
  double* const park = park_lds + threadIdx.x;
  constexpr int K_DG = 0, K_CI = NM, K_PV = 2 * NM, K_PR = 3 * NM, K_M1 = 3 * NM + 1, K_M2 = 3 * NM + 2, K_MD = 3 * NM + 3, K_SH = 3 * NM + 4;
  auto ld = [&](int k) { return park[k * 256]; };
  auto st = [&](int k, double v) { park[k * 256] = v; };

  static_for<NM>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    if constexpr (m == 0) {
      double sumS = 0.0;
      for (int j = 0; j < n; ++j) sumS += Sr[j];
      st(K_DG, th[3] + sumS); st(K_CI, 0.0);
    } else {
      constexpr int lsb = __builtin_ctz(m);
      st(K_CI + m, Sr[lsb]);
      double out = 0.0;
      for (int j = 0; j < n; ++j) out += (m & (1 << j)) ? 1.0 : Sr[j < lsb ? j : lsb];
      st(K_DG + m, out + Dd[m - 1]);
    }
  });

  const double* y0p = A.y0 + (A.y0_batched ? rep * S : 0);
  double y[NM], yR = y0p[0];
  static_for<NM>([&](auto mc) { constexpr int m = decltype(mc)::value; y[m] = y0p[1 + m]; });

  static_for<NM + 5>([&](auto kc) { st(K_PV + decltype(kc)::value, 0.0); });
  const int T5 = T > 5 ? T - 5 : 0;
  auto emit = [&](const int k, const double (&v)[NM], const double vRaw, const bool nan_fill) {
    auto val = [&](double x, int state) {
      if (nan_fill) return __builtin_nan("");
      double r = A.clip ? ((x < 0.0) ? 0.0 : x) : x;
      if (A.normalize) r *= 1.0 / y0p[state];
      return r;
    };
    double* solp = A.sol ? A.sol + (rep * T + k) * S : nullptr;
    double* fl = A.flat ? A.flat + rep * A.F : nullptr;
    const double vR = val(vRaw, 0);
    if (solp) solp[0] = vR;
    if (fl && k >= 5) fl[k - 5] = vR;
    double x[NM], loc = vR;
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const double vv = val(v[m], 1 + m);
      if (solp) solp[1 + m] = vv;
      if (fl) {
        if constexpr (m == 0) fl[T5 + k] = vv;
        else if constexpr (m <= NB) fl[T5 + T + (m - 1) * T + k] = vv;
      }
      x[m] = (m <= NB) ? vv : 0.0;                            // observables: R, P and the masks 1..n
      loc += x[m];
    });
    if (A.metric) {
      st(K_M1, ld(K_M1) + loc);
      if (!(A.metric_id == PK_METRIC_TOTAL_SIGNAL || A.metric_id == PK_METRIC_MEAN_ACTIVITY)) {
        double m2 = ld(K_M2), mdyn = ld(K_MD), shift = ld(K_SH), prevR = ld(K_PR);
        if (k == 0) {
          shift = loc / (2 + n);
          st(K_SH, shift);
          prevR = vR;
          static_for<NM>([&](auto mc) { constexpr int m = decltype(mc)::value; st(K_PV + m, x[m]); });
        }
        const double b = vR - shift;
        m2 = __builtin_fma(b, b, m2);
        const double dr = vR - prevR;
        mdyn = __builtin_fma(dr, dr, mdyn);
        static_for<NM>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          const double a = (m <= NB) ? x[m] - shift : 0.0;
          m2 = __builtin_fma(a, a, m2);
          const double d = x[m] - ld(K_PV + m);
          mdyn = __builtin_fma(d, d, mdyn);
          st(K_PV + m, x[m]);
        });
        st(K_PR, vR); st(K_M2, m2); st(K_MD, mdyn);
      }
    }
  };
  auto finish = [&](const int status, const int acc, const int rej) {
    if (A.metric) {
      const double m1 = ld(K_M1), m2 = ld(K_M2), mdyn = ld(K_MD), shift = ld(K_SH);
      const double L = 2.0 * T + (double)T * n;
      double mm;
      switch (A.metric_id) {
        case PK_METRIC_TOTAL_SIGNAL: mm = m1; break;
        case PK_METRIC_MEAN_ACTIVITY: mm = m1 / L; break;
        case PK_METRIC_VARIANCE: { const double ms = m1 / L - shift; mm = m2 / L - ms * ms; } break;
        case PK_METRIC_DYNAMICS: mm = mdyn; break;
        default: mm = sqrt(fmax(m2 + 2.0 * shift * m1 - L * shift * shift, 0.0)); break;
      }
      A.metric[rep] = mm;
    }
    if (A.status) A.status[rep] = status;
    if (A.n_steps) { A.n_steps[2 * rep] = acc; A.n_steps[2 * rep + 1] = rej; }
  };
  auto fail_from = [&](int k) { for (; k < T; ++k) emit(k, y, yR, true); };

  emit(0, y, yR, false);
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  if (T < 2) { finish(status, 0, 0); return; }

  const double rtol = A.rtol, atol = A.atol;
  auto ratio = [&](double e, double ya, double yb) { return fabs(e) * approx_rcp(__builtin_fma(rtol, fmax(fabs(ya), fabs(yb)), atol)); };
  auto norm = [&](const double (&e)[NM], const double eR, const double (&ya)[NM], const double yRa, const double (&yb)[NM], const double yRb) {
    double m = ratio(eR, yRa, yRb);
    static_for<NM>([&](auto mc) { constexpr int i = decltype(mc)::value; const double q = ratio(e[i], ya[i], yb[i]); m = (q > m || q != q) ? q : m; });
    return m;
  };
  auto rhs_rows = [&](const double (&Y)[NM], const double YR, double (&f)[NM]) {
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const double ci = ld(K_CI + m);
      double a = -ld(K_DG + m) * Y[m];
      static_for<NB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        a = __builtin_fma((m & (1 << j)) ? ci : 1.0, Y[m ^ (1 << j)], a);
      });
      f[m] = a;
    });
    f[0] = __builtin_fma(cC, YR, f[0]);
  };

  double tc = A.t[0];
  int k = 1;
  double te = A.t[1];
  double h;
  {
    double f0[NM];
    rhs_rows(y, yR, f0);
    const double fR = __builtin_fma(-cB, yR, cA);
    const double d0 = norm(y, yR, y, yR, y, yR), d1 = norm(f0, fR, y, yR, y, yR);
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
    if (A.h0 > 0.0) h = A.h0;
    if (!(h > 0.0) || h != h) h = 1e-6;
  }

  double a[NM][NM], winvR, qC;
  auto factor = [&](const double q) {
    winvR = fast_rcp(__builtin_fma(q, cB, 1.0));
    qC = q * cC;
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const double ci = ld(K_CI + m), dgs = ld(K_DG + m);
      static_for<NM>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int d = m ^ c;
        if constexpr (c == m) a[m][c] = __builtin_fma(q, dgs, 1.0);
        else if constexpr ((d & (d - 1)) == 0) a[m][c] = -q * ((m & d) ? ci : 1.0);
        else a[m][c] = 0.0;
      });
    });
    static_for<NM>([&](auto kc) {                              // in-place Gauss-Jordan inverse
      constexpr int kk = decltype(kc)::value;
      const double rp = fast_rcp(a[kk][kk]);
      static_for<NM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i != kk) {
          const double ml = a[i][kk] * rp;
          static_for<NM>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j != kk) a[i][j] = __builtin_fma(-ml, a[kk][j], a[i][j]);
          });
          a[i][kk] = -ml;
        }
      });
      static_for<NM>([&](auto jc) { constexpr int j = decltype(jc)::value; if constexpr (j != kk) a[kk][j] *= rp; });
      a[kk][kk] = rp;
    });
  };
  auto solve = [&](const double (&r)[NM], const double rR, double (&z)[NM], double& zR) {
    zR = rR * winvR;
    double rr[NM];
    static_for<NM>([&](auto mc) { constexpr int m = decltype(mc)::value; rr[m] = r[m]; });
    rr[0] = __builtin_fma(qC, zR, rr[0]);                      // the -q C z_R coupling of row P moved to the right-hand side
    static_for<NM>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      double x = a[i][0] * rr[0];
      static_for<NM - 1>([&](auto jc) { constexpr int j = 1 + decltype(jc)::value; x = __builtin_fma(a[i][j], rr[j], x); });
      z[i] = x;
    });
  };

  bool after_reject = false;
  while (true) {
    if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; fail_from(k); break; }
    const bool last = (tc + 1.0001 * h >= te);
    const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
    if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; fail_from(k); break; }
    factor(Tab::GAM * hs);

    double f[NM], z[NM], zR, yn[NM], e[NM];
    rhs_rows(y, yR, f);
    static_for<NM>([&](auto mc) { f[decltype(mc)::value] *= hs; });
    solve(f, hs * __builtin_fma(-cB, yR, cA), z, zR);
    double ynR = __builtin_fma(Tab::B[0], zR, yR), eR = 0.0;
    static_for<NM>([&](auto mc) { constexpr int m = decltype(mc)::value; yn[m] = __builtin_fma(Tab::B[0], z[m], y[m]); e[m] = 0.0; });
    static_for<Tab::NS - 1>([&](auto kc) {
      constexpr int kk = 1 + decltype(kc)::value;
      double zn[NM], zRn;
      solve(z, zR, zn, zRn);
      zR = zRn;
      ynR = __builtin_fma(Tab::B[kk], zR, ynR); eR = __builtin_fma(Tab::E[kk], zR, eR);
      static_for<NM>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        z[m] = zn[m];
        yn[m] = __builtin_fma(Tab::B[kk], z[m], yn[m]);
        e[m] = __builtin_fma(Tab::E[kk], z[m], e[m]);
      });
    });

    const double err = norm(e, eR, y, yR, yn, ynR);
    if (err != err || err > 1e300) {
      ++nrej; after_reject = true; h = 0.1 * hs;
      bool nf = nonfinite(yR) || nonfinite(cA) || nonfinite(cB) || nonfinite(cC);
      static_for<NM>([&](auto mc) { constexpr int m = decltype(mc)::value; nf = nf || nonfinite(y[m]) || nonfinite(ld(K_DG + m)) || nonfinite(ld(K_CI + m)); });
      if (nf) { status |= PK_ST_NONFINITE; fail_from(k); break; }
      continue;
    }
    double fac = root_q(err, Tab::Q) * (1.0 / 0.9);
    fac = fmax(1.0 / 6.0, fmin(5.0, fac));
    double hnew = hs * fast_rcp(fac);
    if (err <= 1.0) {
      ++nacc;
      static_for<NM>([&](auto mc) { constexpr int m = decltype(mc)::value; y[m] = yn[m]; });
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

}  // namespace pk
