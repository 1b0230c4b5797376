// pk_rand_fast2.hpp -- random model with n = 5 sites (32 coupled bit-mask rows): TWO rows per lane, 16 lanes per replica.
//
// pk_rand_fast.hpp runs n = 5 with one row per lane in 32-lane groups, whose cross-lane broadcasts are ds_swizzle trips through the LDS
// crossbar (2 048 per Gauss-Jordan inversion per wave): the kernel is bound by that pipe.  Here lane l of a 16-lane group owns the masks
// l (slot 0) and l + 16 (slot 1), so that
//   * every broadcast of the inversion and of the mat-vec solves is a DPP row_newbcast move (VALU only), 4 replicas per wave;
//   * the bit-4 neighbour of a mask is the lane's own other slot; bits 0..3 are XOR partners inside the group.
// Method, coefficients, controller and outputs are those of pk_rand_fast.hpp (reference: models/randmod.py:122-247).
#pragma once
#include "pk_solve_kernel.hpp"

namespace pk {

template <int METHOD>
__global__ __launch_bounds__(256, 2) void rand_fast2_kernel(const SolveArgs A) {
  using Tab = ResolventTab<METHOD>;
  constexpr int NB = 5, NM = 32, G = 16, RPB = 256 / G;
  const int lane = lane_id();
  const int l = threadIdx.x & (G - 1);
  const long long rep = (long long)blockIdx.x * RPB + (threadIdx.x / G);
  if (rep >= A.B) return;
  const int n = NB, S = A.S, T = A.T;
  const double* __restrict__ th = A.theta + rep * A.P;
  const double* Sr = th + 4;
  const double* Dd = th + 4 + n;
  const double cA = th[0], cB = th[1], cC = th[2];

  // ---- coefficients of the lane's two rows: m0 = l, m1 = l + 16
  double dgn[2], cin[2], nb[2][NB];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int m = l + G * s;
    if (m == 0) {
      double sumS = 0.0;
      for (int j = 0; j < n; ++j) sumS += Sr[j];
      dgn[s] = th[3] + sumS; cin[s] = 0.0;
    } else {
      const int lsb = __builtin_ctz(m);
      cin[s] = Sr[lsb];
      double out = 0.0;
      for (int j = 0; j < n; ++j) out += (m & (1 << j)) ? 1.0 : Sr[j < lsb ? j : lsb];
      dgn[s] = out + Dd[m - 1];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) nb[s][j] = (m & (1 << j)) ? cin[s] : 1.0;
  }

  const double* y0p = A.y0 + (A.y0_batched ? rep * S : 0);
  double y[2] = {y0p[1 + l], y0p[1 + l + G]};
  double yR = y0p[0];

  // ---- outputs / fused metric: observables are R, P (mask 0) and masks 1..5, all in slot 0
  const bool obs = l <= n;
  const int T5 = T > 5 ? T - 5 : 0;
  double m1 = 0.0, m2 = 0.0, mdyn = 0.0, shift = 0.0, prev = 0.0, prevR = 0.0;
  auto emit = [&](const int k, const double (&v)[2], const double vRaw, const bool nan_fill) {
    auto val = [&](double x, int state) {
      if (nan_fill) return __builtin_nan("");
      double r = A.clip ? ((x < 0.0) ? 0.0 : x) : x;
      if (A.normalize) r *= 1.0 / y0p[state];
      return r;
    };
    const double v0 = val(v[0], 1 + l), v1 = val(v[1], 1 + l + G), vR = val(vRaw, 0);
    if (A.sol) {
      double* solp = A.sol + (rep * T + k) * S;
      solp[1 + l] = v0; solp[1 + l + G] = v1;
      if (l == 0) solp[0] = vR;
    }
    if (A.flat) {
      double* fl = A.flat + rep * A.F;
      if (l == 0) { if (k >= 5) fl[k - 5] = vR; fl[T5 + k] = v0; }
      else if (obs) fl[T5 + T + (l - 1) * T + k] = v0;
    }
    if (A.metric) {
      const double x = obs ? v0 : 0.0;
      const double xr = (l == 0) ? vR : 0.0;
      if (k == 0) { shift = gsum<G>(x + xr, lane) / (2 + n); prev = x; prevR = xr; }
      m1 += x + xr;
      const double a = obs ? x - shift : 0.0, b = (l == 0) ? xr - shift : 0.0;
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
  // f(Y) of the lane's two rows.  Bits 0..3: XOR partner lanes, same slot; bit 4: the lane's other slot.
  auto rhs_rows = [&](const double (&Y)[2], const double YR, double (&f)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      double a = -dgn[s] * Y[s];
      a = __builtin_fma(nb[s][0], xor_partner<1>(Y[s]), a);
      a = __builtin_fma(nb[s][1], xor_partner<2>(Y[s]), a);
      a = __builtin_fma(nb[s][2], xor_partner<4>(Y[s]), a);
      a = __builtin_fma(nb[s][3], xor_partner<8>(Y[s]), a);
      a = __builtin_fma(nb[s][4], Y[s ^ 1], a);
      f[s] = a;
    }
    if (l == 0) f[0] = __builtin_fma(cC, YR, f[0]);
  };

  double tc = A.t[0];
  int k = 1;
  double te = A.t[1];
  double h;
  {
    double f0[2];
    rhs_rows(y, yR, f0);
    const double fR = __builtin_fma(-cB, yR, cA);
    const double d0 = gmax<G>(mxn(mxn(ratio(y[0], y[0], y[0]), ratio(y[1], y[1], y[1])), ratio(yR, yR, yR)), lane);
    const double d1 = gmax<G>(mxn(mxn(ratio(f0[0], y[0], y[0]), ratio(f0[1], y[1], y[1])), ratio(fR, yR, yR)), lane);
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
    if (A.h0 > 0.0) h = A.h0;
    if (!(h > 0.0) || h != h) h = 1e-6;
  }

  // M = I - q J (32 x 32), rows l and l + 16 per lane, inverted in registers (Gauss-Jordan, no pivoting: M-matrix)
  double a0[NM], a1[NM];
  double winvR, qC;
  auto factor = [&](const double q) {
    winvR = fast_rcp(__builtin_fma(q, cB, 1.0));
    qC = q * cC;
    static_for<NM>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      {
        const int m = l, d = m ^ c;
        double v = 0.0;
        if (d != 0 && (d & (d - 1)) == 0) v = -q * ((m & d) ? cin[0] : 1.0);
        if (c == m) v = __builtin_fma(q, dgn[0], 1.0);
        a0[c] = v;
      }
      {
        const int m = l + G, d = m ^ c;
        double v = 0.0;
        if (d != 0 && (d & (d - 1)) == 0) v = -q * ((m & d) ? cin[1] : 1.0);
        if (c == m) v = __builtin_fma(q, dgn[1], 1.0);
        a1[c] = v;
      }
    });
    static_for<NM>([&](auto kc) {
      constexpr int kk = decltype(kc)::value;
      constexpr int kl = kk & (G - 1);
      constexpr bool hi = kk >= G;
      const double rp = fast_rcp(bcast<G, kl>(hi ? a1[kk] : a0[kk]));
      const double ml0 = (!hi && l == kl) ? 1.0 - rp : a0[kk] * rp;
      const double ml1 = (hi && l == kl) ? 1.0 - rp : a1[kk] * rp;
      static_for<NM>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j != kk) {
          const double u = bcast<G, kl>(hi ? a1[j] : a0[j]);
          a0[j] = __builtin_fma(-ml0, u, a0[j]);
          a1[j] = __builtin_fma(-ml1, u, a1[j]);
        }
      });
      a0[kk] = (!hi && l == kl) ? rp : -ml0;
      a1[kk] = (hi && l == kl) ? rp : -ml1;
    });
  };
  auto solve = [&](const double (&r)[2], const double rR, double (&z)[2], double& zR) {
    zR = rR * winvR;
    const double r0 = (l == 0) ? __builtin_fma(qC, zR, r[0]) : r[0];      // move the -q C z_R coupling of row P to the right
    double x0 = 0.0, x1 = 0.0;
    static_for<G>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const double b0 = bcast<G, j>(r0), b1 = bcast<G, j>(r[1]);          // entries j and j + 16 of the right-hand side
      x0 = __builtin_fma(a0[j], b0, x0); x0 = __builtin_fma(a0[j + G], b1, x0);
      x1 = __builtin_fma(a1[j], b0, x1); x1 = __builtin_fma(a1[j + G], b1, x1);
    });
    z[0] = x0; z[1] = x1;
  };

  bool after_reject = false;
  while (true) {
    if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; fail_from(k); break; }
    const bool last = (tc + 1.0001 * h >= te);
    const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
    if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; fail_from(k); break; }
    factor(Tab::GAM * hs);

    double f[2], z[2], zR;
    rhs_rows(y, yR, f);
    f[0] *= hs; f[1] *= hs;
    solve(f, hs * __builtin_fma(-cB, yR, cA), z, zR);
    double yn[2] = {__builtin_fma(Tab::B[0], z[0], y[0]), __builtin_fma(Tab::B[0], z[1], y[1])};
    double ynR = __builtin_fma(Tab::B[0], zR, yR);
    double e[2] = {0.0, 0.0}, eR = 0.0;
    static_for<Tab::NS - 1>([&](auto kc) {
      constexpr int kk = 1 + decltype(kc)::value;
      double zn[2], zRn;
      solve(z, zR, zn, zRn);
      z[0] = zn[0]; z[1] = zn[1]; zR = zRn;
      yn[0] = __builtin_fma(Tab::B[kk], z[0], yn[0]); yn[1] = __builtin_fma(Tab::B[kk], z[1], yn[1]); ynR = __builtin_fma(Tab::B[kk], zR, ynR);
      e[0] = __builtin_fma(Tab::E[kk], z[0], e[0]); e[1] = __builtin_fma(Tab::E[kk], z[1], e[1]); eR = __builtin_fma(Tab::E[kk], zR, eR);
    });

    const double err = gmax<G>(mxn(mxn(ratio(e[0], y[0], yn[0]), ratio(e[1], y[1], yn[1])), ratio(eR, yR, ynR)), lane);
    if (err != err || err > 1e300) {
      ++nrej; after_reject = true; h = 0.1 * hs;
      const double bad = gmax<G>(((nonfinite(y[0])) || (nonfinite(y[1])) || (nonfinite(yR)) || (nonfinite(dgn[0])) || (nonfinite(dgn[1])) ||
                                  (nonfinite(cin[0])) || (nonfinite(cin[1])) || (nonfinite(cA)) || (nonfinite(cB)) || (nonfinite(cC))) ? 1.0 : 0.0, lane);
      if (bad != 0.0) { status |= PK_ST_NONFINITE; fail_from(k); break; }
      continue;
    }
    double fac = root_q(err, Tab::Q) * (1.0 / 0.9);
    fac = fmax(1.0 / 6.0, fmin(5.0, fac));
    double hnew = hs * fast_rcp(fac);
    if (err <= 1.0) {
      ++nacc;
      y[0] = yn[0]; y[1] = yn[1]; yR = ynR; tc += hs;
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
