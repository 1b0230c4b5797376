// pk_dist_fast.hpp -- throughput kernel for the distributive model (models/distmod.py:7-65), adaptive RODAS4 / LRP8.
//
// Same integrator, same arithmetic per state as solve_kernel<M_DIST, G, RODAS4, true> (arrow elimination), but laid
// out for the VALU instead of for generality -- measured on MI355X the generic kernel spends its time in the LDS
// crossbar (ds_swizzle broadcasts / reductions: 194 LDS-pipe instructions per step, profiles/r01_a_*):
//
//   * a replica is owned by G lanes (G = 4, 8 or 16), each lane holding RPL site rows in registers
//     (site i = lane + G * j), so S = 32 runs EIGHT replicas per wavefront instead of two;
//   * the two coupling rows (mRNA R and unphosphorylated protein P) are "shadowed": every lane of the group carries
//     them as wave-uniform-per-group scalars and updates them redundantly, so no broadcast is ever needed;
//   * the site sum that closes row P is tracked through the stage recurrences (it is linear in the stage vectors), so a
//     stage costs exactly ONE group reduction (inside the arrow solve), done with DPP moves only;
//   * no LDS-pipe instruction in the solve chain; LDS only as thread-private parking space in the PARK layouts (below).
//
// dR/dt = A - B R ; dP/dt = C R - (D + sum S_i) P + sum X_i ; dX_i/dt = S_i P - (1 + D_i) X_i
#pragma once
#include "pk_solve_kernel.hpp"

namespace pk {

template <int RPL>
struct Trk {               // one vector of the system as seen by a lane
  double s[RPL];           // this lane's site rows
  double R, P;             // shadow rows (identical in every lane of the group)
  double sg;               // sum over ALL sites of the group (identical in every lane)
};

template <int RPL>
__device__ __forceinline__ void trk_axpy(Trk<RPL>& acc, const double a, const Trk<RPL>& u) {
#pragma unroll
  for (int j = 0; j < RPL; ++j) acc.s[j] = __builtin_fma(a, u.s[j], acc.s[j]);
  acc.R = __builtin_fma(a, u.R, acc.R);
  acc.P = __builtin_fma(a, u.P, acc.P);
  acc.sg = __builtin_fma(a, u.sg, acc.sg);
}
template <int RPL>
__device__ __forceinline__ Trk<RPL> trk_scale(const double a, const Trk<RPL>& u) {
  Trk<RPL> r;
#pragma unroll
  for (int j = 0; j < RPL; ++j) r.s[j] = a * u.s[j];
  r.R = a * u.R; r.P = a * u.P; r.sg = a * u.sg;
  return r;
}

// Per-lane values that are touched once per step (site rates) or once per output (metric bookkeeping).  In registers by default;
// PARK = true keeps them in LDS (slot-major: slot * 256 + thread, conflict-free), which frees 6 * RPL + 12 VGPRs: what lets the
// 4-lane x 8-row layout (30 % fewer instructions per replica than 8 x 4) run two waves per SIMD instead of one.
template <int RPL, bool PARK> struct Parked;
template <int RPL> struct Parked<RPL, false> {
  double v[3 * RPL + 6];
  __device__ __forceinline__ explicit Parked(double*) {}
  template <int K> __device__ __forceinline__ double get() const { return v[K]; }
  template <int K> __device__ __forceinline__ void set(double x) { v[K] = x; }
};
template <int RPL> struct Parked<RPL, true> {
  double* base;
  __device__ __forceinline__ explicit Parked(double* lds) : base(lds + threadIdx.x) {}
  template <int K> __device__ __forceinline__ double get() const { return base[K * 256]; }
  template <int K> __device__ __forceinline__ void set(double x) { base[K * 256] = x; }
};
template <int RPL, bool PARK> constexpr size_t dist_fast_lds_bytes() { return PARK ? (size_t)(3 * RPL + 6) * 256 * sizeof(double) : 0; }

template <int G, int RPL, int METHOD, bool PARK = false, int MINB = (PARK ? 2 : 1)>
__global__ __launch_bounds__(256, MINB) void dist_fast_kernel(const SolveArgs A) {
  using Tab = ResolventTab<METHOD>;
  extern __shared__ __align__(16) double park_lds[];
  Parked<RPL, PARK> pk(park_lds);
  // slots: [0, RPL) S_i ; [RPL, 2 RPL) 1 + D_i ; [2 RPL, 3 RPL) previous site outputs ; then prevR, prevP, m1, m2, mdyn, shift
  constexpr int K_SR = 0, K_DG = RPL, K_PS = 2 * RPL, K_PR = 3 * RPL, K_PP = 3 * RPL + 1, K_M1 = 3 * RPL + 2, K_M2 = 3 * RPL + 3,
                K_MD = 3 * RPL + 4, K_SH = 3 * RPL + 5;
  constexpr int RPB = 256 / G;
  const int lane = lane_id();
  const int l = threadIdx.x & (G - 1);
  const long long rep = (long long)blockIdx.x * RPB + (threadIdx.x / G);
  if (rep >= A.B) return;
  const int n = A.n_sites, S = A.S, T = A.T;
  const double* __restrict__ th = A.theta + rep * A.P;

  // ---- coefficients: uniform (A, B, C, D + sum S) and per site (S_i, 1 + D_i); padding sites are inert (S = 0, d = 1)
  const double cA = th[0], cB = th[1], cC = th[2];
  double lsum = 0.0;
  static_for<RPL>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int i = l + G * j;
    const bool ok = i < n;
    const double sr = ok ? th[4 + i] : 0.0;
    pk.template set<K_SR + j>(sr);
    pk.template set<K_DG + j>(ok ? 1.0 + th[4 + n + i] : 1.0);
    lsum += sr;
  });
  const double Dsum = th[3] + gsum<G>(lsum, lane);

  // ---- state
  const double* y0p = A.y0 + (A.y0_batched ? rep * S : 0);
  Trk<RPL> y;
  y.R = y0p[0]; y.P = y0p[1];
  lsum = 0.0;
#pragma unroll
  for (int j = 0; j < RPL; ++j) {
    const int i = l + G * j;
    y.s[j] = (i < n) ? y0p[2 + i] : 0.0;
    lsum += y.s[j];
  }
  y.sg = gsum<G>(lsum, lane);

  // ---- output / fused Morris metric (same semantics as Emitter in pk_solve_kernel.hpp)
  static_for<RPL + 6>([&](auto kc) { pk.template set<K_PS + decltype(kc)::value>(0.0); });
  const int T5 = T > 5 ? T - 5 : 0;
  auto emit = [&](const int k, const Trk<RPL>& v, const bool nan_fill) {
    double* solp = A.sol ? A.sol + (rep * T + k) * S : nullptr;
    double* fl = A.flat ? A.flat + rep * A.F : nullptr;
    auto val = [&](double x, int state) {
      if (nan_fill) return __builtin_nan("");
      double r = A.clip ? ((x < 0.0) ? 0.0 : x) : x;
      if (A.normalize) r *= 1.0 / y0p[state];
      return r;
    };
    const double vR = val(v.R, 0), vP = val(v.P, 1);
    if (l == 0) {
      if (solp) { solp[0] = vR; solp[1] = vP; }
      if (fl) { if (k >= 5) fl[k - 5] = vR; fl[T5 + k] = vP; }
    }
    double vs[RPL];
    double loc = (l == 0) ? vR + vP : 0.0;
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
      const int i = l + G * j;
      vs[j] = (i < n) ? val(v.s[j], 2 + i) : 0.0;
      if (i < n) {
        if (solp) solp[2 + i] = vs[j];
        if (fl) fl[T5 + T + i * T + k] = vs[j];
      }
      loc += vs[j];
    }
    if (A.metric) {
      // total_signal / mean_activity need the running sum only; the second-moment and first-difference bookkeeping (and its LDS
      // traffic in the parked layouts) runs only for the metrics that use it -- metric_id is uniform across the launch
      const bool sum_only = (A.metric_id == PK_METRIC_TOTAL_SIGNAL || A.metric_id == PK_METRIC_MEAN_ACTIVITY);
      pk.template set<K_M1>(pk.template get<K_M1>() + loc);
      if (!sum_only) {
        double m2 = pk.template get<K_M2>(), mdyn = pk.template get<K_MD>(), shift = pk.template get<K_SH>();
        double prevR = pk.template get<K_PR>(), prevP = pk.template get<K_PP>();
        if (k == 0) {
          shift = gsum<G>(loc, lane) / (2 + n);
          pk.template set<K_SH>(shift);
          prevR = vR; prevP = vP;
          static_for<RPL>([&](auto jc) { constexpr int j = decltype(jc)::value; pk.template set<K_PS + j>(vs[j]); });
        }
        static_for<RPL>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const int i = l + G * j;
          const double xs = (i < n) ? vs[j] - shift : 0.0;
          m2 = __builtin_fma(xs, xs, m2);
          const double d = vs[j] - pk.template get<K_PS + j>();
          mdyn = __builtin_fma(d, d, mdyn);
          pk.template set<K_PS + j>(vs[j]);
        });
        if (l == 0) {
          const double a = vR - shift, b = vP - shift;
          m2 = __builtin_fma(a, a, m2); m2 = __builtin_fma(b, b, m2);
          const double dR = vR - prevR, dP = vP - prevP;
          mdyn = __builtin_fma(dR, dR, mdyn); mdyn = __builtin_fma(dP, dP, mdyn);
        }
        pk.template set<K_PR>(vR); pk.template set<K_PP>(vP);
        pk.template set<K_M2>(m2); pk.template set<K_MD>(mdyn);
      }
    }
  };
  auto finish = [&](const int status, const int acc, const int rej) {
    if (A.metric) {
      const double m1 = pk.template get<K_M1>(), m2 = pk.template get<K_M2>(), mdyn = pk.template get<K_MD>(), shift = pk.template get<K_SH>();
      const double L = 2.0 * T + (double)T * n;
      const double tot = gsum<G>(m1, lane);
      double m;
      switch (A.metric_id) {
        case PK_METRIC_TOTAL_SIGNAL: m = tot; break;
        case PK_METRIC_MEAN_ACTIVITY: m = tot / L; break;
        case PK_METRIC_VARIANCE: { const double q = gsum<G>(m2, lane); const double ms = tot / L - shift; m = q / L - ms * ms; } break;
        case PK_METRIC_DYNAMICS: m = gsum<G>(mdyn, lane); break;
        default: { const double q = gsum<G>(m2, lane); m = sqrt(fmax(q + 2.0 * shift * tot - L * shift * shift, 0.0)); } break;
      }
      if (l == 0) A.metric[rep] = m;
    }
    if (l == 0) {
      if (A.status) A.status[rep] = status;
      if (A.n_steps) { A.n_steps[2 * rep] = acc; A.n_steps[2 * rep + 1] = rej; }
    }
  };
  auto fail_from = [&](int k) { for (; k < T; ++k) emit(k, y, true); };

  emit(0, y, false);
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  if (T < 2) { finish(status, 0, 0); return; }

  const double rtol = A.rtol, atol = A.atol;
  // max-norm helpers over the whole system (sites of this lane + shadows, then across the group)
  auto group_max = [&](const Trk<RPL>& num, const Trk<RPL>& a, const Trk<RPL>& b) {
    auto q = [&](double e, double ya, double yb) { return fabs(e) * approx_rcp(__builtin_fma(rtol, fmax(fabs(ya), fabs(yb)), atol)); };
    auto mx = [](double p, double r) { return (p > r || p != p) ? p : r; };
    double m = mx(q(num.R, a.R, b.R), q(num.P, a.P, b.P));
#pragma unroll
    for (int j = 0; j < RPL; ++j) m = mx(m, q(num.s[j], a.s[j], b.s[j]));
    return gmax<G>(m, lane);
  };
  auto rhs_of = [&](const Trk<RPL>& Y) {            // f(Y); .sg unused
    Trk<RPL> f;
    f.R = __builtin_fma(-cB, Y.R, cA);
    f.P = __builtin_fma(cC, Y.R, __builtin_fma(-Dsum, Y.P, Y.sg));
    static_for<RPL>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      f.s[j] = __builtin_fma(pk.template get<K_SR + j>(), Y.P, -pk.template get<K_DG + j>() * Y.s[j]);
    });
    f.sg = 0.0;
    return f;
  };

  double tc = A.t[0];
  int k = 1;
  double te = A.t[1];
  double h;
  {
    const Trk<RPL> f0 = rhs_of(y);
    Trk<RPL> one = y;                                 // |y| / sc and |f0| / sc with sc = atol + rtol |y|
    const double d0 = group_max(y, y, y), d1 = group_max(f0, y, y);
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
    if (A.h0 > 0.0) h = A.h0;
    if (!(h > 0.0) || h != h) h = 1e-6;
    (void)one;
  }

  // Arrow factors of M = I - q J (q = gamma h) for the current step size.
  //   rows: (1 + q B) x_R = r_R ; (1 + q d_i) x_i - q S_i x_P = r_i ; (1 + q Dsum) x_P - q C x_R - q sum x_i = r_P
  double winv[RPL], cw[RPL], winvR, sinv, Scw, qq;
  auto factor = [&](const double q) {
    qq = q;
    winvR = fast_rcp(__builtin_fma(q, cB, 1.0));
    static_for<RPL>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      winv[j] = fast_rcp(__builtin_fma(q, pk.template get<K_DG + j>(), 1.0));
      cw[j] = q * pk.template get<K_SR + j>() * winv[j];
    });
    Scw = gsum<G>(tree_sum(cw), lane);
    sinv = fast_rcp(__builtin_fma(q, Dsum - Scw, 1.0));
  };
  // u = M^{-1} r   (r.sg ignored; u.sg = sum over sites of u): ONE group reduction
  auto solve = [&](const Trk<RPL>& r) {
    Trk<RPL> u;
    const double xR = r.R * winvR;
    double t[RPL];
#pragma unroll
    for (int j = 0; j < RPL; ++j) t[j] = r.s[j] * winv[j];
    const double St = gsum<G>(tree_sum(t), lane);
    const double xP = __builtin_fma(qq, __builtin_fma(cC, xR, St), r.P) * sinv;
#pragma unroll
    for (int j = 0; j < RPL; ++j) u.s[j] = __builtin_fma(cw[j], xP, t[j]);
    u.R = xR; u.P = xP; u.sg = __builtin_fma(xP, Scw, St);
    return u;
  };

  // resolvent-form step (the right-hand side is affine): z_1 = M^{-1} h f(y), z_{k+1} = M^{-1} z_k,
  //   y_new = y + sum_k B_k z_k ,  err = sum_k E_k z_k     (ResolventTab: RODAS4 or LRP8)
  bool after_reject = false;
  while (true) {
    if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; fail_from(k); break; }
    const bool last = (tc + 1.0001 * h >= te);
    const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
    if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; fail_from(k); break; }
    factor(Tab::GAM * hs);

    Trk<RPL> z = solve(trk_scale(hs, rhs_of(y)));
    Trk<RPL> yn = y; trk_axpy(yn, Tab::B[0], z);
    Trk<RPL> u6;
    static_for<Tab::NS - 1>([&](auto kc) {
      constexpr int kk = 1 + decltype(kc)::value;
      z = solve(z);
      trk_axpy(yn, Tab::B[kk], z);
      if constexpr (kk == 1) u6 = trk_scale(Tab::E[1], z); else trk_axpy(u6, Tab::E[kk], z);
    });

    const double err = group_max(u6, y, yn);
    if (err != err || err > 1e300) {
      ++nrej; after_reject = true; h = 0.1 * hs;
      const double bad = gmax<G>(((nonfinite(y.R)) || (nonfinite(y.P)) || (nonfinite(y.sg))) ? 1.0 : 0.0, lane);
      if (bad != 0.0 || (nonfinite(cA)) || (nonfinite(cB)) || (nonfinite(cC)) || (nonfinite(Dsum)) || (nonfinite(Scw))) {
        status |= PK_ST_NONFINITE; fail_from(k); break;
      }
      continue;
    }
    double fac = root_q(err, Tab::Q) * (1.0 / 0.9);
    fac = fmax(1.0 / 6.0, fmin(5.0, fac));
    double hnew = hs * fast_rcp(fac);
    if (err <= 1.0) {
      ++nacc;
      y = yn; tc += hs;
      if (after_reject) hnew = fmin(hnew, hs);
      after_reject = false;
      if (last) {
        tc = te;
        // re-sum the sites at every landing so the tracked sum cannot drift
        double loc = 0.0;
#pragma unroll
        for (int j = 0; j < RPL; ++j) loc += y.s[j];
        y.sg = gsum<G>(loc, lane);
        emit(k, y, false);
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
