// pk_tpr.hpp -- thread-per-replica kernels for SMALL systems: tpr_kernel<MODEL, NS, METHOD>, MODEL in {distributive, successive}.
//
// The lane-group kernels (pk_dist_fast.hpp, pk_solve_kernel.hpp) spread one replica over 4-16 lanes, which is what a 32-state system
// needs to fit the register file.  Proteins with a handful of sites (the common case in the reference's data; BASELINE config 1: n = 4)
// fit into ONE lane: state, stage vectors and factors of an (NS + 2)-row system live in that lane's registers, there is no cross-lane
// instruction at all (no reduction, no shadow rows), and a wave integrates 64 replicas.  Instructions per replica drop 3-7x against the
// group kernels; the price is parallelism -- 64 replicas per wave means B >= ~32 768 before every SIMD has a wave, so the dispatcher
// (pk_capi.hip) uses these kernels only for large batches and falls back to the group kernels otherwise.
//   distributive: arrow elimination, the site sum is a per-lane tree sum;   successive: Thomas algorithm on the (n + 2)-row chain.
// Same method (resolvent form), coefficients, step controller, landing rule, outputs and flags as the group kernels.
// The slow-changing values (rate coefficients, metric bookkeeping) are parked in LDS, slot-major (thread-private, no barrier).
#pragma once
#include "pk_solve_kernel.hpp"

namespace pk {

// STAGE (the smallest systems only): a replica's output rows are S <= 6 doubles, 48 bytes at a 672-byte stride between the lanes of a
// wave -- every store instruction touches 64 cache lines and leaves each of them partially written (measured: 1.39x the algorithmic
// write traffic, profiles/r02_c_tpr_pmc.json).  With STAGE a lane collects its rows in a thread-private 128-byte line buffer in LDS
// (slot-major, conflict-free) and writes a line only when it is complete (the block [T, S] of a replica is contiguous: its lines fill
// up in order as the output times arrive), as eight 16-byte stores to ONE line.
template <int NS, bool STAGE = false> constexpr size_t tpr_lds_bytes() { return (size_t)(3 * (NS + 2) + 4 + (STAGE ? 16 : 0)) * 256 * sizeof(double); }

template <int MODEL, int NS, int METHOD, bool STAGE = false>
__global__ __launch_bounds__(256) void tpr_kernel(const SolveArgs A) {
  static_assert(MODEL == M_DIST || MODEL == M_SUCC, "thread-per-replica kernels: distributive and successive models");
  using Tab = ResolventTab<METHOD>;
  constexpr int NR = NS + 2;                                 // rows: R, P, sites 1..NS (rows beyond n_sites are inert)
  const long long rep = (long long)blockIdx.x * 256 + threadIdx.x;
  if (rep >= A.B) return;
  const int n = A.n_sites, S = A.S, T = A.T;
  const double* __restrict__ th = A.theta + rep * A.P;

  extern __shared__ __align__(16) double park_lds[];
  double* const park = park_lds + threadIdx.x;
  // slots: c1[NR] (coupling to the previous row / to P), dgn[NR] (-J[i][i]), prev[NR], then m1, m2, mdyn, shift
  constexpr int K_C1 = 0, K_DG = NR, K_PV = 2 * NR, K_M1 = 3 * NR, K_M2 = 3 * NR + 1, K_MD = 3 * NR + 2, K_SH = 3 * NR + 3;
  auto ld = [&](int k) { return park[k * 256]; };
  auto st = [&](int k, double v) { park[k * 256] = v; };

  // ---- coefficients.  dist: c1[2+j] = S_j, dgn[1] = D + sum S, dgn[2+j] = 1 + D_j.
  //      succ (models/succmod.py:9-90): c1[1] = C, c1[2+j] = S_j (inflow from the previous level), super-diagonal 1 for rows 1..n,
  //      dgn[1] = D + S_1, dgn[2+j] = 1 + S_{j+1} + D_j (last level: 1 + D_n)
  const double cA = th[0], cB = th[1], cC = th[2];
  {
    double sumS = 0.0;
    for (int j = 0; j < n; ++j) sumS += th[4 + j];
    st(K_C1 + 0, 0.0); st(K_DG + 0, cB);
    st(K_C1 + 1, cC);
    if (MODEL == M_DIST) st(K_DG + 1, th[3] + sumS);
    else                 st(K_DG + 1, th[3] + (n > 0 ? th[4] : 0.0));
    static_for<NS>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const bool ok = j < n;
      st(K_C1 + 2 + j, ok ? th[4 + j] : 0.0);
      double d;
      if (MODEL == M_DIST) d = ok ? 1.0 + th[4 + n + j] : 1.0;
      else d = ok ? ((j == n - 1) ? 1.0 + th[4 + n + j] : 1.0 + th[4 + j + 1] + th[4 + n + j]) : 1.0;
      st(K_DG + 2 + j, d);
    });
  }

  const double* y0p = A.y0 + (A.y0_batched ? rep * S : 0);
  double y[NR];
  static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; y[i] = (i < S) ? y0p[i] : 0.0; });

  // ---- output / fused Morris metric (observables = every row: R, P, sites)
  static_for<NR + 4>([&](auto kc) { st(K_PV + decltype(kc)::value, 0.0); });
  const int T5 = T > 5 ? T - 5 : 0;
  // line buffer of the staged path: 16 slots behind the parked values; element e of `sol` (global index) sits in slot e & 15
  double* const lbuf = park + (size_t)(3 * NR + 4) * 256;
  const long long e_first = rep * (long long)T * S, e_last = e_first + (long long)T * S - 1;
  auto stage_store = [&](const long long e, const double r) {
    const int slot = (int)(e & 15);
    lbuf[slot * 256] = r;
    if (slot == 15 || e == e_last) {                            // the line is complete (or the replica's block ends inside it): write it out
      const long long line = e & ~15LL;
      const int lo = (line < e_first) ? (int)(e_first - line) : 0;      // a block does not start on a line boundary: skip the neighbour's part
      double* g = A.sol + line;
      if (lo == 0 && slot == 15) {
        using d2 = double __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int q = 0; q < 8; ++q) { d2 w; w.x = lbuf[(2 * q) * 256]; w.y = lbuf[(2 * q + 1) * 256]; *(d2*)(g + 2 * q) = w; }
      } else {
        for (int q = lo; q <= slot; ++q) g[q] = lbuf[q * 256];
      }
    }
  };
  auto emit = [&](const int k, const double (&v)[NR], const bool nan_fill) {
    double* solp = A.sol ? A.sol + (rep * T + k) * S : nullptr;
    double* fl = A.flat ? A.flat + rep * A.F : nullptr;
    double x[NR], loc = 0.0;
    static_for<NR>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      double r = 0.0;
      if (i < S) {
        r = nan_fill ? __builtin_nan("") : (A.clip ? ((v[i] < 0.0) ? 0.0 : v[i]) : v[i]);
        if (A.normalize && !nan_fill) r *= 1.0 / y0p[i];
        if constexpr (STAGE) { if (solp) stage_store(e_first + (long long)k * S + i, r); }
        if (fl) {
          if (i == 0) { if (k >= 5) fl[k - 5] = r; }
          else if (i == 1) fl[T5 + k] = r;
          else fl[T5 + T + (i - 2) * T + k] = r;
        }
      }
      x[i] = r; loc += r;
    });
    if constexpr (!STAGE) {
      // the row as ALIGNED 16-byte stores (element pairs whose global index starts even; a single element at an odd start / end): three
      // dwordx4 stores instead of six dwordx2 at S = 6 -- WRITE_SIZE counts 16-byte stores exactly, 8-byte ones as partial writes
      if (solp) {
        using d2 = double __attribute__((ext_vector_type(2)));
        const int odd = (int)((e_first + (long long)k * S) & 1);
        if (odd) solp[0] = x[0];
        static_for<(NR + 1) / 2>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const int i0 = 2 * q + odd;                                // first element of the pair (run time: 2 q or 2 q + 1)
          double a, b2;
          if constexpr (2 * q + 1 < NR) a = odd ? x[2 * q + 1] : x[2 * q]; else a = x[2 * q];
          if constexpr (2 * q + 2 < NR) b2 = odd ? x[2 * q + 2] : x[2 * q + 1];
          else if constexpr (2 * q + 1 < NR) b2 = x[2 * q + 1];
          else b2 = 0.0;
          if (i0 + 1 < S) { d2 w; w.x = a; w.y = b2; *(d2*)(solp + i0) = w; }
          else if (i0 < S) solp[i0] = a;
        });
      }
    }
    if (A.metric) {
      st(K_M1, ld(K_M1) + loc);
      if (!(A.metric_id == PK_METRIC_TOTAL_SIGNAL || A.metric_id == PK_METRIC_MEAN_ACTIVITY)) {
        double m2 = ld(K_M2), mdyn = ld(K_MD), shift = ld(K_SH);
        if (k == 0) {
          shift = loc / (2 + n);
          st(K_SH, shift);
          static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; st(K_PV + i, x[i]); });
        }
        static_for<NR>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          const double a = (i < S) ? x[i] - shift : 0.0;
          m2 = __builtin_fma(a, a, m2);
          const double d = x[i] - ld(K_PV + i);
          mdyn = __builtin_fma(d, d, mdyn);
          st(K_PV + i, x[i]);
        });
        st(K_M2, m2); st(K_MD, mdyn);
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
  auto fail_from = [&](int k) { for (; k < T; ++k) emit(k, y, true); };

  emit(0, y, false);
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  if (T < 2) { finish(status, 0, 0); return; }

  const double rtol = A.rtol, atol = A.atol;
  auto norm = [&](const double (&e)[NR], const double (&ya)[NR], const double (&yb)[NR]) {
    double m = 0.0;
    static_for<NR>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const double q = fabs(e[i]) * approx_rcp(__builtin_fma(rtol, fmax(fabs(ya[i]), fabs(yb[i])), atol));
      m = (q > m || q != q) ? q : m;
    });
    return m;
  };
  // f(Y)
  auto rhs_of = [&](const double (&Y)[NR], double (&f)[NR]) {
    f[0] = __builtin_fma(-cB, Y[0], cA);
    if (MODEL == M_DIST) {
      double sv[NS];
      static_for<NS>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        sv[j] = Y[2 + j];
        f[2 + j] = __builtin_fma(ld(K_C1 + 2 + j), Y[1], -ld(K_DG + 2 + j) * Y[2 + j]);
      });
      f[1] = __builtin_fma(cC, Y[0], __builtin_fma(-ld(K_DG + 1), Y[1], tree_sum(sv)));
    } else {
      // row i >= 1: c1_i Y[i-1] - dgn_i Y[i] + up_i Y[i+1], up_i = 1 for 1 <= i < S - 1
      static_for<NR - 1>([&](auto ic) {
        constexpr int i = 1 + decltype(ic)::value;
        double v = __builtin_fma(ld(K_C1 + i), Y[i - 1], -ld(K_DG + i) * Y[i]);
        if constexpr (i + 1 < NR) { if (i + 1 < S) v += Y[i + 1]; }
        f[i] = (i < S) ? v : 0.0;
      });
    }
  };

  double tc = A.t[0];
  int k = 1;
  double te = A.t[1];
  double h;
  {
    double f0[NR];
    rhs_of(y, f0);
    const double d0 = norm(y, y, y), d1 = norm(f0, y, y);
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
    if (A.h0 > 0.0) h = A.h0;
    if (!(h > 0.0) || h != h) h = 1e-6;
  }

  // ---- factors of M = I - q J and the solve, all in this lane
  //   dist: winv[i] = 1 / (1 + q dgn_i) (i != 1), cw[j] = q S_j winv[2+j], Scw = sum cw, winv[1] = 1 / Schur pivot of row P
  //   succ: Thomas: lo[i] = (-q c1_i) winv[i-1], winv[i] = 1 / pivot_i (super-diagonal -q for rows 1 .. S-2)
  double winv[NR], fx[NR], Scw = 0.0, qq = 0.0;                // fx: cw (dist, entries 2..) or lo (succ)
  auto factor = [&](const double q) {
    qq = q;
    if (MODEL == M_DIST) {
      winv[0] = fast_rcp(__builtin_fma(q, cB, 1.0));
      double cwv[NS];
      static_for<NS>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        winv[2 + j] = fast_rcp(__builtin_fma(q, ld(K_DG + 2 + j), 1.0));
        cwv[j] = q * ld(K_C1 + 2 + j) * winv[2 + j];
        fx[2 + j] = cwv[j];
      });
      Scw = tree_sum(cwv);
      winv[1] = fast_rcp(__builtin_fma(q, ld(K_DG + 1) - Scw, 1.0));
      fx[0] = 0.0; fx[1] = 0.0;
    } else {
      winv[0] = fast_rcp(__builtin_fma(q, cB, 1.0));
      fx[0] = 0.0;
      static_for<NR - 1>([&](auto ic) {
        constexpr int i = 1 + decltype(ic)::value;
        const double up_prev = (i - 1 >= 1 && i < S) ? -q : 0.0;        // super-diagonal of row i - 1 (row 0 has none)
        const double lo = (i < S) ? (-q * ld(K_C1 + i)) * winv[i - 1] : 0.0;
        fx[i] = lo;
        winv[i] = fast_rcp(__builtin_fma(q, ld(K_DG + i), 1.0) - lo * up_prev);
      });
    }
  };
  auto solve = [&](const double (&r)[NR], double (&x)[NR]) {
    if (MODEL == M_DIST) {
      const double xR = r[0] * winv[0];
      double t[NS];
      static_for<NS>([&](auto jc) { constexpr int j = decltype(jc)::value; t[j] = r[2 + j] * winv[2 + j]; });
      const double xP = __builtin_fma(qq, __builtin_fma(cC, xR, tree_sum(t)), r[1]) * winv[1];
      x[0] = xR; x[1] = xP;
      static_for<NS>([&](auto jc) { constexpr int j = decltype(jc)::value; x[2 + j] = __builtin_fma(fx[2 + j], xP, t[j]); });
    } else {
      double g[NR];
      g[0] = r[0];
      static_for<NR - 1>([&](auto ic) { constexpr int i = 1 + decltype(ic)::value; g[i] = __builtin_fma(-fx[i], g[i - 1], r[i]); });
      x[NR - 1] = g[NR - 1] * winv[NR - 1];
      static_for<NR - 1>([&](auto ic) {
        constexpr int i = NR - 2 - decltype(ic)::value;
        const double up = (i >= 1 && i + 1 < S) ? -qq : 0.0;
        x[i] = __builtin_fma(-up, x[i + 1], g[i]) * winv[i];
      });
    }
  };

  bool after_reject = false;
  while (true) {
    if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; fail_from(k); break; }
    const bool last = (tc + 1.0001 * h >= te);
    const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
    if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; fail_from(k); break; }
    factor(Tab::GAM * hs);

    double f[NR], z[NR], yn[NR], e[NR];
    rhs_of(y, f);
    static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; f[i] *= hs; });
    solve(f, z);
    static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; yn[i] = __builtin_fma(Tab::B[0], z[i], y[i]); e[i] = 0.0; });
    static_for<Tab::NS - 1>([&](auto kc) {
      constexpr int kk = 1 + decltype(kc)::value;
      double zn[NR];
      solve(z, zn);
      static_for<NR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        z[i] = zn[i];
        yn[i] = __builtin_fma(Tab::B[kk], z[i], yn[i]);
        e[i] = __builtin_fma(Tab::E[kk], z[i], e[i]);
      });
    });

    const double err = norm(e, y, yn);
    if (err != err || err > 1e300) {
      ++nrej; after_reject = true; h = 0.1 * hs;
      bool nf = nonfinite(cA) || nonfinite(cB) || nonfinite(cC);
      static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; nf = nf || nonfinite(y[i]) || nonfinite(ld(K_C1 + i)) || nonfinite(ld(K_DG + i)); });
      if (nf) { status |= PK_ST_NONFINITE; fail_from(k); break; }
      continue;
    }
    double fac = root_q(err, Tab::Q) * (1.0 / 0.9);
    fac = fmax(1.0 / 6.0, fmin(5.0, fac));
    double hnew = hs * fast_rcp(fac);
    if (err <= 1.0) {
      ++nacc;
      static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; y[i] = yn[i]; });
      tc += hs;
      if (after_reject) hnew = fmin(hnew, hs);
      after_reject = false;
      if (last) {
        tc = te;
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
