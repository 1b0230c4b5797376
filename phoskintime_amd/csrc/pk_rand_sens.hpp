// pk_rand_sens.hpp -- forward parameter sensitivities of the random model at n = 6 and 7 sites (65 / 129 states, 73 / 138 parameters):
// rand_sens_kernel<NB>, one workgroup per (replica, chunk of columns).
//
// What it replaces: the 1 + P calls of models.solve_ode per Jacobian that scipy.optimize.curve_fit's '2-point' rule makes under
// paramest/normest.py:167-326 -- 139 solves of a 129-state system per Jacobian at n = 7 (randmod is the reference's default model and is
// fitted in log space, normest.py:54).  Reference right-hand side: models/randmod.py:122-247 (lowest-set-bit rate quirk at :201).
//
// Same mathematics as pk_sens.hpp (internal differentiation of the LRP12 resolvent step with the factors of the ONE matrix M = I - q J):
//     M z_1 = h f(y),  M z_{k+1} = z_k ;     M z'_1 = h (A y' + b' + A' (y + g z_1)),  M z'_{k+1} = z'_k + q A' z_{k+1}
// and the same exact solve as pk_rand_parity.hpp: the odd-popcount block of M is diagonal, the even Schur complement (32 x 32 / 64 x 64)
// is inverted ONCE per step in the registers of the workgroup (a TS x TS block per thread of a 16 x 16 grid) -- and then serves KC = 8
// right-hand sides per stage: the state and seven tangents, whose vectors live in LDS as [row][column].  That is the point of a
// sensitivity kernel here: differencing inverts the matrix once per step in EACH of its 1 + P replicas.
// The 1 + P columns of a replica are cut into ceil(P / 7) chunks (11 at n = 6, 20 at n = 7), each a workgroup that integrates the state
// again beside its tangents and controls its step size on its own columns (as pk_sens_rows.hpp does); chunk 0 writes flat / n_steps, every
// chunk ORs its flags into status.  The tangent columns run one stage behind the state (13 rounds of solves for 12 stages), because
// A'_c z_{k+1} needs the state's stage vector of the same round.
// A'_c is never stored: its three kinds of entries -- d(loss rate of mask m)/d theta_p (a small integer), d(inflow rate of m)/d theta_p
// (0 or 1), and the mRNA row / the C coupling -- are re-derived from (m, p) with a few integer instructions where they are used.
#pragma once
#include "pk_wide.hpp"
#include "pk_sens.hpp"

namespace pk {

constexpr int kRandSensKC = 8;                                  // 1 state + 7 tangent columns per workgroup
__host__ __device__ constexpr size_t rand_sens_lds_bytes(int n) {
  // five [S][KC] vectors (y, stage source / destination, y_new, error) | dg ci dio [2^n] | pivot row / column x 2 [2^(n-1)] | re [2^(n-1)][KC] | red
  const size_t NALL = (size_t)1 << n, S = NALL + 1, NM = NALL / 2;
  return (5 * S * kRandSensKC + 3 * NALL + 4 * NM + NM * kRandSensKC + 24) * sizeof(double);
}

// TBP: side of the thread grid over the even Schur complement: 16 (256 threads) or 8 (n = 6: ONE wave per column chunk, 4 x 4 blocks per lane)
template <int NB, int TBP = 16>
__global__ __launch_bounds__(TBP * TBP) void rand_sens_kernel(const SensArgs SA) {
  static_assert((NB == 6 || NB == 7) && (TBP == 16 || TBP == 8) && (1 << NB) <= TBP * TBP, "thread grid over the even Schur complement");
  using Tab = ResolventTab<PK_METHOD_LRP12>;
  constexpr int NALL = 1 << NB, NM = NALL / 2, TB = TBP, TS = NM / TB, NT = TB * TB, KC = kRandSensKC, KT = KC - 1, S = NALL + 1;
  const SolveArgs& A = SA.s;
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x, nt = NT;
  const int bi = tid / TB, bj = tid % TB, lane = tid & 63;
  const int T = A.T, P = A.P, F = A.F;
  const int nch = (P + KT - 1) / KT;
  const long long rep = blockIdx.x / nch;
  const int ch = (int)(blockIdx.x - rep * nch);
  if (rep >= A.B) return;
  const double* __restrict__ th = A.theta + rep * P;
  double* Y = lds;                double* ZS = Y + S * KC;        double* ZD = ZS + S * KC;
  double* YN = ZD + S * KC;       double* ER = YN + S * KC;
  double* dg = ER + S * KC;       double* ci = dg + NALL;         double* dio = ci + NALL;
  double* rowb = dio + NALL;      double* colb = rowb + 2 * NM;
  double* re = colb + 2 * NM;     double* red = re + NM * KC;
  auto emask = [](const int e) __attribute__((always_inline)) { return 2 * e + (__builtin_popcount(e) & 1); };
  auto omask = [](const int o) __attribute__((always_inline)) { return 2 * o + 1 - (__builtin_popcount(o) & 1); };
  auto wgt = [&](const int x, const int bit) __attribute__((always_inline)) { return (x & bit) ? ci[x] : 1.0; };
  const double* y0p = A.y0 + (A.y0_batched ? rep * S : 0);
  const double cA = th[0], cB = th[1], cC = th[2];
  // parameter of column c (c >= 1), or -1 for the state column and for columns beyond P
  auto pcol = [&](const int c) __attribute__((always_inline)) { const int p = ch * KT + c - 1; return (c >= 1 && p < P) ? p : -1; };
  // derivative coefficients of the cube row of mask m with respect to parameter p: ddg = d(loss rate)/d theta_p, dci = d(inflow rate)/d theta_p
  auto dcoef = [&](const int m, const int p, int& ddg, int& dci) __attribute__((always_inline)) {
    ddg = 0; dci = 0;
    if (p < 3) return;
    if (p == 3) { ddg = (m == 0) ? 1 : 0; return; }                        // D: protein degradation of the unphosphorylated state
    if (p < 4 + NB) {
      const int k = p - 4;
      if (m == 0) { ddg = 1; return; }                                     // loss of mask 0 = D + sum_j S_j
      const int lsb = __builtin_ctz(m);
      dci = (lsb == k) ? 1 : 0;                                            // inflow into m runs at S[lsb(m)]  (randmod.py:201)
      int cnt = 0;
#pragma unroll
      for (int j = 0; j < NB; ++j) if (!((m >> j) & 1)) cnt += ((j < lsb ? j : lsb) == k) ? 1 : 0;      // phosphorylating site j of m runs at S[lsb(m | 2^j)]
      ddg = cnt;
      return;
    }
    ddg = (m == p - (4 + NB) + 1) ? 1 : 0;                                 // Ddeg of mask m
  };

  if (tid < NALL) {
    const int m = tid;
    if (m == 0) {
      double sumS = 0.0;
      for (int j = 0; j < NB; ++j) sumS += th[4 + j];
      dg[0] = th[3] + sumS; ci[0] = 0.0;
    } else {
      const int lsb = __builtin_ctz(m);
      ci[m] = th[4 + lsb];
      double outr = 0.0;
      for (int j = 0; j < NB; ++j) outr += ((m >> j) & 1) ? 1.0 : th[4 + (j < lsb ? j : lsb)];
      dg[m] = outr + th[4 + NB + m - 1];
    }
  }
  for (int k = tid; k < S * KC; k += nt) Y[k] = ((k % KC) == 0) ? y0p[k / KC] : 0.0;      // the initial condition is data: zero tangents
  __syncthreads();

  // ---- outputs: flat = [R(t5..), P(t0..), the first n phospho columns = masks 1 .. n] (randmod.py:298-299) and its parameter derivatives
  const int T5 = T > 5 ? T - 5 : 0;
  double* const fl = A.flat + rep * F;
  double* const dfl = SA.dflat + rep * (long long)F * P;
  auto emit = [&](const int k, const bool nan_fill) __attribute__((always_inline)) {
    for (int it = tid; it < (2 + NB) * KC; it += nt) {
      const int row = it / KC, c = it % KC;
      const int fi = (row == 0) ? (k >= 5 ? k - 5 : -1) : (row == 1 ? T5 + k : T5 + T + (row - 2) * T + k);
      if (fi < 0) continue;
      const double sc = A.normalize ? 1.0 / y0p[row] : 1.0;
      const bool clipped = A.clip && (Y[row * KC] < 0.0);
      const double v = nan_fill ? __builtin_nan("") : (clipped ? 0.0 : Y[row * KC + c] * sc);
      if (c == 0) { if (ch == 0) fl[fi] = v; }
      else { const int p = pcol(c); if (p >= 0) dfl[(long long)fi * P + p] = v; }
    }
  };
  auto finish = [&](const int status, const int acc, const int rej) __attribute__((always_inline)) {
    if (tid != 0) return;
    if (A.status && status) atomicOr(&A.status[rep], status);            // zeroed by the launcher
    if (ch == 0 && A.n_steps) { A.n_steps[2 * rep] = acc; A.n_steps[2 * rep + 1] = rej; }
  };
  auto fail_from = [&](int kk) __attribute__((always_inline)) { for (; kk < T; ++kk) emit(kk, true); };

  emit(0, false);
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  if (T < 2) { finish(status, 0, 0); return; }
  const double rtol = A.rtol, atol = A.atol;

  // (A V)[row] of column c of a [S][KC] vector (the homogeneous part of the right-hand side)
  auto apply_A = [&](const double* V, const int c, const int row) __attribute__((always_inline)) {
    if (row == 0) return -cB * V[c];
    const int m = row - 1;
    const double civ = ci[m];
    double f = -dg[m] * V[row * KC + c];
#pragma unroll
    for (int j = 0; j < NB; ++j) f = __builtin_fma((m >> j) & 1 ? civ : 1.0, V[(1 + (m ^ (1 << j))) * KC + c], f);
    if (m == 0) f = __builtin_fma(cC, V[c], f);
    return f;
  };
  auto err_norm = [&](const double* e, const double* ya, const double* yb) __attribute__((always_inline)) {
    auto mx = [](double p, double r) { return (p > r || p != p) ? p : r; };
    double m = 0.0;
    for (int k = tid; k < S * KC; k += nt) {
      const int c = k % KC;
      if (c == 0 || pcol(c) >= 0) m = mx(m, fabs(e[k]) / __builtin_fma(rtol, fmax(fabs(ya[k]), fabs(yb[k])), atol));
    }
    return wg_max(m, red);
  };

  // ---- S_ee^-1 in registers (as pk_rand_parity.hpp)
  double a[TS][TS], winvR = 1.0, qC = 0.0;
  auto factor = [&](const double q) __attribute__((always_inline)) {
    winvR = fast_rcp(__builtin_fma(q, cB, 1.0));
    qC = q * cC;
    if (tid < NALL && (__builtin_popcount(tid) & 1)) dio[tid] = fast_rcp(__builtin_fma(q, dg[tid], 1.0));
    __syncthreads();
    const double q2 = q * q;
    static_for<TS>([&](auto ic) {
      constexpr int ii = decltype(ic)::value;
      const int ma = emask(TS * bi + ii);
      static_for<TS>([&](auto jc) {
        constexpr int jj = decltype(jc)::value;
        const int mb = emask(TS * bj + jj), d = ma ^ mb;
        double v = 0.0;
        if (d == 0) {
          double s = 0.0;
#pragma unroll
          for (int j = 0; j < NB; ++j) { const int bit = 1 << j, c = ma ^ bit; s = __builtin_fma(wgt(ma, bit) * wgt(c, bit), dio[c], s); }
          v = __builtin_fma(-q2, s, __builtin_fma(q, dg[ma], 1.0));
        } else if (__builtin_popcount(d) == 2) {
          const int b1 = d & -d, b2 = d ^ b1;
          const int c1 = ma ^ b1, c2 = ma ^ b2;
          v = -q2 * __builtin_fma(wgt(ma, b1) * wgt(c1, b2), dio[c1], wgt(ma, b2) * wgt(c2, b1) * dio[c2]);
        }
        a[ii][jj] = v;
        __builtin_amdgcn_sched_barrier(0);
      });
    });
#pragma unroll 1
    for (int kb = 0; kb < TB; ++kb) {
      const bool prow = (bi == kb), pcl = (bj == kb);
      static_for<TS>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        constexpr int p = kk & 1;
        const int k = TS * kb + kk;
        double* rb = rowb + p * NM; double* cb = colb + p * NM;
        // pivot row / column published TRANSPOSED ([jj][bj]): conflict-free reads (pk_rand_parity.hpp has the measurement)
        if (prow) static_for<TS>([&](auto jc) { constexpr int jj = decltype(jc)::value; rb[jj * TB + bj] = a[kk][jj]; });
        if (pcl) static_for<TS>([&](auto ic) { constexpr int ii = decltype(ic)::value; cb[ii * TB + bi] = a[ii][kk]; });
        __syncthreads();
        const double rp = fast_rcp(rb[kk * TB + kb]);
        double rowv[TS], ml[TS];
        static_for<TS>([&](auto jc) { constexpr int jj = decltype(jc)::value; rowv[jj] = rb[jj * TB + bj]; });
        static_for<TS>([&](auto ic) { constexpr int ii = decltype(ic)::value; ml[ii] = cb[ii * TB + bi] * rp; });
        static_for<TS>([&](auto ic) {
          constexpr int ii = decltype(ic)::value;
          static_for<TS>([&](auto jc) { constexpr int jj = decltype(jc)::value; a[ii][jj] = __builtin_fma(-ml[ii], rowv[jj], a[ii][jj]); });
        });
        if (pcl) static_for<TS>([&](auto ic) { constexpr int ii = decltype(ic)::value; a[ii][kk] = -ml[ii]; });
        if (prow) static_for<TS>([&](auto jc) { constexpr int jj = decltype(jc)::value; a[kk][jj] = rowv[jj] * rp; });
        if (prow && pcl) a[kk][kk] = rp;
      });
    }
  };
  // dst <- M^-1 src for all KC columns ([S][KC] vectors, dst != src); three barrier phases
  auto solve = [&](const double* src, double* dst, const double q) __attribute__((always_inline)) {
    for (int it = tid; it < NM * KC; it += nt) {                  // r'_e = r_e - M_eo D_o^-1 r_o
      const int e = it / KC, c = it % KC;
      const int ma = emask(e);
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < NB; ++j) { const int bit = 1 << j, mc = ma ^ bit; s = __builtin_fma(wgt(ma, bit) * dio[mc], src[(1 + mc) * KC + c], s); }
      double r0 = src[(1 + ma) * KC + c];
      if (ma == 0) { const double zR = src[c] * winvR; dst[c] = zR; r0 = __builtin_fma(qC, zR, r0); }
      re[c * NM + (e % TS) * TB + e / TS] = __builtin_fma(q, s, r0);     // [column][jj][bj]: the product below reads 16 consecutive doubles
    }
    __syncthreads();
    static_for<KC>([&](auto cc) {                                 // x_e = S_ee^-1 r'_e, column by column
      constexpr int c = decltype(cc)::value;
      double r[TS];
      static_for<TS>([&](auto jc) { constexpr int jj = decltype(jc)::value; r[jj] = re[c * NM + jj * TB + bj]; });
      static_for<TS>([&](auto ic) {
        constexpr int ii = decltype(ic)::value;
        double v = a[ii][0] * r[0];
        static_for<TS - 1>([&](auto jc) { constexpr int jj = 1 + decltype(jc)::value; v = __builtin_fma(a[ii][jj], r[jj], v); });
        const double pr = gsum<TB>(v, lane);
        if (bj == ii) dst[(1 + emask(TS * bi + ii)) * KC + c] = pr;
      });
    });
    __syncthreads();
    for (int it = tid; it < NM * KC; it += nt) {                  // x_o = D_o^-1 (r_o - M_oe x_e)
      const int o = it / KC, c = it % KC;
      const int mc = omask(o);
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < NB; ++j) { const int bit = 1 << j; s = __builtin_fma(wgt(mc, bit), dst[(1 + (mc ^ bit)) * KC + c], s); }
      dst[(1 + mc) * KC + c] = dio[mc] * __builtin_fma(q, s, src[(1 + mc) * KC + c]);
    }
    __syncthreads();
  };
  // A'_c v at row `row` for parameter p, from the state column's value v, its neighbour sum nb = sum_{bit in m} v[m \ bit] and v_R
  auto src_of = [&](const int row, const int p, const double v, const double nb, const double vR) __attribute__((always_inline)) {
    if (row == 0) return (p == 1) ? -vR : 0.0;
    const int m = row - 1;
    int ddg, dci;
    dcoef(m, p, ddg, dci);
    double s = -(double)ddg * v;
    if (dci) s += nb;
    if (m == 0 && p == 2) s += vR;
    return s;
  };

  double tc = A.t[0];
  int k = 1;
  double te = A.t[1];
  double h;
  {
    for (int row = tid; row < S; row += nt) ZS[row * KC] = apply_A(Y, 0, row) + (row == 0 ? cA : 0.0);
    __syncthreads();
    auto mx = [](double p, double r) { return (p > r || p != p) ? p : r; };
    double m0 = 0.0, m1 = 0.0;
    for (int row = tid; row < S; row += nt) {
      const double sc = __builtin_fma(rtol, fabs(Y[row * KC]), atol);
      m0 = mx(m0, fabs(Y[row * KC]) / sc); m1 = mx(m1, fabs(ZS[row * KC]) / sc);
    }
    const double d0 = wg_max(m0, red), d1 = wg_max(m1, red);
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
    if (A.h0 > 0.0) h = A.h0;
    if (!(h > 0.0) || h != h) h = 1e-6;
  }
  const double* const kB = Tab::B;
  const double* const kE = Tab::E;
  bool after_reject = false;
  while (true) {
    if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; fail_from(k); break; }
    const bool last = (tc + 1.0001 * h >= te);
    const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
    if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; fail_from(k); break; }
    const double q = Tab::GAM * hs;
    factor(q);
    // round 0 input: h f(y) in the state column, zeros in the tangent columns; accumulators start at y
    for (int kk = tid; kk < S * KC; kk += nt) {
      const int row = kk / KC, c = kk % KC;
      ZS[kk] = (c == 0) ? hs * (apply_A(Y, 0, row) + (row == 0 ? cA : 0.0)) : 0.0;
      YN[kk] = Y[kk]; ER[kk] = 0.0;
    }
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it <= Tab::NS; ++it) {
      // round `it`: the state column forms stage it + 1, the tangent columns stage it
      solve(ZS, ZD, q);
      const double bB = it < Tab::NS ? kB[it] : 0.0, eB = it < Tab::NS ? kE[it] : 0.0;
      const double bT = it >= 1 ? kB[it - 1] : 0.0, eT = it >= 1 ? kE[it - 1] : 0.0;
      // one thread per row: accumulate this round's stage vectors, then turn the tangent entries into the NEXT round's right-hand sides
      for (int row = tid; row < S; row += nt) {
        const double z0 = ZD[row * KC], zR = ZD[0];
        YN[row * KC] = __builtin_fma(bB, z0, YN[row * KC]); ER[row * KC] = __builtin_fma(eB, z0, ER[row * KC]);
        double nbz = 0.0, nby = 0.0;
        if (row >= 1) {
          const int m = row - 1;
#pragma unroll
          for (int j = 0; j < NB; ++j) if ((m >> j) & 1) { nbz += ZD[(1 + (m ^ (1 << j))) * KC]; if (it == 0) nby += Y[(1 + (m ^ (1 << j))) * KC]; }
        }
        static_for<KT>([&](auto cc) {
          constexpr int c = 1 + decltype(cc)::value;
          const int p = pcol(c);
          const double zc = ZD[row * KC + c];
          YN[row * KC + c] = __builtin_fma(bT, zc, YN[row * KC + c]); ER[row * KC + c] = __builtin_fma(eT, zc, ER[row * KC + c]);
          double nx = 0.0;
          if (p >= 0 && it < Tab::NS) {
            if (it == 0) {                                                   // z'_1: h (A y' + b' + A' (y + g z_1))
              const double v = __builtin_fma(Tab::GAM, z0, Y[row * KC]), vR = __builtin_fma(Tab::GAM, zR, Y[0]);
              nx = hs * (apply_A(Y, c, row) + ((row == 0 && p == 0) ? 1.0 : 0.0) + src_of(row, p, v, __builtin_fma(Tab::GAM, nbz, nby), vR));
            } else {                                                         // z'_{it+1}: z'_it + q A' z_{it+1}
              nx = __builtin_fma(q, src_of(row, p, z0, nbz, zR), zc);
            }
          }
          ZD[row * KC + c] = nx;
        });
      }
      __syncthreads();
      { double* t_ = ZS; ZS = ZD; ZD = t_; }
    }
    const double err = err_norm(ER, Y, YN);
    if (err != err || err > 1e300) {
      ++nrej; after_reject = true; h = 0.1 * hs;
      double bad = 0.0;
      for (int row = tid; row < S; row += nt) if (nonfinite(Y[row * KC])) bad = 1.0;
      if (tid < NALL && (nonfinite(dg[tid]) || nonfinite(ci[tid]))) bad = 1.0;
      if (nonfinite(cA) || nonfinite(cB) || nonfinite(cC)) bad = 1.0;
      if (wg_max(bad, red) != 0.0) { status |= PK_ST_NONFINITE; fail_from(k); break; }
      continue;
    }
    double fac = root_q(err, Tab::Q) * (1.0 / 0.9);
    fac = fmax(1.0 / 6.0, fmin(5.0, fac));
    double hnew = hs / fac;
    if (err <= 1.0) {
      ++nacc;
      for (int kk = tid; kk < S * KC; kk += nt) Y[kk] = YN[kk];
      __syncthreads();
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

template <int NB, int TBP = 16>
static hipError_t launch_rand_sens_one(const SensArgs& a, hipStream_t st) {
  constexpr int KT = kRandSensKC - 1;
  const long long nch = (a.s.P + KT - 1) / KT, nblk = a.s.B * nch;
  if (nblk > 0x7fffffffLL) return hipErrorInvalidValue;
  if (a.s.status) {
    hipError_t e = hipMemsetAsync(a.s.status, 0, (size_t)a.s.B * sizeof(int32_t), st);
    if (e != hipSuccess) return e;
  }
  constexpr size_t lds = rand_sens_lds_bytes(NB);
  static_assert(lds <= 64 * 1024, "fits the default dynamic-LDS limit");
  hipLaunchKernelGGL((rand_sens_kernel<NB, TBP>), dim3((unsigned)nblk), dim3(TBP * TBP), lds, st, a);
  return hipGetLastError();
}

}  // namespace pk
