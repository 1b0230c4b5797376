// pk_rand_parity.hpp -- random model, n = 8 sites (257 states): rand_parity_kernel, EXACT resolvent solves, one workgroup per replica.
//
// Reference: models/randmod.py:122-247 (ode_system incl. the lowest-set-bit rate quirk at :201), solve_ode at :249-305 -- the reference's
// default model (config.toml:186), which has no size limit.
//
// The n-cube is bipartite: every transition (phosphorylation of one site, dephosphorylation of one site) changes the popcount PARITY of
// the bit mask, and degradation is diagonal.  So in M = I - q J the odd-odd block is DIAGONAL: the 128 odd-popcount states are eliminated
// exactly with 128 reciprocals, and what remains is a dense Schur complement on the 128 EVEN states,
//     S_ee[a][b] = delta_ab (1 + q dg_a)  -  sum over the odd neighbours c of both a and b of  M_ac (1 + q dg_c)^-1 M_cb ,
// (a == b: n terms; a and b two bits apart: two terms; else zero) -- exactly the size the n = 7 kernel (pk_rand_dense.hpp) inverts in the
// registers of one workgroup: an 8 x 8 block of S_ee^-1 per thread of a 16 x 16 grid, 128 Gauss-Jordan pivots per step with the pivot
// row / column published in LDS (one barrier per pivot), no pivoting (Schur complement of an M-matrix).  A solve M x = r is then
//     r'_e = r_e - M_eo D_o^-1 r_o   (eight neighbours per even state)          1 barrier
//     x_e  = S_ee^-1 r'_e            (64 FMAs per thread + a 16-lane DPP sum)   1 barrier
//     x_o  = D_o^-1 (r_o - M_oe x_e) (eight neighbours per odd state)           1 barrier
// -- three barrier phases, against 18 for the block elimination over popcount levels (pk_rand_level.hpp, kept as the independent twin the
// tests compare with) and ~130 per step for the approximate factorisation of pk_wide.hpp, whose W-method needs 500 .. 30 000 steps where
// this kernel takes the default LRP12's 25 .. 60 for EVERY parameter draw.  The mRNA row is decoupled and carried as a scalar.
// Even state e <-> mask 2 e + parity(e); odd state o <-> mask 2 o + 1 - parity(o)   (e, o = 0 .. 127).
// Same controller, landing rule, outputs, fused metric and flags as the other workgroup-per-replica kernels (WideOut of pk_wide.hpp).
#pragma once
#include "pk_wide.hpp"

namespace pk {

constexpr size_t rand_parity_lds_bytes(int n) {
  // y, yn, u6, two z buffers (2^n + 1 each); dg, ci, dio (2^n each); pivot row / column x 2 (2^(n-1) each); re (2^(n-1)); prevv; red
  const size_t NALL = (size_t)1 << n, NM = NALL / 2;
  return (5 * (NALL + 1) + 3 * NALL + 4 * NM + NM + (2 + n) + 24) * sizeof(double);
}

// TBP: side of the thread grid over the even Schur complement.  16 (256 threads: n = 7, 8) or 8 (64 threads = ONE wave per replica: n = 6,
// 4 x 4 blocks per lane, every workgroup barrier is a single wave's)
// TBJ: columns of the thread grid (default: square).  16 x 32 = 512 threads at n = 8: an 8 x 4 block per thread -- 64 VGPRs of matrix instead
// of 128, no AGPR traffic, and two workgroups fit a CU.
// A grid of fewer than 64 threads (4 x 4 at n = 5) shares its wave with 64 / (TBP TBJ) - 1 other replicas: one wave per workgroup, every
// per-replica quantity (LDS region, reductions, outputs) is the group's; the replicas of a wave diverge in their step loops like lanes do.
template <int NB, int TBP = 16, int TBJ = TBP>
__global__ __launch_bounds__(TBP * TBJ < 64 ? 64 : TBP * TBJ) void rand_parity_kernel(const SolveArgs A) {
  static_assert(NB >= 5 && NB <= 8 && (TBP == 16 || TBP == 8 || TBP == 4) && (TBJ == TBP || TBJ == 2 * TBP) && (1 << NB) <= 2 * TBP * TBJ,
                "thread grid over the even Schur complement (at least one thread per even state)");
  constexpr int NALL = 1 << NB;
  using Tab = ResolventTab<PK_METHOD_LRP12>;
  constexpr int NM = NALL / 2, TB = TBP, TS = NM / TB, TJ = TBJ, SJ = NM / TJ, NT = TB * TJ;   // NM even states: a TS x SJ block per thread
  constexpr int CPB = TS / SJ;                                                                // column blocks per row block (1 or 2)
  constexpr int RPB = NT < 64 ? 64 / NT : 1;                   // replicas per workgroup (> 1: one wave, aligned lane groups)
  extern __shared__ __align__(16) double lds_all[];
  const int tid = RPB > 1 ? (int)threadIdx.x % NT : (int)threadIdx.x, nt = NT;
  const int bi = tid / TJ, bj = tid % TJ, lane = threadIdx.x & 63;
  const int n = NB, S = A.S, T = A.T;
  const long long rep = RPB > 1 ? (long long)blockIdx.x * RPB + threadIdx.x / NT : (long long)blockIdx.x;
  if (rep >= A.B) return;
  double* lds = lds_all + (RPB > 1 ? (threadIdx.x / NT) * (rand_parity_lds_bytes(NB) / sizeof(double)) : 0);
  const double* __restrict__ th = A.theta + rep * A.P;
  double* y = lds;               double* yn = y + S;          double* u6 = yn + S;
  double* zs = u6 + S;           double* zd = zs + S;                                   // stage vector: source / destination of a solve
  double* dg = zd + S;           double* ci = dg + NALL;      double* dio = ci + NALL;  // loss rate, inflow rate, 1 / (1 + q dg) of the odd states: by mask
  double* rowb = dio + NALL;     double* colb = rowb + 2 * NM;                          // pivot row / column, double-buffered
  double* re = colb + 2 * NM;                                                           // right-hand side of the even system
  double* prevv = re + NM;       double* red = prevv + (2 + n);
  auto emask = [](const int e) __attribute__((always_inline)) { return 2 * e + (__builtin_popcount(e) & 1); };        // even popcount
  auto omask = [](const int o) __attribute__((always_inline)) { return 2 * o + 1 - (__builtin_popcount(o) & 1); };    // odd popcount
  // weight of the transition (x ^ bit) -> x in row x: phosphorylation into x at its inflow rate, dephosphorylation at unit rate
  auto wgt = [&](const int x, const int bit) __attribute__((always_inline)) { return (x & bit) ? ci[x] : 1.0; };
  const double* y0p = A.y0 + (A.y0_batched ? rep * S : 0);
  const double cA = th[0], cB = th[1], cC = th[2];

  for (int m = tid; m < NALL; m += nt) {
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
  for (int row = tid; row < S; row += nt) y[row] = y0p[row];
  __syncthreads();
  WideOut out = RPB > 1 ? WideOut(A, rep, y0p, prevv, red, tid, nt) : WideOut(A, rep, y0p, prevv, red);
  auto rmax = [&](double v) __attribute__((always_inline)) { if constexpr (RPB > 1) return grp_max(v, nt); else return wg_max(v, red); };
  out.emit(0, y, false);
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  if (T < 2) { out.finish(status, 0, 0); return; }
  const double rtol = A.rtol, atol = A.atol;

  // f(Y) * scale into dst (dst != Y); ends with a barrier
  auto rhs_into = [&](const double* Y, double* dst, const double scale) {
    for (int row = tid; row < S; row += nt) {
      double f;
      if (row == 0) f = __builtin_fma(-cB, Y[0], cA);
      else {
        const int m = row - 1;
        const double civ = ci[m];
        f = -dg[m] * Y[row];
#pragma unroll
        for (int j = 0; j < NB; ++j) f = __builtin_fma((m >> j) & 1 ? civ : 1.0, Y[1 + (m ^ (1 << j))], f);
        if (m == 0) f = __builtin_fma(cC, Y[0], f);
      }
      dst[row] = scale * f;
    }
    __syncthreads();
  };
  auto err_norm = [&](const double* e, const double* ya, const double* yb) {
    auto mx = [](double p, double r) { return (p > r || p != p) ? p : r; };
    double m = 0.0;
    for (int row = tid; row < S; row += nt) m = mx(m, fabs(e[row]) / __builtin_fma(rtol, fmax(fabs(ya[row]), fabs(yb[row])), atol));
    return rmax(m);
  };

  // ---- S_ee^-1 in registers: block (bi, bj) = even rows TS bi .. TS bi + TS - 1, even columns TS bj .. TS bj + TS - 1
  double a[TS][SJ], winvR = 1.0, qC = 0.0;
  auto factor = [&](const double q) {
    winvR = fast_rcp(__builtin_fma(q, cB, 1.0));
    qC = q * cC;
    {                                                            // the odd states' diagonal, inverted
      for (int c = tid; c < NALL; c += nt)
        if (__builtin_popcount(c) & 1) dio[c] = fast_rcp(__builtin_fma(q, dg[c], 1.0));
    }
    __syncthreads();
    const double q2 = q * q;
    static_for<TS>([&](auto ic) {
      constexpr int ii = decltype(ic)::value;
      const int ma = emask(TS * bi + ii);
      static_for<SJ>([&](auto jc) {
        constexpr int jj = decltype(jc)::value;
        const int mb = emask(SJ * bj + jj), d = ma ^ mb;
        double v = 0.0;
        if (d == 0) {
          double s = 0.0;
#pragma unroll
          for (int j = 0; j < NB; ++j) { const int bit = 1 << j, c = ma ^ bit; s = __builtin_fma(wgt(ma, bit) * wgt(c, bit), dio[c], s); }
          v = __builtin_fma(-q2, s, __builtin_fma(q, dg[ma], 1.0));
        } else if (__builtin_popcount(d) == 2) {
          const int bi_ = d & -d, bj_ = d ^ bi_;                 // the two bits a and b differ in
          const int c1 = ma ^ bi_, c2 = ma ^ bj_;                // their two common (odd) neighbours
          v = -q2 * __builtin_fma(wgt(ma, bi_) * wgt(c1, bj_), dio[c1], wgt(ma, bj_) * wgt(c2, bi_) * dio[c2]);
        }
        a[ii][jj] = v;
        // one entry at a time: left alone, the scheduler hoists the loads of all TS^2 entries (each with eight candidate terms) and the
        // kernel needs 512 registers + scratch; the fill runs once per step, its latency does not matter
        __builtin_amdgcn_sched_barrier(0);
      });
    });
#pragma unroll 1
    for (int kb = 0; kb < TB; ++kb) {
      const bool prow = (bi == kb);
      static_for<TS>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;                  // local row of the pivot
        constexpr int kc_ = kk % SJ;                             // its local column, in column block CPB kb + kk / SJ
        constexpr int p = kk & 1;
        const bool pcol = (bj == CPB * kb + kk / SJ);
        double* rb = rowb + p * NM; double* cb = colb + p * NM;
        // [r3] the published pivot row / column are stored TRANSPOSED (entry jj of block bj at [jj][bj]): the lanes of a wave then read 16
        // consecutive doubles per entry.  In block order ([bj][jj], 64 B between the blocks) the sixteen distinct addresses of a read fell
        // on two 16-byte bank groups -- 8-way conflicts on every one of the 8 + 8 reads of a pivot, 2 100 clocks per pivot for 450 of issue
        if (prow) static_for<SJ>([&](auto jc) { constexpr int jj = decltype(jc)::value; rb[jj * TJ + bj] = a[kk][jj]; });
        if (pcol) static_for<TS>([&](auto ic) { constexpr int ii = decltype(ic)::value; cb[ii * TB + bi] = a[ii][kc_]; });
        __syncthreads();
        const double rp = fast_rcp(rb[kc_ * TJ + CPB * kb + kk / SJ]);
        double rowv[SJ], ml[TS];
        static_for<SJ>([&](auto jc) { constexpr int jj = decltype(jc)::value; rowv[jj] = rb[jj * TJ + bj]; });
        static_for<TS>([&](auto ic) { constexpr int ii = decltype(ic)::value; ml[ii] = cb[ii * TB + bi] * rp; });
        static_for<TS>([&](auto ic) {
          constexpr int ii = decltype(ic)::value;
          static_for<SJ>([&](auto jc) { constexpr int jj = decltype(jc)::value; a[ii][jj] = __builtin_fma(-ml[ii], rowv[jj], a[ii][jj]); });
        });
        if (pcol) static_for<TS>([&](auto ic) { constexpr int ii = decltype(ic)::value; a[ii][kc_] = -ml[ii]; });               // pivot column: -a_ik / a_kk
        if (prow) static_for<SJ>([&](auto jc) { constexpr int jj = decltype(jc)::value; a[kk][jj] = rowv[jj] * rp; });          // pivot row: a_kj / a_kk
        if (prow && pcol) a[kk][kc_] = rp;                                                                                    // pivot: 1 / a_kk
      });
    }
  };
  // dst <- M^-1 src (mask order, row 0 = mRNA); ends with a barrier.  dst != src.
  auto solve = [&](const double* src, double* dst, const double q) __attribute__((always_inline)) {
    const double zR = src[0] * winvR;
    if (tid == 0) dst[0] = zR;
    if (tid < NM) {                                              // r'_e = r_e - M_eo D_o^-1 r_o
      const int ma = emask(tid);
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < NB; ++j) { const int bit = 1 << j, c = ma ^ bit; s = __builtin_fma(wgt(ma, bit) * dio[c], src[1 + c], s); }
      double r0 = src[1 + ma];
      if (ma == 0) r0 = __builtin_fma(qC, zR, r0);               // the -q C z_R coupling of the mask-0 row moved to the right-hand side
      re[(tid % SJ) * TJ + tid / SJ] = __builtin_fma(q, s, r0);      // transposed like the pivot buffers: conflict-free reads below
    }
    __syncthreads();
    double r[SJ], pr[TS];
    static_for<SJ>([&](auto jc) { constexpr int jj = decltype(jc)::value; r[jj] = re[jj * TJ + bj]; });
    static_for<TS>([&](auto ic) {
      constexpr int ii = decltype(ic)::value;
      double v = a[ii][0] * r[0];
      static_for<SJ - 1>([&](auto jc) { constexpr int jj = 1 + decltype(jc)::value; v = __builtin_fma(a[ii][jj], r[jj], v); });
      pr[ii] = gsum<TJ>(v, lane);
    });
    static_for<TS>([&](auto ic) { constexpr int ii = decltype(ic)::value; if (bj == ii) dst[1 + emask(TS * bi + ii)] = pr[ii]; });
    __syncthreads();
    if (tid < NM) {                                              // x_o = D_o^-1 (r_o - M_oe x_e)
      const int mc = omask(tid);
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < NB; ++j) { const int bit = 1 << j; s = __builtin_fma(wgt(mc, bit), dst[1 + (mc ^ bit)], s); }
      dst[1 + mc] = dio[mc] * __builtin_fma(q, s, src[1 + mc]);
    }
    __syncthreads();
  };

  double tc = A.t[0];
  int k = 1;
  double te = A.t[1];
  double h;
  {
    rhs_into(y, zs, 1.0);
    const double d0 = err_norm(y, y, y), d1 = err_norm(zs, y, y);
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
    if (A.h0 > 0.0) h = A.h0;
    if (!(h > 0.0) || h != h) h = 1e-6;
  }
  auto fail_from = [&](int kk) { for (; kk < T; ++kk) out.emit(kk, y, true); };
  bool after_reject = false;
  while (true) {
    if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; fail_from(k); break; }
    const bool last = (tc + 1.0001 * h >= te);
    const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
    if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; fail_from(k); break; }
    const double q = Tab::GAM * hs;
    factor(q);
    rhs_into(y, zs, hs);
    solve(zs, zd, q);
    { double* t_ = zs; zs = zd; zd = t_; }
    for (int row = tid; row < S; row += nt) { yn[row] = __builtin_fma(Tab::B[0], zs[row], y[row]); u6[row] = 0.0; }
#pragma unroll 1
    for (int kk = 1; kk < Tab::NS; ++kk) {
      solve(zs, zd, q);
      { double* t_ = zs; zs = zd; zd = t_; }
      const double bk = Tab::B[kk], ek = Tab::E[kk];
      for (int row = tid; row < S; row += nt) { yn[row] = __builtin_fma(bk, zs[row], yn[row]); u6[row] = __builtin_fma(ek, zs[row], u6[row]); }
    }
    const double err = err_norm(u6, y, yn);
    if (err != err || err > 1e300) {
      ++nrej; after_reject = true; h = 0.1 * hs;
      double bad = 0.0;
      for (int row = tid; row < S; row += nt) if (nonfinite(y[row])) bad = 1.0;
      for (int m = tid; m < NALL; m += nt) if (nonfinite(dg[m]) || nonfinite(ci[m])) bad = 1.0;
      if (nonfinite(cA) || nonfinite(cB) || nonfinite(cC)) bad = 1.0;
      if (rmax(bad) != 0.0) { status |= PK_ST_NONFINITE; fail_from(k); break; }
      continue;
    }
    double fac = root_q(err, Tab::Q) * (1.0 / 0.9);
    fac = fmax(1.0 / 6.0, fmin(5.0, fac));
    double hnew = hs / fac;
    if (err <= 1.0) {
      ++nacc;
      for (int row = tid; row < S; row += nt) y[row] = yn[row];
      __syncthreads();
      tc += hs;
      if (after_reject) hnew = fmin(hnew, hs);
      after_reject = false;
      if (last) {
        tc = te;
        out.emit(k, y, false);
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
  out.finish(status, nacc, nrej);
}

}  // namespace pk
