// pk_rand_dense.hpp -- random model, n = 7 sites (129 states): rand_dense_kernel<7>, one workgroup of 256 threads per replica.
//
// The approximate-factorisation kernels of pk_wide.hpp (Rosenbrock-W / additive Runge-Kutta on the n-cube) are bound by the latency of their
// ~100 level phases per step and need 400 ... 30 000 steps per replica: a parameter draw whose solution never comes to rest (mRNA
// degradation B ~ 0: linear growth over the whole time span) keeps the splitting error alive and takes 40x the steps of its neighbours
// (tools/gpu_wide_outlier.py: one such replica in 1 024 sets the time of the batch).  At n = 7 the exact route is open: the 128 x 128 matrix
// M = I - q J of the bit-mask block is 128 KB -- a quarter of a CU's register file.  So: the same LRP12 resolvent method as every kernel
// below 65 states (25 ... 60 steps, exact solves), with M^-1 held in REGISTERS, an 8 x 8 block per thread of a 16 x 16 thread grid:
//   factor   in-place Gauss-Jordan (no pivoting: M-matrix), 128 pivots; per pivot the owners publish the pivot row and column in LDS
//            (double-buffered: one barrier per pivot), every thread updates its block with 64 FMAs on 16 values read back
//   solve    x = M^-1 r: 64 FMAs per thread on the 8 entries of r its block columns need, then a DPP sum over the 16 lanes that share
//            the block's rows (one row of the thread grid = one 16-lane DPP row), one barrier
// The mRNA row is decoupled (lower triangular) and carried as a scalar by every thread, as in pk_tpr_rand.hpp.
// Same controller, landing rule, outputs, fused metric and flags as the other kernels (WideOut of pk_wide.hpp).
// Reference: models/randmod.py:122-247 (lowest-set-bit rate quirk at :201), solve_ode at :249-305.
#pragma once
#include "pk_wide.hpp"

namespace pk {

template <int NB> constexpr int rand_dense_threads() { return (1 << NB) / 8 * ((1 << NB) / 8); }
__host__ __device__ inline size_t rand_dense_lds_bytes(int n) {
  const size_t NM = (size_t)1 << n, S = NM + 1;
  return (5 * S + 2 * NM + 4 * NM + (2 + n) + 24) * sizeof(double);     // y, yn, u6, two z buffers; dg, ci; pivot row / column x 2; prevv; red
}

template <int NB>
__global__ __launch_bounds__(rand_dense_threads<NB>()) void rand_dense_kernel(const SolveArgs A) {
  using Tab = ResolventTab<PK_METHOD_LRP12>;
  constexpr int NM = 1 << NB, TB = NM / 8, NT = TB * TB;
  static_assert(TB == 16, "one row of the thread grid = one 16-lane DPP row");
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x, nt = NT;
  const int bi = tid / TB, bj = tid % TB, lane = tid & 63;
  const int n = NB, S = A.S, T = A.T;
  const long long rep = blockIdx.x;
  if (rep >= A.B) return;
  const double* __restrict__ th = A.theta + rep * A.P;
  double* y = lds;               double* yn = y + S;          double* u6 = yn + S;
  double* zs = u6 + S;           double* zd = zs + S;                                   // stage vector: source / destination of a solve
  double* dg = zd + S;           double* ci = dg + NM;
  double* rowb = ci + NM;        double* colb = rowb + 2 * NM;                          // pivot row / column, double-buffered
  double* prevv = colb + 2 * NM; double* red = prevv + (2 + n);
  const double* y0p = A.y0 + (A.y0_batched ? rep * S : 0);
  const double cA = th[0], cB = th[1], cC = th[2];

  if (tid < NM) {
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
  for (int row = tid; row < S; row += nt) y[row] = y0p[row];
  __syncthreads();
  WideOut out(A, rep, y0p, prevv, red);
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
    return wg_max(m, red);
  };

  // ---- M^-1 in registers: block (bi, bj) = rows 8 bi .. 8 bi + 7, columns 8 bj .. 8 bj + 7
  double a[8][8], winvR = 1.0, qC = 0.0;
  auto factor = [&](const double q) {
    winvR = fast_rcp(__builtin_fma(q, cB, 1.0));
    qC = q * cC;
    static_for<8>([&](auto ic) {
      constexpr int ii = decltype(ic)::value;
      const int m = 8 * bi + ii;
      const double dgm = dg[m], cim = ci[m];
      static_for<8>([&](auto jc) {
        constexpr int jj = decltype(jc)::value;
        const int c = 8 * bj + jj, d = m ^ c;
        double v;
        if (d == 0) v = __builtin_fma(q, dgm, 1.0);
        else if ((d & (d - 1)) == 0) v = -q * ((m & d) ? cim : 1.0);
        else v = 0.0;
        a[ii][jj] = v;
      });
    });
#pragma unroll 1
    for (int kb = 0; kb < TB; ++kb) {
      const bool prow = (bi == kb), pcol = (bj == kb);
      static_for<8>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        constexpr int p = kk & 1;
        const int k = 8 * kb + kk;
        double* rb = rowb + p * NM; double* cb = colb + p * NM;
        if (prow) static_for<8>([&](auto jc) { constexpr int jj = decltype(jc)::value; rb[jj * 16 + bj] = a[kk][jj]; });
        if (pcol) static_for<8>([&](auto ic) { constexpr int ii = decltype(ic)::value; cb[ii * 16 + bi] = a[ii][kk]; });
        __syncthreads();
        const double rp = fast_rcp(rb[kk * 16 + kb]);                    // [r3] buffers transposed ([jj][bj]): conflict-free reads (pk_rand_parity.hpp)
        double rowv[8], ml[8];
        static_for<8>([&](auto jc) { constexpr int jj = decltype(jc)::value; rowv[jj] = rb[jj * 16 + bj]; });
        static_for<8>([&](auto ic) { constexpr int ii = decltype(ic)::value; ml[ii] = cb[ii * 16 + bi] * rp; });
        static_for<8>([&](auto ic) {
          constexpr int ii = decltype(ic)::value;
          static_for<8>([&](auto jc) { constexpr int jj = decltype(jc)::value; a[ii][jj] = __builtin_fma(-ml[ii], rowv[jj], a[ii][jj]); });
        });
        if (pcol) static_for<8>([&](auto ic) { constexpr int ii = decltype(ic)::value; a[ii][kk] = -ml[ii]; });                 // pivot column: -a_ik / a_kk
        if (prow) static_for<8>([&](auto jc) { constexpr int jj = decltype(jc)::value; a[kk][jj] = rowv[jj] * rp; });          // pivot row: a_kj / a_kk
        if (prow && pcol) a[kk][kk] = rp;                                                                                     // pivot: 1 / a_kk
      });
    }
  };
  // dst <- M^-1 src; ends with a barrier
  auto solve = [&](const double* src, double* dst) {
    const double zR = src[0] * winvR;
    double r[8], pr[8];
    static_for<8>([&](auto jc) { constexpr int jj = decltype(jc)::value; r[jj] = src[1 + 8 * bj + jj]; });
    if (bj == 0) r[0] = __builtin_fma(qC, zR, r[0]);          // the -q C z_R coupling of the mask-0 row moved to the right-hand side
    static_for<8>([&](auto ic) {
      constexpr int ii = decltype(ic)::value;
      double v = a[ii][0] * r[0];
      static_for<7>([&](auto jc) { constexpr int jj = 1 + decltype(jc)::value; v = __builtin_fma(a[ii][jj], r[jj], v); });
      pr[ii] = gsum<TB>(v, lane);
    });
    static_for<8>([&](auto ic) { constexpr int ii = decltype(ic)::value; if (bj == ii) dst[1 + 8 * bi + ii] = pr[ii]; });
    if (tid == 0) dst[0] = zR;
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
    factor(Tab::GAM * hs);
    rhs_into(y, zs, hs);
    solve(zs, zd);
    { double* t_ = zs; zs = zd; zd = t_; }
    for (int row = tid; row < S; row += nt) { yn[row] = __builtin_fma(Tab::B[0], zs[row], y[row]); u6[row] = 0.0; }
#pragma unroll 1
    for (int kk = 1; kk < Tab::NS; ++kk) {
      solve(zs, zd);
      { double* t_ = zs; zs = zd; zd = t_; }
      const double bk = Tab::B[kk], ek = Tab::E[kk];
      for (int row = tid; row < S; row += nt) { yn[row] = __builtin_fma(bk, zs[row], yn[row]); u6[row] = __builtin_fma(ek, zs[row], u6[row]); }
    }
    const double err = err_norm(u6, y, yn);
    if (err != err || err > 1e300) {
      ++nrej; after_reject = true; h = 0.1 * hs;
      double bad = 0.0;
      for (int row = tid; row < S; row += nt) if (nonfinite(y[row])) bad = 1.0;
      if (tid < NM && (nonfinite(dg[tid]) || nonfinite(ci[tid]))) bad = 1.0;
      if (nonfinite(cA) || nonfinite(cB) || nonfinite(cC)) bad = 1.0;
      if (wg_max(bad, red) != 0.0) { status |= PK_ST_NONFINITE; fail_from(k); break; }
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
