// pk_wide.hpp -- per-protein systems that do not fit one wavefront's lane groups: ONE WORKGROUP PER REPLICA, state and stage
// vectors in LDS (or, for the largest random-model systems, in an HBM scratch row), rows strided over the threads.
//
//   wide_chain_kernel<MODEL>   distmod with n_sites > 64 and succmod with n_sites > 62 (models/distmod.py:7-65, succmod.py:9-90): the
//                              default integrator LRP12 (resolvent form, exact linear solves): arrow elimination with one workgroup
//                              reduction per solve (DIST) / parallel cyclic reduction of the tridiagonal system in LDS (SUCC).
//   wide_rand_kernel<LDSV>     randmod with n_sites >= 7 (models/randmod.py:122-247): 2^n bit-mask states on the n-cube, Jacobian
//                              -diag(loss) + F + K with F (phosphorylation, mask -> mask | bit) strictly lower and K (dephosphorylation,
//                              unit rate) strictly upper triangular in mask order.  No exact sparse factorisation of g I - J exists
//                              short of a dense one, so this kernel integrates with the Rosenbrock-W method ROS34PW2 (order 3 for ANY
//                              Jacobian approximation; tables: pk_network_solve.hpp, tools/check_ros34pw2.py) and the approximate
//                              factorisation  g I - J ~= (D_g - F) D_g^-1 (D_g - K),  D_g = g I + diag(loss):  two sweeps over the
//                              cube, level by level (popcount order: all masks of one level are independent).  A numpy model of exactly
//                              this scheme takes the same number of steps as ROS34PW2 with the exact Jacobian (tools/proto_rand_wide.py).
//                              Default since the second half of round 2: the order-4 additive method ARK4(3)6L[2]SA on the same approximate
//                              factorisation (pk_network_solve_ark.hpp explains the construction): 2-3x fewer steps, worst band error on the
//                              reference fixtures 0.13 instead of 0.36, one-theta latency at n = 7 29 ms instead of 51 ms
//                              (PK_WIDE_RAND_ROSW=1 selects the order-3 method).  Being of low order they run at 0.05 x (order 3) / 0.02 x
//                              (order 4) the caller's tolerances so that the library's default options (tuned for the order-11 LRP12) keep
//                              their margin inside the parity band.
//
// Outputs (sol / flat / fused Morris metric / status / n_steps) have exactly the semantics of Emitter in pk_solve_kernel.hpp.
#pragma once
#include "pk_solve_kernel.hpp"

namespace pk {

namespace rosw_tab {     // ROS34PW2 in implementation form (same numbers as namespace rosw of pk_network_solve.hpp)
constexpr double GAM = 0.435866521508459;
__device__ constexpr double TA[4][3] = {{0, 0, 0}, {2.0, 0, 0}, {1.41921731745576465, -0.25923221167296971378, 0},
                                        {4.1847604823191607312, -0.2851920173554959137, 2.2942803602790417167}};
__device__ constexpr double TC[4][3] = {{0, 0, 0}, {-4.5885607205580834861, 0, 0}, {-4.1847604823191607312, 0.2851920173554959137, 0},
                                        {-6.3681792001283577635, -6.7956209444668361844, 2.8700986043310560892}};
__device__ constexpr double TE[4] = {0.27774994764796811038, -1.4032398951759990242, 1.7726301276675507452, 0.5};
}  // namespace rosw_tab

// ---- workgroup reductions (every thread of the block must call; `red` holds >= 16 doubles of LDS)
__device__ __forceinline__ double wg_max(double v, double* red) {
  auto mx = [](double a, double b) { return (a > b || a != a) ? a : b; };      // NaN-propagating
  for (int off = 32; off > 0; off >>= 1) v = mx(v, __shfl_xor(v, off));
  const int nw = (blockDim.x + 63) >> 6;
  if (nw == 1) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = red[0];
  for (int i = 1; i < nw; ++i) r = mx(r, red[i]);
  return r;
}
__device__ __forceinline__ double wg_sum(double v, double* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int nw = (blockDim.x + 63) >> 6;
  if (nw == 1) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = red[0];
  for (int i = 1; i < nw; ++i) r += red[i];
  return r;
}

// sums / maxima over an aligned group of gnt < 64 lanes of one wave (several replicas per wave: pk_rand_parity.hpp at n = 5)
__device__ __forceinline__ double grp_sum(double v, const int gnt) { for (int off = gnt >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off); return v; }
__device__ __forceinline__ double grp_max(double v, const int gnt) {
  auto mx = [](double a, double b) { return (a > b || a != a) ? a : b; };
  for (int off = gnt >> 1; off > 0; off >>= 1) v = mx(v, __shfl_xor(v, off));
  return v;
}

// ---- outputs of one replica owned by a whole workgroup (or, gnt < 64, by an aligned group of gnt lanes: gtid = lane within the group)
struct WideOut {
  const SolveArgs& A; long long rep; const double* y0p; double* prevv; double* red;     // prevv: LDS [2 + n_obs]
  double s1 = 0.0, s2 = 0.0, dyn = 0.0, shift = 0.0;
  int gtid, gnt; bool grp;
  __device__ __forceinline__ WideOut(const SolveArgs& a, long long r, const double* y0, double* pv, double* rd) : A(a), rep(r), y0p(y0), prevv(pv), red(rd),
      gtid(threadIdx.x), gnt(blockDim.x), grp(false) {}
  __device__ __forceinline__ WideOut(const SolveArgs& a, long long r, const double* y0, double* pv, double* rd, int gt, int gn) : A(a), rep(r), y0p(y0), prevv(pv),
      red(rd), gtid(gt), gnt(gn), grp(true) {}
  __device__ __forceinline__ double rsum(double v) { return grp ? grp_sum(v, gnt) : wg_sum(v, red); }
  __device__ __forceinline__ double val(double x, int row, bool nan_fill) const {
    if (nan_fill) return __builtin_nan("");
    double v = A.clip ? ((x < 0.0) ? 0.0 : x) : x;
    if (A.normalize) v *= 1.0 / y0p[row];
    return v;
  }
  // y: vector of S state values (LDS or global); every thread of the block calls
  __device__ __forceinline__ void emit(int k, const double* y, bool nan_fill) {
    const int tid = gtid, nt = gnt, S = A.S, T = A.T, nobs = 2 + A.n_obs;
    const int T5 = T > 5 ? T - 5 : 0;
    double* solp = A.sol ? A.sol + (rep * T + k) * S : nullptr;
    double* fl = A.flat ? A.flat + rep * A.F : nullptr;
    if (A.metric && k == 0) {
      double loc = 0.0;
      for (int row = tid; row < nobs; row += nt) loc += val(y[row], row, nan_fill);
      shift = rsum(loc) / nobs;
    }
    for (int row = tid; row < S; row += nt) {
      const double v = val(y[row], row, nan_fill);
      if (solp) solp[row] = v;
      if (fl) {
        if (row == 0) { if (k >= 5) fl[k - 5] = v; }
        else if (row == 1) fl[T5 + k] = v;
        else if (row < nobs) fl[T5 + T + (row - 2) * T + k] = v;
      }
      if (A.metric && row < nobs) {
        const double pv = (k == 0) ? v : prevv[row];
        const double xs = v - shift, d = v - pv;
        s1 += v; s2 = __builtin_fma(xs, xs, s2); dyn = __builtin_fma(d, d, dyn);
        prevv[row] = v;
      }
    }
  }
  __device__ __forceinline__ void finish(int status, int acc, int rej) {
    if (A.metric) {
      const double L = 2.0 * A.T + (double)A.T * A.n_obs;
      const double tot = rsum(s1);
      double m;
      switch (A.metric_id) {
        case PK_METRIC_TOTAL_SIGNAL: m = tot; break;
        case PK_METRIC_MEAN_ACTIVITY: m = tot / L; break;
        case PK_METRIC_VARIANCE: { const double q = rsum(s2); const double ms = tot / L - shift; m = q / L - ms * ms; } break;
        case PK_METRIC_DYNAMICS: m = rsum(dyn); break;
        default: { const double q = rsum(s2); m = sqrt(fmax(q + 2.0 * shift * tot - L * shift * shift, 0.0)); } break;
      }
      if (gtid == 0) A.metric[rep] = m;
    }
    if (gtid == 0) {
      if (A.status) A.status[rep] = status;
      if (A.n_steps) { A.n_steps[2 * rep] = acc; A.n_steps[2 * rep + 1] = rej; }
    }
  }
};

// =====================================================================================================================================
// distributive / successive model, any n that fits LDS: LRP12 in resolvent form with exact solves.
//   LDS doubles: 15 vectors of S + (2 + n) + 24   (=> n_sites <= 1276 in 160 KB)
constexpr int wide_chain_vectors = 15;
__host__ __device__ inline size_t wide_chain_lds_bytes(int S, int n) { return ((size_t)wide_chain_vectors * S + (2 + n) + 24) * sizeof(double); }

template <int MODEL>
__global__ __launch_bounds__(256) void wide_chain_kernel(const SolveArgs A) {
  static_assert(MODEL == M_DIST || MODEL == M_SUCC, "chain kernel: distributive / successive");
  using Tab = ResolventTab<PK_METHOD_LRP12>;
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int n = A.n_sites, S = A.S, T = A.T;
  const long long rep = blockIdx.x;
  if (rep >= A.B) return;
  const double* __restrict__ th = A.theta + rep * A.P;
  double* y = lds;            double* yn = y + S;        double* z = yn + S;       double* u6 = z + S;
  double* c1 = u6 + S;        double* dgv = c1 + S;      double* c2 = dgv + S;                    // row coefficients (RowCoef of pk_models.hpp)
  double* pa = c2 + S;        double* pb = pa + S;       double* pc = pb + S;      double* pr = pc + S;      // PCR buffers 0 | arrow: winv, cw in pa, pb
  double* qa = pr + S;        double* qb = qa + S;       double* qc = qb + S;      double* qr = qc + S;      // PCR buffers 1
  double* prevv = qr + S;     double* red = prevv + (2 + n);
  const double* y0p = A.y0 + (A.y0_batched ? rep * S : 0);

  for (int row = tid; row < S; row += nt) {
    const RowCoef c = load_row<MODEL>(th, n, S, row);
    c1[row] = c.c1; dgv[row] = c.dg; c2[row] = c.c2;
    y[row] = y0p[row];
  }
  const double cA = th[0];
  __syncthreads();
  WideOut out(A, rep, y0p, prevv, red);
  out.emit(0, y, false);
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  if (T < 2) { out.finish(status, 0, 0); return; }
  const double rtol = A.rtol, atol = A.atol;

  // f(Y) into dst (dst != Y); all threads; ends with a barrier
  auto rhs_into = [&](const double* Y, double* dst, const double scale) {
    if constexpr (MODEL == M_DIST) {
      double loc = 0.0;
      for (int row = 2 + tid; row < S; row += nt) loc += Y[row];
      const double sites = wg_sum(loc, red);
      const double R = Y[0], P = Y[1];
      for (int row = tid; row < S; row += nt) {
        double f;
        if (row == 0) f = __builtin_fma(dgv[0], R, cA);
        else if (row == 1) f = __builtin_fma(c2[1], R, __builtin_fma(dgv[1], P, sites));
        else f = __builtin_fma(c1[row], P, dgv[row] * Y[row]);
        dst[row] = scale * f;
      }
    } else {
      for (int row = tid; row < S; row += nt) {
        double f = __builtin_fma(dgv[row], Y[row], row == 0 ? cA : 0.0);
        if (row >= 1) f = __builtin_fma(c1[row], Y[row - 1], f);
        if (row + 1 < S) f = __builtin_fma(c2[row], Y[row + 1], f);
        dst[row] = scale * f;
      }
    }
    __syncthreads();
  };
  auto err_norm = [&](const double* e, const double* ya, const double* yb) {
    auto mx = [](double p, double r) { return (p > r || p != p) ? p : r; };
    double m = 0.0;
    for (int row = tid; row < S; row += nt) m = mx(m, fabs(e[row]) / __builtin_fma(rtol, fmax(fabs(ya[row]), fabs(yb[row])), atol));
    return wg_max(m, red);
  };

  // ---- linear solves with M = I - q J
  double qq = 0.0, sinv = 0.0, Scw = 0.0;
  auto factor = [&](const double q) {
    qq = q;
    if constexpr (MODEL == M_DIST) {
      double loc = 0.0;
      for (int row = tid; row < S; row += nt) {
        const double w = 1.0 / __builtin_fma(-q, dgv[row], 1.0);
        pa[row] = w;                                               // winv
        const double cw = (row >= 2) ? q * c1[row] * w : 0.0;      // q S_i / (1 + q d_i)
        pb[row] = cw;
        loc += cw;
      }
      Scw = wg_sum(loc, red);
      sinv = 1.0 / __builtin_fma(-q, dgv[1] + Scw, 1.0);            // 1 + q (Dsum - sum_i cw_i): Schur pivot of the P row
      __syncthreads();
    }
  };
  // z <- M^-1 z, in place; all threads; ends with a barrier
  auto solve = [&]() {
    if constexpr (MODEL == M_DIST) {
      double loc = 0.0;
      for (int row = 2 + tid; row < S; row += nt) { const double t = z[row] * pa[row]; z[row] = t; loc += t; }
      const double St = wg_sum(loc, red);
      const double xR = z[0] * pa[0];
      const double xP = __builtin_fma(qq, __builtin_fma(c2[1], xR, St), z[1]) * sinv;
      __syncthreads();                                            // everyone has read z[0], z[1]
      for (int row = tid; row < S; row += nt) {
        if (row == 0) z[0] = xR;
        else if (row == 1) z[1] = xP;
        else z[row] = __builtin_fma(pb[row], xP, z[row]);
      }
      __syncthreads();
    } else {
      // parallel cyclic reduction on (a, b, c, r): a_i = -q c1_i, b_i = 1 - q dg_i, c_i = -q c2_i
      double *sa = pa, *sb = pb, *sc = pc, *sr = pr, *da = qa, *db = qb, *dc = qc, *dr = qr;
      for (int row = tid; row < S; row += nt) { sa[row] = -qq * c1[row]; sb[row] = __builtin_fma(-qq, dgv[row], 1.0); sc[row] = -qq * c2[row]; sr[row] = z[row]; }
      __syncthreads();
      for (int d = 1; d < S; d <<= 1) {
        for (int row = tid; row < S; row += nt) {
          const int lo = row - d, hi = row + d;
          double na = 0.0, nc = 0.0, nb = sb[row], nr = sr[row];
          if (lo >= 0) { const double al = -sa[row] / sb[lo]; na = al * sa[lo]; nb = __builtin_fma(al, sc[lo], nb); nr = __builtin_fma(al, sr[lo], nr); }
          if (hi < S) { const double ga = -sc[row] / sb[hi]; nc = ga * sc[hi]; nb = __builtin_fma(ga, sa[hi], nb); nr = __builtin_fma(ga, sr[hi], nr); }
          da[row] = na; db[row] = nb; dc[row] = nc; dr[row] = nr;
        }
        __syncthreads();
        double* t;
        t = sa; sa = da; da = t; t = sb; sb = db; db = t; t = sc; sc = dc; dc = t; t = sr; sr = dr; dr = t;
      }
      for (int row = tid; row < S; row += nt) z[row] = sr[row] / sb[row];
      __syncthreads();
    }
  };

  double tc = A.t[0];
  int k = 1;
  double te = A.t[1];
  double h;
  {
    rhs_into(y, z, 1.0);
    const double d0 = err_norm(y, y, y), d1 = err_norm(z, y, y);
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
    rhs_into(y, z, hs);
    solve();
    for (int row = tid; row < S; row += nt) { yn[row] = __builtin_fma(Tab::B[0], z[row], y[row]); u6[row] = 0.0; }
    __syncthreads();
    for (int kk = 1; kk < Tab::NS; ++kk) {
      solve();
      const double bk = Tab::B[kk], ek = Tab::E[kk];
      for (int row = tid; row < S; row += nt) { yn[row] = __builtin_fma(bk, z[row], yn[row]); u6[row] = __builtin_fma(ek, z[row], u6[row]); }
      __syncthreads();
    }
    const double err = err_norm(u6, y, yn);
    if (err != err || err > 1e300) {
      ++nrej; after_reject = true; h = 0.1 * hs;
      double bad = 0.0;
      for (int row = tid; row < S; row += nt) if (nonfinite(y[row]) || nonfinite(c1[row]) || nonfinite(dgv[row])) bad = 1.0;
      if (nonfinite(cA)) bad = 1.0;
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

// =====================================================================================================================================
// random model on the n-cube.  Vectors of S = 2^n + 1 doubles: y, Ys, U0..U3, f, loss (index 1 + mask), dinv : 9 vectors
// ROS34PW2: y, Ys, U0..U3, f, loss, dinv = 9 vectors; ARK436: y, Y, w, v, g r, R3..R6, the two running sums, loss, dinv = 13 vectors
__host__ __device__ constexpr int wide_rand_vectors(bool ark) { return ark ? 16 : 9; }        // ARK: + s, u, p(t) of the drift removal
__host__ __device__ inline size_t wide_rand_small_doubles(int n) { return (size_t)n + (2 + n) + 24; }                 // Sr, prevv, red
__host__ __device__ inline size_t wide_rand_lds_bytes(int n, bool ldsv, bool ark) {                                    // + lvl, binomials [, ord]
  const size_t NM = (size_t)1 << n, S = NM + 1;
  return ((ldsv ? (size_t)wide_rand_vectors(ark) * S : 0) + wide_rand_small_doubles(n)) * sizeof(double) + ((n + 2) + 21 * 21 + (ldsv ? NM : 0)) * sizeof(int);
}
// doubles of HBM scratch per replica when the vectors do not fit LDS (the vectors + the level-order table)
__host__ __device__ inline size_t wide_rand_scratch_doubles(int n, bool ark) { const size_t NM = (size_t)1 << n, S = NM + 1; return wide_rand_vectors(ark) * S + (NM + 1) / 2 + 1; }

template <bool LDSV, bool ARK>
__global__ __launch_bounds__(256) void wide_rand_kernel(const SolveArgs A, double* __restrict__ scratch, const size_t stride, const int drift_on) {
  using namespace rosw_tab;
  constexpr int NV = wide_rand_vectors(ARK);
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int n = A.n_sites, NM = 1 << n, S = A.S, T = A.T;
  const long long rep = blockIdx.x;
  if (rep >= A.B) return;
  const double* __restrict__ th = A.theta + rep * A.P;
  double* small = LDSV ? lds + (size_t)NV * S : lds;
  double* vec = LDSV ? lds : scratch + (size_t)rep * stride;
  double* y = vec;            double* Ys = y + S;
  double* U[4] = {Ys + S, Ys + 2 * (size_t)S, Ys + 3 * (size_t)S, Ys + 4 * (size_t)S};
  double* f = Ys + 5 * (size_t)S;     double* loss = f + S;       double* dinv = loss + S;
  double* extra = dinv + S;                          // ARK only: four more vectors
  double* Sr = small;         double* prevv = Sr + n;     double* red = prevv + (2 + n);
  int* lvl = reinterpret_cast<int*>(red + 24);       int* binom = lvl + (n + 2);
  int* ord = LDSV ? binom + 21 * 21 : reinterpret_cast<int*>(vec + (size_t)NV * S);      // masks in popcount-level order
  const double* y0p = A.y0 + (A.y0_batched ? rep * S : 0);
  const double cA = th[0], cB = th[1], cC = th[2], cD = th[3];
  double cAe = cA;                                    // the constant of the mRNA row as the integrator sees it (0 once the drift is removed)

  // ---- tables: binomials, popcount levels, masks in level order
  if (tid == 0) {
    for (int a = 0; a <= 20; ++a) for (int b = 0; b <= 20; ++b) binom[a * 21 + b] = (b == 0) ? 1 : (a == 0 ? 0 : binom[(a - 1) * 21 + b - 1] + binom[(a - 1) * 21 + b]);
    lvl[0] = 0;
    for (int L = 0; L <= n; ++L) lvl[L + 1] = lvl[L] + binom[n * 21 + L];
  }
  for (int j = tid; j < n; j += nt) Sr[j] = th[4 + j];
  __syncthreads();
  double sumS = 0.0;
  for (int j = 0; j < n; ++j) sumS += Sr[j];                           // same order as the reference (randmod.py:145-151)
  for (int m = tid; m < NM; m += nt) {
    int kk = __popc(m), rank = 0;
    { int c = kk; for (int j = n - 1; j >= 0 && c > 0; --j) if ((m >> j) & 1) { rank += binom[j * 21 + c]; --c; } }
    ord[lvl[kk] + rank] = m;
    double l;
    if (m == 0) l = cD + sumS;
    else {
      const int lsb = __builtin_ctz(m);
      double o = 0.0;
      for (int j = 0; j < n; ++j) o += ((m >> j) & 1) ? 1.0 : Sr[j < lsb ? j : lsb];     // same accumulation order as load_row<M_RAND>
      l = o + th[4 + n + m - 1];
    }
    loss[1 + m] = l;
  }
  if (tid == 0) loss[0] = cB;
  for (int row = tid; row < S; row += nt) y[row] = y0p[row];
  __syncthreads();
  WideOut out(A, rep, y0p, prevv, red);
  out.emit(0, y, false);
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  if (T < 2) { out.finish(status, 0, 0); return; }
  // order-3 method at 0.05 x, order-4 additive method at 0.02 x the caller's tolerances: see the header of this file
  const double rtol = (ARK ? 0.02 : 0.05) * A.rtol, atol = (ARK ? 0.02 : 0.05) * A.atol;

  // dst = f(Y) + sum_u coef[u] * U[u]   (dst != Y); ends with a barrier
  auto rhs_into = [&](const double* Y, double* dst, const int nu, const double* coef) {
    for (int row = tid; row < S; row += nt) {
      double v;
      if (row == 0) v = __builtin_fma(-cB, Y[0], cAe);
      else {
        const int m = row - 1;
        double lo = 0.0, hi = 0.0;
        // (a fixed-trip-count loop over all n neighbours with a select measured 15 % slower than these two set-bit walks)
        for (int mm = m; mm; mm &= mm - 1) lo += Y[1 + (m ^ (mm & -mm))];
        for (int mm = ~m & (NM - 1); mm; mm &= mm - 1) hi += Y[1 + (m | (mm & -mm))];
        v = __builtin_fma(-loss[row], Y[row], hi);
        v = (m == 0) ? __builtin_fma(cC, Y[0], v) : __builtin_fma(Sr[__builtin_ctz(m)], lo, v);
      }
      for (int u = 0; u < nu; ++u) v = __builtin_fma(coef[u], U[u][row], v);
      dst[row] = v;
    }
    __syncthreads();
  };
  // x = W~^-1 r   (x != r): forward sweep over ascending levels, backward over descending; ends with a barrier
  auto solve = [&](const double* r, double* x) {
    if (tid == 0) {
      const double xR = r[0] * dinv[0];
      x[0] = xR;
      x[1] = __builtin_fma(cC, xR, r[1]) * dinv[1];
    }
    __syncthreads();
    for (int L = 1; L <= n; ++L) {
      for (int idx = lvl[L] + tid; idx < lvl[L + 1]; idx += nt) {
        const int m = ord[idx];
        double lo = 0.0;
        for (int mm = m; mm; mm &= mm - 1) lo += x[1 + (m ^ (mm & -mm))];
        x[1 + m] = __builtin_fma(Sr[__builtin_ctz(m)], lo, r[1 + m]) * dinv[1 + m];
      }
      __syncthreads();
    }
    for (int L = n - 1; L >= 0; --L) {
      for (int idx = lvl[L] + tid; idx < lvl[L + 1]; idx += nt) {
        const int m = ord[idx];
        double hi = 0.0;
        for (int mm = ~m & (NM - 1); mm; mm &= mm - 1) hi += x[1 + (m | (mm & -mm))];
        x[1 + m] = __builtin_fma(hi, dinv[1 + m], x[1 + m]);
      }
      __syncthreads();
    }
  };
  const double* pofs = nullptr;                       // drift removal: the integrated vector is y - p(t); tolerances stay relative to y
  auto norm_of = [&](const double* e, const double* ya, const double* yb) {
    auto mx = [](double p, double r) { return (p > r || p != p) ? p : r; };
    double m = 0.0;
    for (int row = tid; row < S; row += nt) {
      const double po = pofs ? pofs[row] : 0.0;
      m = mx(m, fabs(e[row]) / __builtin_fma(rtol, fmax(fabs(ya[row] + po), fabs(yb[row] + po)), atol));
    }
    return wg_max(m, red);
  };

  double tc = A.t[0];
  int k = 1;
  double te = A.t[1];
  const double t0 = tc;
  // ---- drift removal (ARK only).  The splitting error of the approximate factorisation is proportional to the increments, so a replica
  // whose solution never comes to rest -- mRNA degradation B ~ 0: R(t) and with it every state keep moving over the whole time span -- pays
  // 10-40x the steps of its neighbours (tools/gpu_wide_outlier.py).  That motion is known in closed form.  With phi(tau) = (1 - e^{-B tau}) / B,
  //   R(tau) = R0 + (A - B R0) phi,      cube: y_c = s - u phi + w,     (M_c + B I) u = C (A - B R0) e_0,    M_c s = -C R0 e_0 - u,
  // the remainder w obeys the HOMOGENEOUS w' = M_c w and comes to rest at the cube's own rates (the form is cancellation-free for B -> 0,
  // where phi -> tau).  The two linear systems are solved by the sweeps this kernel already has, iterated to 1e-14 (defect correction:
  // a regular splitting of an M-matrix); if an iteration does not converge the replica is integrated as it stands.
  bool drift = false;
  [[maybe_unused]] double* sv = nullptr; [[maybe_unused]] double* uv = nullptr; [[maybe_unused]] double* pv = nullptr;
  const double R0 = y0p[0];
  auto phi_of = [&](const double tau) {
    const double x = cB * tau;
    return (x < 1e-5) ? tau * (1.0 - 0.5 * x + x * x * (1.0 / 6.0)) : -expm1(-x) / cB;
  };
  auto set_p = [&](const double tau) {                  // p(t) into pv; ends with a barrier
    const double ph = phi_of(tau);
    for (int row = tid; row < S; row += nt) pv[row] = (row == 0) ? __builtin_fma(cA - cB * R0, ph, R0) : __builtin_fma(-uv[row], ph, sv[row]);
    __syncthreads();
  };
  if constexpr (ARK) {
    sv = extra + 4 * (size_t)S; uv = extra + 5 * (size_t)S; pv = extra + 6 * (size_t)S;
    double ml = 1e300;
    for (int row = 1 + tid; row < S; row += nt) ml = fmin(ml, loss[row]);
    ml = -wg_max(-ml, red);
    const bool want = drift_on && cB >= 0.0 && cB < 0.25 * ml && !nonfinite(cA) && !nonfinite(cC) && !nonfinite(R0) && !nonfinite(ml);
    if (want) {
      double* res = U[0]; double* dx = U[1];
      // x <- W^-1 r,  W = diag(loss - shift) - F - K = -(M_c + shift I),  r = r1 e_0 (+ radd);  false if the defect correction stalls
      auto gs = [&](const double shift, double* x, const double r1, const double* radd) {
        cAe = 0.0;
        for (int row = tid; row < S; row += nt) { dinv[row] = (row == 0) ? 1.0 : 1.0 / (loss[row] - shift); x[row] = 0.0; }
        __syncthreads();
        bool ok = false;
        double ref = 0.0;                                               // correction size after 20 sweeps: the iteration must have shrunk it 30 sweeps later
        for (int it = 0; it < 600; ++it) {
          rhs_into(x, f, 0, nullptr);                                   // f = M_c x on the cube rows (x[0] = 0: no mRNA coupling)
          for (int row = tid; row < S; row += nt)
            res[row] = (row == 0) ? 0.0 : ((row == 1 ? r1 : 0.0) + (radd ? radd[row] : 0.0)) + f[row] + shift * x[row];
          __syncthreads();
          solve(res, dx);
          double mdx = 0.0, mxx = 0.0;
          for (int row = tid; row < S; row += nt) { const double xn = x[row] + dx[row]; x[row] = xn; mdx = fmax(mdx, fabs(dx[row])); mxx = fmax(mxx, fabs(xn)); }
          mdx = wg_max(mdx, red); mxx = wg_max(mxx, red);
          if (!(mdx == mdx) || !(mxx < 1e300)) break;
          if (mdx <= 1e-14 * mxx + 1e-300) { ok = true; break; }
          if (it == 20) ref = mdx;
          if (it >= 50 && (it % 30) == 20) { if (mdx > 0.5 * ref) break; ref = mdx; }        // stalling or diverging (W_B indefinite: B beyond the cube's slowest rate)
        }
        __syncthreads();
        return ok;
      };
      bool ok = gs(cB, uv, -cC * (cA - cB * R0), nullptr);            // W_B u = -C (A - B R0) e_0
      if (ok) ok = gs(0.0, sv, cC * R0, uv);                           // W_0 s =  C R0 e_0 + u
      if (ok) {
        drift = true;
        for (int row = tid; row < S; row += nt) y[row] = (row == 0) ? 0.0 : y[row] - sv[row];
        __syncthreads();
        set_p(0.0);
        pofs = pv;
        cAe = 0.0;
      } else {
        cAe = cA;
      }
    }
  }
  double h;
  {
    rhs_into(y, f, 0, nullptr);
    const double d0 = norm_of(y, y, y), d1 = norm_of(f, y, y);
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
    if (A.h0 > 0.0) h = A.h0;
    if (!(h > 0.0) || h != h) h = 1e-6;
  }
  auto fail_from = [&](int kk) { for (; kk < T; ++kk) out.emit(kk, y, true); };
  bool after_reject = false;
  if constexpr (ARK) {
    // ---- ARK4(3)6L[2]SA as a linearly implicit additive method (see pk_network_solve_ark.hpp): implicit operator A~ = g I - P with
    // P = (D_g - F) D_g^-1 (D_g - K) the approximate factorisation the sweeps invert; stage solves P Y = g r; h A~ Y = Y / gamma - h (g r)
    constexpr double AGAM = 0.25;
    constexpr double AE[5][5] = {{0.5, 0, 0, 0, 0}, {0.221776, 0.110224, 0, 0, 0},
                                 {-0.04884659515311858, -0.177720652326401, 0.8465672474795196, 0, 0},
                                 {-0.15541685842491548, -0.3567050098221991, 1.0587258798684427, 0.30339598837867193, 0},
                                 {0.20142435067267633, 0.008742057842904185, 0.15993995707168115, 0.4038290605220775, 0.22606457389066084}};
    constexpr double DI[5][5] = {{-0.25, 0, 0, 0, 0}, {-0.084, -0.166, 0, 0, 0},
                                 {0.19348346118010076, -0.04621125528694374, -0.39727220589315704, 0, 0},
                                 {0.2536756417084803, -0.23483923299747125, -0.24860482604014308, -0.020231582670865906, 0},
                                 {-0.04350805551100497, -0.008742057842904185, 0.02681898345231962, 0.27673623478725706, -0.5013051048856676}};
    constexpr double BB[6] = {0.15791629516167136, 0.0, 0.18675894052400077, 0.6805652953093346, -0.27524053099500667, 0.25};
    constexpr double EB[6] = {0.0032044943984591762, 0.0, -0.0024462511366794577, -0.02148007591958727, 0.043946868068572426, -0.02322503541076487};
    double* Y = Ys;  double* w = U[0];  double* v = U[1];  double* gr = U[2];
    double* Rr[4] = {U[3], f, extra, extra + S};                        // R_3 .. R_6
    double* sb = extra + 2 * (size_t)S;  double* se = extra + 3 * (size_t)S;
    while (true) {
      if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; fail_from(k); break; }
      const bool last = (tc + 1.0001 * h >= te);
      const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
      if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; fail_from(k); break; }
      const double g = 1.0 / (hs * AGAM);
      for (int row = tid; row < S; row += nt) dinv[row] = 1.0 / (g + loss[row]);
      __syncthreads();
      // stage 1: w = h f(y);  v = h (g y - P y),  P y = (D_g - F) t2,  t2 = D_g^-1 (D_g - K) y  (two plain products: no level order needed)
      rhs_into(y, w, 0, nullptr);
      for (int row = tid; row < S; row += nt) {                          // t2 into gr
        if (row == 0) { gr[0] = y[0]; continue; }
        const int m = row - 1;
        double hi = 0.0;
        for (int mm = ~m & (NM - 1); mm; mm &= mm - 1) hi += y[1 + (m | (mm & -mm))];
        gr[row] = __builtin_fma(-hi, dinv[row], y[row]);
      }
      __syncthreads();
      for (int row = tid; row < S; row += nt) {
        double Py;
        if (row == 0) Py = (g + loss[0]) * gr[0];
        else {
          const int m = row - 1;
          double lo = 0.0;
          for (int mm = m; mm; mm &= mm - 1) lo += gr[1 + (m ^ (mm & -mm))];
          Py = (g + loss[row]) * gr[row] - ((m == 0) ? cC * gr[0] : Sr[__builtin_ctz(m)] * lo);
        }
        const double ww = hs * w[row], vv = hs * (g * y[row] - Py);
        w[row] = ww; v[row] = vv;
        sb[row] = BB[0] * ww; se[row] = EB[0] * ww;
        for (int ii = 0; ii < 4; ++ii) Rr[ii][row] = y[row] + AE[ii + 1][0] * ww + DI[ii + 1][0] * vv;
      }
      __syncthreads();
      for (int s = 2; s <= 6; ++s) {
        for (int row = tid; row < S; row += nt) {
          const double rk = (s == 2) ? y[row] + AE[0][0] * w[row] + DI[0][0] * v[row] : Rr[s - 3][row];
          gr[row] = g * rk;
        }
        __syncthreads();
        solve(gr, Y);
        rhs_into(Y, w, 0, nullptr);
        for (int row = tid; row < S; row += nt) {
          const double vv = __builtin_fma(-hs, gr[row], (1.0 / AGAM) * Y[row]);
          const double ww = hs * w[row];
          v[row] = vv; w[row] = ww;
          sb[row] = __builtin_fma(BB[s - 1], ww, sb[row]);
          se[row] = __builtin_fma(EB[s - 1], ww, se[row]);
          for (int ii = 0; ii < 4; ++ii) if (ii + 3 > s) Rr[ii][row] += AE[ii + 1][s - 1] * ww + DI[ii + 1][s - 1] * vv;
        }
        __syncthreads();
      }
      for (int row = tid; row < S; row += nt) sb[row] += y[row];         // y_{n+1}
      __syncthreads();
      const double err = norm_of(se, y, sb);
      if (err != err || err > 1e300) {
        ++nrej; after_reject = true; h = 0.1 * hs;
        double bad = 0.0;
        for (int row = tid; row < S; row += nt) if (nonfinite(y[row]) || nonfinite(loss[row])) bad = 1.0;
        if (nonfinite(cA) || nonfinite(cC)) bad = 1.0;
        if (wg_max(bad, red) != 0.0) { status |= PK_ST_NONFINITE; fail_from(k); break; }
        continue;
      }
      double fac = sqrt(sqrt(err)) * (1.0 / 0.9);
      fac = fmax(1.0 / 6.0, fmin(5.0, fac));
      double hnew = hs / fac;
      if (err <= 1.0) {
        ++nacc;
        for (int row = tid; row < S; row += nt) y[row] = sb[row];
        __syncthreads();
        tc += hs;
        if (after_reject) hnew = fmin(hnew, hs);
        after_reject = false;
        if (last) {
          tc = te;
          if (drift) {
            set_p(tc - t0);
            for (int row = tid; row < S; row += nt) Y[row] = y[row] + pv[row];
            __syncthreads();
            out.emit(k, Y, false);
          } else {
            out.emit(k, y, false);
          }
          ++k;
          h = (hs < h) ? fmax(hnew, h) : hnew;
          if (k >= T) break;
          te = A.t[k];
        } else {
          if (drift) set_p(tc - t0);
          h = hnew;
        }
      } else {
        ++nrej; after_reject = true;
        h = hnew;
      }
    }
    out.finish(status, nacc, nrej);
    return;
  }
  while (true) {
    if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; fail_from(k); break; }
    const bool last = (tc + 1.0001 * h >= te);
    const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
    if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; fail_from(k); break; }
    const double hinv = 1.0 / hs, g = hinv * (1.0 / GAM);
    for (int row = tid; row < S; row += nt) dinv[row] = 1.0 / (g + loss[row]);
    __syncthreads();
    for (int sg = 0; sg < 4; ++sg) {
      const double* Y = y;
      if (sg > 0) {
        for (int row = tid; row < S; row += nt) {
          double v = y[row];
          for (int u = 0; u < sg; ++u) v = __builtin_fma(TA[sg][u], U[u][row], v);
          Ys[row] = v;
        }
        __syncthreads();
        Y = Ys;
      }
      double coef[3];
      for (int u = 0; u < 3; ++u) coef[u] = TC[sg][u] * hinv;
      rhs_into(Y, f, sg, coef);
      solve(f, U[sg]);
    }
    // new value = stage-4 point + U4 (stiffly accurate); error estimate = sum E_i U_i (kept in f)
    for (int row = tid; row < S; row += nt) {
      Ys[row] += U[3][row];
      f[row] = TE[0] * U[0][row] + TE[1] * U[1][row] + TE[2] * U[2][row] + TE[3] * U[3][row];
    }
    __syncthreads();
    const double err = norm_of(f, y, Ys);
    if (err != err || err > 1e300) {
      ++nrej; after_reject = true; h = 0.1 * hs;
      double bad = 0.0;
      for (int row = tid; row < S; row += nt) if (nonfinite(y[row]) || nonfinite(loss[row])) bad = 1.0;
      if (nonfinite(cA) || nonfinite(cC)) bad = 1.0;
      if (wg_max(bad, red) != 0.0) { status |= PK_ST_NONFINITE; fail_from(k); break; }
      continue;
    }
    double fac = cbrt(err) * (1.0 / 0.9);
    fac = fmax(1.0 / 6.0, fmin(5.0, fac));
    double hnew = hs / fac;
    if (err <= 1.0) {
      ++nacc;
      for (int row = tid; row < S; row += nt) y[row] = Ys[row];
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

// =====================================================================================================================================
// Right-hand side / Jacobian / steady state for the wide systems of the distributive and successive models: one thread per (replica, row).
template <int MODEL>
__global__ void chain_rhs_wide_kernel(const double* __restrict__ theta, const double* __restrict__ y, double* __restrict__ dydt,
                                      const long long B, const int n, const int S, const int P) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= B * S) return;
  const long long rep = gid / S;
  const int row = (int)(gid - rep * S);
  const RowCoef c = load_row<MODEL>(theta + rep * P, n, S, row);
  const double* yr = y + rep * S;
  double f = __builtin_fma(c.dg, yr[row], c.bias);
  if constexpr (MODEL == M_DIST) {
    if (row == 1) { double s = 0.0; for (int i = 2; i < S; ++i) s += yr[i]; f = __builtin_fma(c.c2, yr[0], f) + s; }
    else if (row >= 2) f = __builtin_fma(c.c1, yr[1], f);
  } else {
    if (row >= 1) f = __builtin_fma(c.c1, yr[row - 1], f);
    if (row + 1 < S) f = __builtin_fma(c.c2, yr[row + 1], f);
  }
  dydt[gid] = f;
}

template <int MODEL>
__global__ void chain_jac_wide_kernel(const double* __restrict__ theta, double* __restrict__ J, const long long B, const int n,
                                      const int S, const int P) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= B * S) return;
  const long long rep = gid / S;
  const int row = (int)(gid - rep * S);
  const RowCoef c = load_row<MODEL>(theta + rep * P, n, S, row);
  double* Jr = J + gid * S;
  for (int j = 0; j < S; ++j) Jr[j] = 0.0;
  if constexpr (MODEL == M_DIST) {
    if (row == 1) { for (int j = 2; j < S; ++j) Jr[j] = 1.0; Jr[0] = c.c2; }
    else if (row >= 2) Jr[1] = c.c1;
  } else {
    if (row >= 1) Jr[row - 1] = c.c1;
    if (row + 1 < S) Jr[row + 1] = c.c2;
  }
  Jr[row] = c.dg;
}


// =====================================================================================================================================
// Steady states y* of dy/dt = J y + b = 0 for the wide systems (pk_steady_state_protein_batch beyond 64 states).
//   distmod: closed form of the arrow system;  succmod: the tridiagonal system -J y = b by cyclic reduction in LDS;
//   randmod: -J y = b on the n-cube by symmetric Gauss-Seidel sweeps over the popcount levels (an M-matrix with strictly dominant columns
//   whenever the degradation rates are positive: the iteration converges; tolerance 1e-13 relative, budget 200 000 sweeps).
// A singular J (no degradation) gives non-finite values: PK_ST_NONFINITE and a NaN row; an exhausted budget: PK_ST_MAXSTEPS and a NaN row.
__host__ __device__ inline size_t wide_steady_chain_lds_bytes(int S) { return ((size_t)11 * S + 24) * sizeof(double); }

template <int MODEL>
__global__ __launch_bounds__(256) void wide_steady_chain_kernel(const double* __restrict__ theta, double* __restrict__ yss, int32_t* __restrict__ status,
                                                                const long long B, const int n, const int S, const int P) {
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const long long rep = blockIdx.x;
  if (rep >= B) return;
  const double* __restrict__ th = theta + rep * P;
  double* c1 = lds;       double* dgv = c1 + S;    double* c2 = dgv + S;
  double* pa = c2 + S;    double* pb = pa + S;     double* pc = pb + S;    double* pr = pc + S;
  double* qa = pr + S;    double* qb = qa + S;     double* qc = qb + S;    double* qr = qc + S;
  double* red = qr + S;
  for (int row = tid; row < S; row += nt) { const RowCoef c = load_row<MODEL>(th, n, S, row); c1[row] = c.c1; dgv[row] = c.dg; c2[row] = c.c2; }
  __syncthreads();
  double* x = pr;                                               // result vector
  if constexpr (MODEL == M_DIST) {
    // A - B R = 0 ; S_i P - (1 + D_i) X_i = 0 ; C R - Dsum P + sum X_i = 0
    double loc = 0.0;
    for (int row = 2 + tid; row < S; row += nt) { const double cw = c1[row] / (-dgv[row]); pa[row] = cw; loc += cw; }
    const double Scw = wg_sum(loc, red);
    const double R = th[0] / (-dgv[0]);
    const double Pv = c2[1] * R / (-dgv[1] - Scw);
    __syncthreads();
    for (int row = tid; row < S; row += nt) x[row] = (row == 0) ? R : (row == 1 ? Pv : pa[row] * Pv);
    __syncthreads();
  } else {
    double *sa = pa, *sb = pb, *sc = pc, *sr = pr, *da = qa, *db = qb, *dc = qc, *dr = qr;
    for (int row = tid; row < S; row += nt) { sa[row] = -c1[row]; sb[row] = -dgv[row]; sc[row] = -c2[row]; sr[row] = (row == 0) ? th[0] : 0.0; }
    __syncthreads();
    for (int d = 1; d < S; d <<= 1) {
      for (int row = tid; row < S; row += nt) {
        const int lo = row - d, hi = row + d;
        double na = 0.0, nc = 0.0, nb = sb[row], nr = sr[row];
        if (lo >= 0) { const double al = -sa[row] / sb[lo]; na = al * sa[lo]; nb = __builtin_fma(al, sc[lo], nb); nr = __builtin_fma(al, sr[lo], nr); }
        if (hi < S) { const double ga = -sc[row] / sb[hi]; nc = ga * sc[hi]; nb = __builtin_fma(ga, sa[hi], nb); nr = __builtin_fma(ga, sr[hi], nr); }
        da[row] = na; db[row] = nb; dc[row] = nc; dr[row] = nr;
      }
      __syncthreads();
      double* t;
      t = sa; sa = da; da = t; t = sb; sb = db; db = t; t = sc; sc = dc; dc = t; t = sr; sr = dr; dr = t;
    }
    for (int row = tid; row < S; row += nt) c1[row] = sr[row] / sb[row];       // c1 is free by now
    __syncthreads();
    x = c1;
  }
  double bad = 0.0;
  for (int row = tid; row < S; row += nt) if (nonfinite(x[row])) bad = 1.0;
  bad = wg_max(bad, red);
  for (int row = tid; row < S; row += nt) yss[rep * S + row] = (bad != 0.0) ? __builtin_nan("") : x[row];
  if (tid == 0 && status) status[rep] = (bad != 0.0) ? PK_ST_NONFINITE : PK_ST_OK;
}

__host__ __device__ inline size_t wide_steady_rand_lds_bytes(int n) {
  const size_t NM = (size_t)1 << n;
  return ((size_t)2 * (NM + 1) + n + 24) * sizeof(double) + (NM + (n + 2) + 21 * 21) * sizeof(int);
}

__global__ __launch_bounds__(256) void wide_steady_rand_kernel(const double* __restrict__ theta, double* __restrict__ yss, int32_t* __restrict__ status,
                                                               const long long B, const int n, const int S, const int P) {
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int NM = 1 << n;
  const long long rep = blockIdx.x;
  if (rep >= B) return;
  const double* __restrict__ th = theta + rep * P;
  double* x = lds;            double* loss = x + S;       double* Sr = loss + S;      double* red = Sr + n;
  int* lvl = reinterpret_cast<int*>(red + 24);   int* binom = lvl + (n + 2);   int* ord = binom + 21 * 21;
  const double cA = th[0], cB = th[1], cC = th[2], cD = th[3];
  if (tid == 0) {
    for (int a = 0; a <= 20; ++a) for (int b = 0; b <= 20; ++b) binom[a * 21 + b] = (b == 0) ? 1 : (a == 0 ? 0 : binom[(a - 1) * 21 + b - 1] + binom[(a - 1) * 21 + b]);
    lvl[0] = 0;
    for (int L = 0; L <= n; ++L) lvl[L + 1] = lvl[L] + binom[n * 21 + L];
  }
  for (int j = tid; j < n; j += nt) Sr[j] = th[4 + j];
  __syncthreads();
  double sumS = 0.0;
  for (int j = 0; j < n; ++j) sumS += Sr[j];
  const double R = cA / cB;
  for (int m = tid; m < NM; m += nt) {
    int kk = __popc(m), rank = 0;
    { int c = kk; for (int j = n - 1; j >= 0 && c > 0; --j) if ((m >> j) & 1) { rank += binom[j * 21 + c]; --c; } }
    ord[lvl[kk] + rank] = m;
    double l;
    if (m == 0) l = cD + sumS;
    else {
      const int lsb = __builtin_ctz(m);
      double o = 0.0;
      for (int j = 0; j < n; ++j) o += ((m >> j) & 1) ? 1.0 : Sr[j < lsb ? j : lsb];
      l = o + th[4 + n + m - 1];
    }
    loss[1 + m] = l;
    x[1 + m] = 0.0;
  }
  if (tid == 0) x[0] = R;
  __syncthreads();
  // one state update: x_m = (source_m + S[lsb m] sum_{j in m} x_{m ^ j} + sum_{j not in m} x_{m | j}) / loss_m ; returns |change| / scale
  auto relax = [&](const int m) {
    double lo = 0.0, hi = 0.0;
    for (int mm = m; mm; mm &= mm - 1) lo += x[1 + (m ^ (mm & -mm))];
    for (int mm = ~m & (NM - 1); mm; mm &= mm - 1) hi += x[1 + (m | (mm & -mm))];
    const double src = (m == 0) ? cC * R : Sr[__builtin_ctz(m)] * lo;
    const double xn = (src + hi) / loss[1 + m];
    const double d = fabs(xn - x[1 + m]) / (fabs(xn) + 1e-300);
    x[1 + m] = xn;
    return d;
  };
  int st = PK_ST_MAXSTEPS;
  for (int it = 0; it < 200000; ++it) {
    double dmax = 0.0;
    for (int L = 0; L <= n; ++L) {
      for (int idx = lvl[L] + tid; idx < lvl[L + 1]; idx += nt) { const double d = relax(ord[idx]); dmax = (d > dmax || d != d) ? d : dmax; }
      __syncthreads();
    }
    for (int L = n - 1; L >= 0; --L) {
      for (int idx = lvl[L] + tid; idx < lvl[L + 1]; idx += nt) { const double d = relax(ord[idx]); dmax = (d > dmax || d != d) ? d : dmax; }
      __syncthreads();
    }
    dmax = wg_max(dmax, red);
    if (dmax != dmax) { st = PK_ST_NONFINITE; break; }
    if (dmax <= 1e-13) { st = PK_ST_OK; break; }
  }
  double bad = (st != PK_ST_OK) ? 1.0 : 0.0;
  for (int row = tid; row < S; row += nt) if (nonfinite(x[row])) bad = 1.0;
  bad = wg_max(bad, red);
  if (bad != 0.0 && st == PK_ST_OK) st = PK_ST_NONFINITE;
  for (int row = tid; row < S; row += nt) yss[rep * S + row] = (bad != 0.0) ? __builtin_nan("") : x[row];
  if (tid == 0 && status) status[rep] = st;
}

}  // namespace pk
