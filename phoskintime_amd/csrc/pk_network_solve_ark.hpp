// pk_network_solve_ark.hpp -- order-4 network integrator: ARK4(3)6L[2]SA (Kennedy & Carpenter 2003) as a LINEARLY IMPLICIT additive
// Runge-Kutta method, one thread per protein (the layout of pk_network_solve_reg.hpp).
//
//   y' = [f(y) - A y]  (explicit tableau a^E)  +  A y  (L-stable ESDIRK tableau a^I, gamma = 1/4),
//   A  = the per-protein BLOCK-DIAGONAL part of the Jacobian at the step start.
//
// The additive order conditions -- including every coupling condition between the two tableaus -- hold for ANY splitting of the right-hand
// side, hence for any fixed matrix A (tools/check_ark436.py verifies them in exact rational arithmetic: residuals <= 3e-26): order 4,
// embedded order 3, no matter how much of the Jacobian A leaves out.  That is the licence ROS34PW2 (order 3) drew from being a W-method,
// one order higher.  Every implicit stage is ONE block solve with (g I - A), g = 1 / (gamma h): exactly the cost of a Rosenbrock-W stage.
//
//   stage 1: Y_1 = y_n
//   stage i: r_i = y_n + h sum_{j<i} [ a^E_ij F_j + d_ij G_j ],  d = a^I - a^E,  F_j = f(Y_j),  G_j = A Y_j ;   (g I - A) Y_i = g r_i ;
//            G_i = g (Y_i - r_i)   (no block product needed: it falls out of the solve)
//   y_{n+1} = y_n + h sum_j b_j F_j ,   err = h sum_j (b_j - bhat_j) F_j ,   step factor err^(-1/4).
//
// Measured against ROS34PW2 on the reference-run networks (numpy model tools/proto_ark_network.py, then this kernel): 5-7x fewer steps
// at rtol = atol = 1e-8, i.e. 3.5-5x fewer block solves, at band errors of the reference LSODA run's own size.
//
// Registers: y, Y, w = h F, v = h G and the scaled right-hand side of the current stage: 5 block vectors.  The right-hand sides R_3..R_6
// of the later stages and the two running sums (b, b - bhat) accumulate in LDS, thread-private ([slot][state]), touched once per stage.
#pragma once
#include "pk_network_solve_reg.hpp"

namespace pk {

namespace ark436 {
constexpr double GAM = 0.25;
// a^E_ij, rows i = 2..6 (index [i - 2][j - 1])
__device__ constexpr double AE[5][5] = {
    {0.5, 0, 0, 0, 0},
    {0.221776, 0.110224, 0, 0, 0},
    {-0.04884659515311858, -0.177720652326401, 0.8465672474795196, 0, 0},
    {-0.15541685842491548, -0.3567050098221991, 1.0587258798684427, 0.30339598837867193, 0},
    {0.20142435067267633, 0.008742057842904185, 0.15993995707168115, 0.4038290605220775, 0.22606457389066084}};
// d_ij = a^I_ij - a^E_ij
__device__ constexpr double DI[5][5] = {
    {-0.25, 0, 0, 0, 0},
    {-0.084, -0.166, 0, 0, 0},
    {0.19348346118010076, -0.04621125528694374, -0.39727220589315704, 0, 0},
    {0.2536756417084803, -0.23483923299747125, -0.24860482604014308, -0.020231582670865906, 0},
    {-0.04350805551100497, -0.008742057842904185, 0.02681898345231962, 0.27673623478725706, -0.5013051048856676}};
__device__ constexpr double B[6] = {0.15791629516167136, 0.0, 0.18675894052400077, 0.6805652953093346, -0.27524053099500667, 0.25};
__device__ constexpr double EB[6] = {0.0032044943984591762, 0.0, -0.0024462511366794577, -0.02148007591958727, 0.043946868068572426, -0.02322503541076487};
}  // namespace ark436

// LDS parking: thread-private accumulators, slot-major [slot][row][thread] (conflict-free; padding rows are stored like the others: they
// hold exact zeros).  Slots 0-3: the right-hand sides R_3 .. R_6; with PK_ARK_PARK_SUMS also 4: sum b_j w_j, 5: sum (b_j - bhat_j) w_j.
#ifndef PK_ARK_PARK_SUMS
#define PK_ARK_PARK_SUMS 0
#endif
constexpr int kArkSlots = PK_ARK_PARK_SUMS ? 6 : 4;
__host__ __device__ constexpr size_t net_ark_park_doubles(int rows, int threads) { return (size_t)kArkSlots * rows * threads; }

template <int MODEL, int MAXS>
__global__ __launch_bounds__(256, 2) void net_solve_ark_kernel(const NetDev n, const NetSolveArgs A) {
  using namespace ark436;
  constexpr int NR = 2 + MAXS;                       // rows of a block vector: R, P, sites
  extern __shared__ __align__(16) double lds[];
  const int N = n.N, S = n.S;
  double* Kt = lds;                       // [n_K]
  double* Pv = Kt + n.n_K;                // [2][N]
  double* red = Pv + 2 * N;               // [24]
  const int nnzT = n.TF_indptr[N];
  double* tf_dat = red + 24;              // [nnzT]
  int32_t* tf_idx = reinterpret_cast<int32_t*>(tf_dat + nnzT);
  double* park = tf_dat + nnzT + (nnzT + 1) / 2;     // [kArkSlots][NR][nt]
  const NetSlices sl(n.n_K, N, n.sites);
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int i = tid;
  const bool own = i < N;
  const double* stops = A.stops_p ? A.stops_p : A.stops_v;
  const int32_t* stop_out = A.stop_out_p ? A.stop_out_p : A.stop_out_v;
  const double* xb = A.x + b * n.n_var;
  auto par = [&](int off) { const double v = xb[off]; return A.x_is_raw ? softplus(v) : v; };
  for (int k = tid; k < nnzT; k += nt) { tf_dat[k] = n.TF_data[k]; tf_idx[k] = n.TF_indices[k]; }

  const int st = own ? n.offset_y[i] : 0, ss = own ? n.offset_s[i] : 0, ns = own ? n.n_sites[i] : 0, drv = own ? n.driver_map[i] : -1;
  const int tf0 = own ? n.TF_indptr[i] : 0, tf1 = own ? n.TF_indptr[i + 1] : 0;
  const double tfdeg_inv = own ? 1.0 / n.tf_deg[i] : 1.0;
  const double Ai = own ? par(sl.A + i) : 0.0, Bi = own ? par(sl.B + i) : 1.0, Ci = own ? par(sl.C + i) : 0.0, Di = own ? par(sl.D + i) : 1.0,
               Ei = own ? par(sl.E + i) : 0.0, ts = par(sl.tf);
  double Dp[MAXS], Sr[MAXS];
#pragma unroll
  for (int j = 0; j < MAXS; ++j) { Dp[j] = (j < ns) ? par(sl.Dp + ss + j) : 0.0; Sr[j] = 0.0; }
  const double* y0 = A.y0 + (A.y0_batched ? b * S : 0);
  double* Yout = A.Y + b * (size_t)A.T * S;
  // block vectors as arrays of NR doubles: [0] = mRNA, [1] = protein, [2 + j] = site j
  double y[NR];
  y[0] = own ? y0[st] : 0.0; y[1] = own ? y0[st + 1] : 0.0;
#pragma unroll
  for (int j = 0; j < MAXS; ++j) y[2 + j] = (j < ns) ? y0[st + 2 + j] : 0.0;
  auto write_row = [&](int row) {
    if (!own) return;
    double* o = Yout + (size_t)row * S + st;
    o[0] = y[0]; o[1] = y[1];
#pragma unroll
    for (int j = 0; j < MAXS; ++j) if (j < ns) o[2 + j] = y[2 + j];
  };
  write_row(0);

  auto set_bucket = [&](const int jb) {
    __syncthreads();
    for (int k = tid; k < n.n_K; k += nt) Kt[k] = n.kin_Kmat[(size_t)k * n.n_grid + jb] * (A.x_is_raw ? softplus(xb[sl.ck + k]) : xb[sl.ck + k]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MAXS; ++j) {
      double acc = 0.0;
      if (j < ns) for (int q = n.W_indptr[ss + j]; q < n.W_indptr[ss + j + 1]; ++q) acc += n.W_data[q] * Kt[n.W_indices[q]];
      Sr[j] = acc;
    }
  };

  // f(Y) of the whole block (same arithmetic as rhs_block of pk_network_solve_reg.hpp); ends after ONE barrier
  int buf = 0;
  auto rhs_block = [&](const double (&Y)[NR], double (&f)[NR]) {
    double tot;
    if (drv >= 0) tot = Kt[drv];
    else { tot = Y[1];
#pragma unroll
      for (int j = 0; j < MAXS; ++j) tot += Y[2 + j]; }
    if (own) Pv[buf * N + i] = tot;
    __syncthreads();
    double acc = 0.0;
    for (int e = tf0; e < tf1; ++e) acc += tf_dat[e] * Pv[buf * N + tf_idx[e]];
    buf ^= 1;
    double v = acc * tfdeg_inv;
    if (MODEL != 4) v = v * net_rcp(1.0 + fabs(v));
    f[0] = synth_rate_fast(Ai, ts, v) - Bi * Y[0];
    if (MODEL == 0) {
      double sumS = 0.0, back = 0.0;
#pragma unroll
      for (int j = 0; j < MAXS; ++j) { sumS += Sr[j]; back += Ei * Y[2 + j]; f[2 + j] = Sr[j] * Y[1] - (Ei + Dp[j] + Di) * Y[2 + j]; }
      f[1] = Ci * Y[0] - (Di + sumS) * Y[1] + back;
    } else if (MODEL == 4) {
      const double q = Y[1] * net_rcp(1.0 + Y[1]);
      double fw_ = 0.0, back = 0.0;
#pragma unroll
      for (int j = 0; j < MAXS; ++j) { const double fw = Sr[j] * q; fw_ += fw; back += Ei * Y[2 + j]; f[2 + j] = fw - (Dp[j] + Di) * Y[2 + j] - Ei * Y[2 + j]; }
      f[1] = (Ci * Y[0]) * net_rcp(1.0 + Y[0]) - Di * Y[1] - fw_ + back;
    } else {
      f[1] = Ci * Y[0] - Di * Y[1] - Sr[0] * Y[1] + Ei * Y[2];
      static_for<MAXS>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        double next = 0.0, knext = 0.0;
        if constexpr (j + 1 < MAXS) { next = Y[3 + j]; knext = Sr[j + 1]; }
        f[2 + j] = Sr[j] * Y[1 + j] + Ei * next - (knext + Ei + Dp[j] + Di) * Y[2 + j];
      });
    }
  };

  // A = block Jacobian at the step start: its entries (frozen per step), the product A Y and the factors of g I - A
  double cRv = 0.0, gPv = 1.0, sumSg = 0.0;          // d f_P / d R ; saturation factor 1 / (1 + P)^2 ; sum_j S_j gPv
  double winvR = 1.0, sinv = 1.0, wv[MAXS + 1];
  auto freeze = [&]() {
    const bool sat = MODEL == 4;
    gPv = sat ? net_rcp((1.0 + y[1]) * (1.0 + y[1])) : 1.0;
    cRv = sat ? Ci * net_rcp((1.0 + y[0]) * (1.0 + y[0])) : Ci;
    sumSg = 0.0;
#pragma unroll
    for (int j = 0; j < MAXS; ++j) sumSg += Sr[j] * gPv;
  };
  auto block_matvec = [&](const double (&Y)[NR], double (&G)[NR]) {
    G[0] = -Bi * Y[0];
    if (MODEL == 1) {
      G[1] = cRv * Y[0] - (Di + Sr[0]) * Y[1] + Ei * Y[2];
      static_for<MAXS>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        double next = 0.0, knext = 0.0;
        if constexpr (j + 1 < MAXS) { next = Y[3 + j]; knext = Sr[j + 1]; }
        G[2 + j] = Sr[j] * Y[1 + j] + Ei * next - (knext + Ei + Dp[j] + Di) * Y[2 + j];
      });
    } else {
      double back = 0.0;
#pragma unroll
      for (int j = 0; j < MAXS; ++j) { back += Ei * Y[2 + j]; G[2 + j] = (Sr[j] * gPv) * Y[1] - (Ei + Dp[j] + Di) * Y[2 + j]; }
      G[1] = cRv * Y[0] - (Di + sumSg) * Y[1] + back;
    }
  };
  auto factor = [&](const double g) {
    winvR = net_rcp(g + Bi);
    if (MODEL == 1) {
      double d = g + Di + Sr[0];
      wv[0] = net_rcp(d);
      static_for<MAXS>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        double knext = 0.0;
        if constexpr (j + 1 < MAXS) knext = Sr[j + 1];
        d = (g + Ei + Dp[j] + Di + knext) - (Sr[j] * Ei) * wv[j];
        wv[j + 1] = net_rcp(d);
      });
    } else {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < MAXS; ++j) { const double w = net_rcp(g + Ei + Dp[j] + Di); wv[j] = w; acc += Ei * (Sr[j] * gPv) * w; }
      sinv = net_rcp(g + Di + sumSg - acc);
    }
  };
  // x = (g I - A)^-1 r
  auto block_solve = [&](const double (&r)[NR], double (&x)[NR]) {
    const double xR = r[0] * winvR;
    x[0] = xR;
    if (MODEL == 1) {
      double fw[MAXS + 1];
      fw[0] = r[1] + cRv * xR;
      static_for<MAXS>([&](auto jc) { constexpr int j = decltype(jc)::value; fw[j + 1] = r[2 + j] + Sr[j] * fw[j] * wv[j]; });
      double xn = 0.0;
      static_for<MAXS>([&](auto jc) { constexpr int q = MAXS - decltype(jc)::value; xn = (fw[q] + Ei * xn) * wv[q]; x[1 + q] = xn; });
      x[1] = (fw[0] + Ei * xn) * wv[0];
    } else {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < MAXS; ++j) { const double t = r[2 + j] * wv[j]; x[2 + j] = t; acc += Ei * t; }
      const double xP = (r[1] + cRv * xR + acc) * sinv;
      x[1] = xP;
#pragma unroll
      for (int j = 0; j < MAXS; ++j) x[2 + j] += (Sr[j] * gPv) * wv[j] * xP;
    }
  };
  double* const mypark = park + tid;
  auto park_ld = [&](int slot, int row) { return mypark[(size_t)(slot * NR + row) * nt]; };
  auto park_st = [&](int slot, int row, double x) { mypark[(size_t)(slot * NR + row) * nt] = x; };

  __syncthreads();
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  double tc = A.t0;
  int jb = net_bucket(tc, n.kin_grid, n.n_grid);
  set_bucket(jb);
  double h;
  {
    double f[NR];
    rhs_block(y, f);
    auto q = [&](double v, double yv) { return fabs(v) / (A.atol + A.rtol * fabs(yv)); };
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int k = 0; k < NR; ++k) if (own && k < 2 + ns) { d0 = fmax(d0, q(y[k], y[k])); d1 = fmax(d1, q(f[k], y[k])); }
    d0 = block_max(d0, red); d1 = block_max(d1, red);
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
    if (A.h0 > 0.0) h = A.h0;
    if (!(h > 0.0) || h != h) h = 1e-6;
  }
  const bool rms = A.err_rms;
  const double safety_inv = 1.0 / (A.ctl_safety > 0.0 ? A.ctl_safety : 0.9), grow_inv = 1.0 / (A.ctl_grow > 1.0 ? A.ctl_grow : 6.0);
  bool after_reject = false;
  for (int si = 0; si < A.n_stops && status == PK_ST_OK; ++si) {
    const double te = stops[si];
    while (true) {
      if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; break; }
      const bool last = (tc + 1.0001 * h >= te);
      const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
      if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; break; }
      const double g = net_rcp(hs * GAM);
      freeze();
      factor(g);
      double Y[NR], w[NR], v[NR];
      // ---- stage 1: Y_1 = y_n
      rhs_block(y, w);
      block_matvec(y, v);
#pragma unroll
      for (int k = 0; k < NR; ++k) { w[k] *= hs; v[k] *= hs; }
      static_for<4>([&](auto ic) {                                           // R_3 .. R_6 start in LDS
        constexpr int ii = decltype(ic)::value;
#pragma unroll
        for (int k = 0; k < NR; ++k) park_st(ii, k, y[k] + AE[ii + 1][0] * w[k] + DI[ii + 1][0] * v[k]);
      });
      double sb[NR], se[NR];
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        if constexpr (PK_ARK_PARK_SUMS) { park_st(4, k, B[0] * w[k]); park_st(5, k, EB[0] * w[k]); }
        else { sb[k] = B[0] * w[k]; se[k] = EB[0] * w[k]; }
      }
      // ---- stages 2 .. 6.  Register economy: the right-hand side r_s lives only until the solve; h G_s = (Y_s - r_s) / gamma is taken
      // from the scaled right-hand side g r_s as  Y_s / gamma - h (g r_s)  before the stage's f-evaluation, so r_s is dead by then
      static_for<5>([&](auto sc) {
        constexpr int s = 2 + decltype(sc)::value;                           // stage number
        double gr[NR];                                                       // g * r_s
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          double rk;
          if constexpr (s == 2) rk = y[k] + AE[0][0] * w[k] + DI[0][0] * v[k]; else rk = park_ld(s - 3, k);
          gr[k] = g * rk;
        }
        block_solve(gr, Y);
#pragma unroll
        for (int k = 0; k < NR; ++k) v[k] = __builtin_fma(-hs, gr[k], (1.0 / GAM) * Y[k]);
        rhs_block(Y, w);
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          w[k] *= hs;
          if constexpr (s != 2) {                                            // b_2 = bhat_2 = 0
            if constexpr (PK_ARK_PARK_SUMS) {
              park_st(4, k, __builtin_fma(B[s - 1], w[k], park_ld(4, k)));
              park_st(5, k, __builtin_fma(EB[s - 1], w[k], park_ld(5, k)));
            } else { sb[k] = __builtin_fma(B[s - 1], w[k], sb[k]); se[k] = __builtin_fma(EB[s - 1], w[k], se[k]); }
          }
        }
        static_for<4>([&](auto ic) {                                         // later right-hand sides take this stage's contribution
          constexpr int ii = decltype(ic)::value;                            // R_{ii + 3}
          if constexpr (ii + 3 > s) {
#pragma unroll
            for (int k = 0; k < NR; ++k) park_st(ii, k, park_ld(ii, k) + AE[ii + 1][s - 1] * w[k] + DI[ii + 1][s - 1] * v[k]);
          }
        });
      });
      if constexpr (PK_ARK_PARK_SUMS) {
#pragma unroll
        for (int k = 0; k < NR; ++k) { sb[k] = park_ld(4, k); se[k] = park_ld(5, k); }
      }
      // ---- new value and error estimate
      auto q = [&](double ev, double ya, double yb) { return fabs(ev) * net_rcp(A.atol + A.rtol * fmax(fabs(ya), fabs(yb))); };
      double e = 0.0;
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        sb[k] += y[k];                                                       // y_{n+1}
        if (own && k < 2 + ns) e = err_acc(e, q(se[k], y[k], sb[k]), rms);
      }
      const double err = err_reduce(e, rms, S, red);
      if (err != err || err > 1e300) {
        ++nrej; after_reject = true; h = 0.1 * hs;
        double bad = (nonfinite(Ai) || nonfinite(Bi) || nonfinite(Ci) || nonfinite(Di) || nonfinite(Ei) || nonfinite(ts)) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < NR; ++k) if (nonfinite(y[k])) bad = 1.0;
#pragma unroll
        for (int j = 0; j < MAXS; ++j) if (nonfinite(Dp[j]) || nonfinite(Sr[j])) bad = 1.0;
        if (block_max(bad, red) != 0.0) { status |= PK_ST_NONFINITE; break; }
        continue;
      }
      double fac = sqrt(sqrt(err)) * safety_inv;                             // embedded order 3: err^(1/4)
      fac = fmax(grow_inv, fmin(5.0, fac));
      double hnew = hs * net_rcp(fac);
      if (err <= 1.0) {
        ++nacc;
#pragma unroll
        for (int k = 0; k < NR; ++k) y[k] = sb[k];
        tc += hs;
        if (after_reject) hnew = fmin(hnew, hs);
        after_reject = false;
        if (last) { tc = te; h = (hs < h) ? fmax(hnew, h) : hnew; break; }
        h = hnew;
      } else {
        ++nrej; after_reject = true;
        h = hnew;
      }
    }
    if (status != PK_ST_OK) break;
    const int row = stop_out[si];
    if (row >= 0) write_row(row);
    const int jn = net_bucket(tc, n.kin_grid, n.n_grid);
    if (jn != jb) { jb = jn; set_bucket(jb); }
  }
  if (status != PK_ST_OK && own) {
    const double qnan = __builtin_nan("");
    for (int si = 0; si < A.n_stops; ++si) {
      const int row = stop_out[si];
      if (row >= 0 && !(stops[si] <= tc)) { double* o = Yout + (size_t)row * S + st; for (int k = 0; k < 2 + ns; ++k) o[k] = qnan; }
    }
  }
  if (tid == 0) {
    if (A.status) A.status[b] = status;
    if (A.n_steps) { A.n_steps[2 * b] = nacc; A.n_steps[2 * b + 1] = nrej; }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Combinatorial topology (model 2), one thread per protein, 1 + 2^ns states (ns <= NB): the same additive method.  The register kernel
// of round 1 (pk_network_solve_reg2.hpp) never forms g I - A_block; it applies the approximate factorisation
//   P = (D_g - F) D_g^-1 (D_g - K),   D_g = g I + diag(loss),  F = phosphorylation (+ the C R -> state 0 coupling), K = dephosphorylation.
// An additive method accepts that as it stands: the implicit operator is simply A~ := g I - P (= A_block - F D_g^-1 K; it depends on the
// step size, which the order conditions do not mind), stage solves are  P Y_i = g r_i  (the two sweeps),  A~ Y_i = g (Y_i - r_i) falls out
// as before, and only stage 1 needs one product with P.  numpy model: tools/proto_ark_network.py solve_sgs.
//
// [r3] EXACT = true (the default since round 3): the implicit operator is the block Jacobian A itself, solved exactly.  The bit-pattern block
// is bipartite like the per-protein random model (pk_rand_parity.hpp): every transition flips one bit, so the odd-popcount states couple
// only to even ones and their diagonal block of W = g I - A is DIAGONAL.  Eliminating them leaves a Schur complement on the 2^(NB-1) even
// states (4 x 4 at three sites, 2 x 2 at two): built from short sums, inverted by an unrolled Gauss-Jordan in registers (no pivoting:
// M-matrix), 16 doubles.  A stage solve is then three in-thread passes (odd -> even right-hand side, 4 x 4 product, even -> odd), no more
// expensive than the two sweeps -- and exact, so the additive method sees the stiffness it was promised: half the steps of the
// approximate factorisation (numpy model tools/proto_ark_network.py: 540-580 against 1 100-1 400), and stage 1 needs A y, not P y.
template <int NB, bool EXACT = true>
__global__ __launch_bounds__(256, 2) void net_solve_ark2_kernel(const NetDev n, const NetSolveArgs A) {
  using namespace ark436;
  constexpr int NM = 1 << NB;
  constexpr int NE = NM / 2;                         // even-popcount states
  constexpr int NR = 1 + NM;                         // rows of a block vector: mRNA, then the 2^NB bit-pattern states
  extern __shared__ __align__(16) double lds[];
  const int N = n.N, S = n.S;
  double* Kt = lds;
  double* Pv = Kt + n.n_K;
  double* red = Pv + 2 * N;
  const int nnzT = n.TF_indptr[N];
  double* tf_dat = red + 24;
  int32_t* tf_idx = reinterpret_cast<int32_t*>(tf_dat + nnzT);
  double* park = tf_dat + nnzT + (nnzT + 1) / 2;     // [kArkSlots][NR][nt]
  const NetSlices sl(n.n_K, N, n.sites);
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int i = tid;
  const bool own = i < N;
  const double* stops = A.stops_p ? A.stops_p : A.stops_v;
  const int32_t* stop_out = A.stop_out_p ? A.stop_out_p : A.stop_out_v;
  const double* xb = A.x + b * n.n_var;
  auto par = [&](int off) { const double v = xb[off]; return A.x_is_raw ? softplus(v) : v; };
  for (int k = tid; k < nnzT; k += nt) { tf_dat[k] = n.TF_data[k]; tf_idx[k] = n.TF_indices[k]; }

  const int st = own ? n.offset_y[i] : 0, ss = own ? n.offset_s[i] : 0, ns = own ? n.n_sites[i] : 0;
  const int nst = 1 << ns;
  const int tf0 = own ? n.TF_indptr[i] : 0, tf1 = own ? n.TF_indptr[i + 1] : 0;
  const double tfdeg_inv = own ? 1.0 / n.tf_deg[i] : 1.0;
  const double Ai = own ? par(sl.A + i) : 0.0, Bi = own ? par(sl.B + i) : 1.0, Ci = own ? par(sl.C + i) : 0.0, Di = own ? par(sl.D + i) : 1.0,
               Ei = own ? par(sl.E + i) : 0.0, ts = par(sl.tf);
  double Dp[NB], Sr[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) { Dp[j] = (j < ns) ? par(sl.Dp + ss + j) : 0.0; Sr[j] = 0.0; }
  const double* y0 = A.y0 + (A.y0_batched ? b * S : 0);
  double* Yout = A.Y + b * (size_t)A.T * S;
  double y[NR];
  y[0] = own ? y0[st] : 0.0;
#pragma unroll
  for (int m = 0; m < NM; ++m) y[1 + m] = (own && m < nst) ? y0[st + 1 + m] : 0.0;
  auto write_row = [&](int row) {
    if (!own) return;
    double* o = Yout + (size_t)row * S + st;
    o[0] = y[0];
#pragma unroll
    for (int m = 0; m < NM; ++m) if (m < nst) o[1 + m] = y[1 + m];
  };
  write_row(0);

  auto set_bucket = [&](const int jb) {
    __syncthreads();
    for (int k = tid; k < n.n_K; k += nt) Kt[k] = n.kin_Kmat[(size_t)k * n.n_grid + jb] * (A.x_is_raw ? softplus(xb[sl.ck + k]) : xb[sl.ck + k]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      double acc = 0.0;
      if (j < ns) for (int q = n.W_indptr[ss + j]; q < n.W_indptr[ss + j + 1]; ++q) acc += n.W_data[q] * Kt[n.W_indices[q]];
      Sr[j] = acc;
    }
  };
  auto loss_of = [&](auto mc) {
    constexpr int m = decltype(mc)::value;
    double l = 0.0;
    static_for<NB>([&](auto jc) { constexpr int j = decltype(jc)::value; if constexpr ((m >> j) & 1) l += Ei + Dp[j] + Di; else l += Sr[j]; });
    if constexpr (m == 0) l += Di;
    return l;
  };
  int buf = 0;
  auto rhs_block = [&](const double (&Y)[NR], double (&f)[NR]) {
    double tot = 0.0;
#pragma unroll
    for (int m = 0; m < NM; ++m) tot += Y[1 + m];                  // the combinatorial RHS ignores driver_map (jacspeedup.py:319-327)
    if (own) Pv[buf * N + i] = tot;
    __syncthreads();
    double acc = 0.0;
    for (int e = tf0; e < tf1; ++e) acc += tf_dat[e] * Pv[buf * N + tf_idx[e]];
    buf ^= 1;
    double v = acc * tfdeg_inv;
    v = v * net_rcp(1.0 + fabs(v));
    f[0] = synth_rate_fast(Ai, ts, v) - Bi * Y[0];
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      double a = (m == 0) ? Ci * Y[0] : 0.0;
      static_for<NB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr ((m >> j) & 1) a = __builtin_fma(Sr[j], Y[1 + (m ^ (1 << j))], a);
        else a = __builtin_fma(Ei, Y[1 + (m | (1 << j))], a);
      });
      f[1 + m] = a - loss_of(mc) * Y[1 + m];
    });
  };
  double winvR = 1.0, gB = 1.0, dinv[NM], dg[NM];
  double sinv[EXACT ? NE : 1][EXACT ? NE : 1];       // EXACT: inverse of the even Schur complement
  // magnitude of W[x][x ^ bit_j] (W = g I - A has it with a minus sign): inflow into x by phosphorylation of site j (x has the bit: rate
  // S_j) or by dephosphorylation (x lacks it: rate E)
  auto wgt = [&](auto xc, auto jc) { constexpr int x = decltype(xc)::value, j = decltype(jc)::value; if constexpr ((x >> j) & 1) return Sr[j]; else return Ei; };
  auto factor = [&](const double g) {
    gB = g + Bi;
    winvR = net_rcp(gB);
    static_for<NM>([&](auto mc) { constexpr int m = decltype(mc)::value; dg[m] = g + loss_of(mc); dinv[m] = net_rcp(dg[m]); });
    if constexpr (EXACT) {
      static_for<NE>([&](auto ac) {
        constexpr int ea = decltype(ac)::value, a = 2 * ea + (__builtin_popcount(ea) & 1);
        static_for<NE>([&](auto bc) {
          constexpr int eb = decltype(bc)::value, b = 2 * eb + (__builtin_popcount(eb) & 1), d = a ^ b;
          double v = 0.0;
          if constexpr (d == 0) {
            v = dg[a];
            static_for<NB>([&](auto jc) {
              constexpr int j = decltype(jc)::value, c = a ^ (1 << j);
              v = __builtin_fma(-(wgt(std::integral_constant<int, a>{}, jc) * wgt(std::integral_constant<int, c>{}, jc)), dinv[c], v);
            });
          } else if constexpr (__builtin_popcount(d) == 2) {
            constexpr int i1 = __builtin_ctz(d), i2 = __builtin_ctz(d & (d - 1)), c1 = a ^ (1 << i1), c2 = a ^ (1 << i2);
            using I = std::integral_constant<int, i1>; using J = std::integral_constant<int, i2>;
            v = -(wgt(std::integral_constant<int, a>{}, I{}) * wgt(std::integral_constant<int, c1>{}, J{}) * dinv[c1] +
                  wgt(std::integral_constant<int, a>{}, J{}) * wgt(std::integral_constant<int, c2>{}, I{}) * dinv[c2]);
          }
          sinv[ea][eb] = v;
        });
      });
      static_for<NE>([&](auto kc) {                    // in-place Gauss-Jordan inverse
        constexpr int k = decltype(kc)::value;
        const double rp = net_rcp(sinv[k][k]);
        static_for<NE>([&](auto ic) {
          constexpr int ii = decltype(ic)::value;
          if constexpr (ii != k) {
            const double ml = sinv[ii][k] * rp;
            static_for<NE>([&](auto jc) { constexpr int jj = decltype(jc)::value; if constexpr (jj != k) sinv[ii][jj] = __builtin_fma(-ml, sinv[k][jj], sinv[ii][jj]); });
            sinv[ii][k] = -ml;
          }
        });
        static_for<NE>([&](auto jc) { constexpr int jj = decltype(jc)::value; if constexpr (jj != k) sinv[k][jj] *= rp; });
        sinv[k][k] = rp;
      });
    }
  };
  // A Y of the block (EXACT: stage 1)
  auto block_matvec = [&](const double (&Yv)[NR], double (&out)[NR]) {
    out[0] = -Bi * Yv[0];
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      double a = (m == 0) ? Ci * Yv[0] : 0.0;
      static_for<NB>([&](auto jc) { constexpr int j = decltype(jc)::value; a = __builtin_fma(wgt(mc, jc), Yv[1 + (m ^ (1 << j))], a); });
      out[1 + m] = a - loss_of(mc) * Yv[1 + m];
    });
  };
  // EXACT: x = (g I - A)^-1 r by parity elimination;  else x = P^-1 r : forward sweep (ascending masks), rescale, backward sweep (descending masks)
  auto block_solve = [&](const double (&r)[NR], double (&x)[NR]) {
    const double xR = r[0] * winvR;
    x[0] = xR;
    if constexpr (EXACT) {
      double re[NE], xe[NE];
      const double r0 = r[1] + Ci * xR;                // the R -> state 0 coupling moved to the right-hand side
      auto rr = [&](auto mc) { constexpr int m = decltype(mc)::value; if constexpr (m == 0) return r0; else return r[1 + m]; };
      static_for<NE>([&](auto ac) {                    // r'_e = r_e - W_eo D_o^-1 r_o
        constexpr int ea = decltype(ac)::value, a = 2 * ea + (__builtin_popcount(ea) & 1);
        double v = rr(std::integral_constant<int, a>{});
        static_for<NB>([&](auto jc) { constexpr int j = decltype(jc)::value, c = a ^ (1 << j); v = __builtin_fma(wgt(std::integral_constant<int, a>{}, jc) * dinv[c], rr(std::integral_constant<int, c>{}), v); });
        re[ea] = v;
      });
      static_for<NE>([&](auto ac) {
        constexpr int ea = decltype(ac)::value, a = 2 * ea + (__builtin_popcount(ea) & 1);
        double v = sinv[ea][0] * re[0];
        static_for<NE - 1>([&](auto bc) { constexpr int eb = 1 + decltype(bc)::value; v = __builtin_fma(sinv[ea][eb], re[eb], v); });
        xe[ea] = v; x[1 + a] = v;
      });
      static_for<NE>([&](auto oc) {                    // x_o = D_o^-1 (r_o - W_oe x_e)
        constexpr int eo = decltype(oc)::value, c = 2 * eo + 1 - (__builtin_popcount(eo) & 1);
        double v = rr(std::integral_constant<int, c>{});
        static_for<NB>([&](auto jc) { constexpr int j = decltype(jc)::value, a = c ^ (1 << j), ea = a >> 1; v = __builtin_fma(wgt(std::integral_constant<int, c>{}, jc), xe[ea], v); });
        x[1 + c] = v * dinv[c];
      });
      return;
    }
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      double a = r[1 + m] + ((m == 0) ? Ci * xR : 0.0);
      static_for<NB>([&](auto jc) { constexpr int j = decltype(jc)::value; if constexpr ((m >> j) & 1) a = __builtin_fma(Sr[j], x[1 + (m ^ (1 << j))], a); });
      x[1 + m] = a * dinv[m];
    });
    static_for<NM>([&](auto mc) {
      constexpr int m = NM - 1 - decltype(mc)::value;
      double a = x[1 + m] * dg[m];
      static_for<NB>([&](auto jc) { constexpr int j = decltype(jc)::value; if constexpr (!((m >> j) & 1)) a = __builtin_fma(Ei, x[1 + (m | (1 << j))], a); });
      x[1 + m] = a * dinv[m];
    });
  };
  // out = P x = (D_g - F) D_g^-1 (D_g - K) x   (stage 1 only)
  auto apply_P = [&](const double (&x)[NR], double (&out)[NR]) {
    double t2[NM];
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      double a = dg[m] * x[1 + m];
      static_for<NB>([&](auto jc) { constexpr int j = decltype(jc)::value; if constexpr (!((m >> j) & 1)) a = __builtin_fma(-Ei, x[1 + (m | (1 << j))], a); });
      t2[m] = a * dinv[m];
    });
    out[0] = gB * x[0];
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      double a = dg[m] * t2[m] - ((m == 0) ? Ci * x[0] : 0.0);
      static_for<NB>([&](auto jc) { constexpr int j = decltype(jc)::value; if constexpr ((m >> j) & 1) a = __builtin_fma(-Sr[j], t2[m ^ (1 << j)], a); });
      out[1 + m] = a;
    });
  };
  double* const mypark = park + tid;
  auto park_ld = [&](int slot, int row) { return mypark[(size_t)(slot * NR + row) * nt]; };
  auto park_st = [&](int slot, int row, double x) { mypark[(size_t)(slot * NR + row) * nt] = x; };

  __syncthreads();
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  double tc = A.t0;
  int jb = net_bucket(tc, n.kin_grid, n.n_grid);
  set_bucket(jb);
  double h;
  {
    double f[NR];
    rhs_block(y, f);
    auto q = [&](double v, double yv) { return fabs(v) / (A.atol + A.rtol * fabs(yv)); };
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int k = 0; k < NR; ++k) if (own && k < 1 + nst) { d0 = fmax(d0, q(y[k], y[k])); d1 = fmax(d1, q(f[k], y[k])); }
    d0 = block_max(d0, red); d1 = block_max(d1, red);
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
    if (A.h0 > 0.0) h = A.h0;
    if (!(h > 0.0) || h != h) h = 1e-6;
  }
  const bool rms = A.err_rms;
  const double safety_inv = 1.0 / (A.ctl_safety > 0.0 ? A.ctl_safety : 0.9), grow_inv = 1.0 / (A.ctl_grow > 1.0 ? A.ctl_grow : 6.0);
  bool after_reject = false;
  for (int si = 0; si < A.n_stops && status == PK_ST_OK; ++si) {
    const double te = stops[si];
    while (true) {
      if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; break; }
      const bool last = (tc + 1.0001 * h >= te);
      const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
      if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; break; }
      const double g = net_rcp(hs * GAM);
      factor(g);
      double Y[NR], w[NR], v[NR], sb[NR], se[NR];
      // ---- stage 1: Y_1 = y_n ;  EXACT: h A y ;  else h A~ y = h (g y - P y)
      rhs_block(y, w);
      if constexpr (EXACT) block_matvec(y, v); else apply_P(y, v);
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        w[k] *= hs; v[k] = EXACT ? hs * v[k] : hs * (g * y[k] - v[k]);
        sb[k] = B[0] * w[k]; se[k] = EB[0] * w[k];
      }
      static_for<4>([&](auto ic) {
        constexpr int ii = decltype(ic)::value;
#pragma unroll
        for (int k = 0; k < NR; ++k) park_st(ii, k, y[k] + AE[ii + 1][0] * w[k] + DI[ii + 1][0] * v[k]);
      });
      static_for<5>([&](auto sc) {
        constexpr int s = 2 + decltype(sc)::value;
        double gr[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          double rk;
          if constexpr (s == 2) rk = y[k] + AE[0][0] * w[k] + DI[0][0] * v[k]; else rk = park_ld(s - 3, k);
          gr[k] = g * rk;
        }
        block_solve(gr, Y);
#pragma unroll
        for (int k = 0; k < NR; ++k) v[k] = __builtin_fma(-hs, gr[k], (1.0 / GAM) * Y[k]);
        rhs_block(Y, w);
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          w[k] *= hs;
          if constexpr (s != 2) { sb[k] = __builtin_fma(B[s - 1], w[k], sb[k]); se[k] = __builtin_fma(EB[s - 1], w[k], se[k]); }
        }
        static_for<4>([&](auto ic) {
          constexpr int ii = decltype(ic)::value;
          if constexpr (ii + 3 > s) {
#pragma unroll
            for (int k = 0; k < NR; ++k) park_st(ii, k, park_ld(ii, k) + AE[ii + 1][s - 1] * w[k] + DI[ii + 1][s - 1] * v[k]);
          }
        });
      });
      auto q = [&](double ev, double ya, double yb) { return fabs(ev) * net_rcp(A.atol + A.rtol * fmax(fabs(ya), fabs(yb))); };
      double e = 0.0;
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        sb[k] += y[k];
        if (own && k < 1 + nst) e = err_acc(e, q(se[k], y[k], sb[k]), rms);
      }
      const double err = err_reduce(e, rms, S, red);
      if (err != err || err > 1e300) {
        ++nrej; after_reject = true; h = 0.1 * hs;
        double bad = (nonfinite(Ai) || nonfinite(Bi) || nonfinite(Ci) || nonfinite(Di) || nonfinite(Ei) || nonfinite(ts)) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < NR; ++k) if (nonfinite(y[k])) bad = 1.0;
#pragma unroll
        for (int j = 0; j < NB; ++j) if (nonfinite(Dp[j]) || nonfinite(Sr[j])) bad = 1.0;
        if (block_max(bad, red) != 0.0) { status |= PK_ST_NONFINITE; break; }
        continue;
      }
      double fac = sqrt(sqrt(err)) * safety_inv;
      fac = fmax(grow_inv, fmin(5.0, fac));
      double hnew = hs * net_rcp(fac);
      if (err <= 1.0) {
        ++nacc;
#pragma unroll
        for (int k = 0; k < NR; ++k) y[k] = sb[k];
        tc += hs;
        if (after_reject) hnew = fmin(hnew, hs);
        after_reject = false;
        if (last) { tc = te; h = (hs < h) ? fmax(hnew, h) : hnew; break; }
        h = hnew;
      } else {
        ++nrej; after_reject = true;
        h = hnew;
      }
    }
    if (status != PK_ST_OK) break;
    const int row = stop_out[si];
    if (row >= 0) write_row(row);
    const int jn = net_bucket(tc, n.kin_grid, n.n_grid);
    if (jn != jb) { jb = jn; set_bucket(jb); }
  }
  if (status != PK_ST_OK && own) {
    const double qnan = __builtin_nan("");
    for (int si = 0; si < A.n_stops; ++si) {
      const int row = stop_out[si];
      if (row >= 0 && !(stops[si] <= tc)) { double* o = Yout + (size_t)row * S + st; for (int k = 0; k < 1 + nst; ++k) o[k] = qnan; }
    }
  }
  if (tid == 0) {
    if (A.status) A.status[b] = status;
    if (A.n_steps) { A.n_steps[2 * b] = nacc; A.n_steps[2 * b + 1] = nrej; }
  }
}

__host__ inline size_t net_solve_ark_lds_bytes(const NetDev& n, int nnzT, int rows, int threads) {
  return ((size_t)n.n_K + 2 * (size_t)n.N + 24 + nnzT + (nnzT + 1) / 2 + net_ark_park_doubles(rows, threads)) * 8;
}

}  // namespace pk
