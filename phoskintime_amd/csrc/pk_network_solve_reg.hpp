// pk_network_solve_reg.hpp -- register-resident variant of the network integrator (pk_network_solve.hpp): ONE THREAD PER PROTEIN.
//
// Profile of the LDS variant at S = 500 (profiles/r01_c_*): 65 % of all wave-cycles are spent parked at barriers / waitcnt --
// ~25 workgroup barriers per step separate phases that are really per-protein work.  Here a thread owns one protein's whole block
// (mRNA, protein, <= MAXS phospho states): its state, the four ROS34PW2 stage vectors, its parameters, site rates and block
// factors all live in VGPRs, and the block solve (arrow elimination / Thomas) is thread-local.  The only things that cross threads are
//   * P_vec (one double per protein) -> TF input of the mRNA rows: ONE barrier per stage (double-buffered in LDS),
//   * the max-norm of the error estimate: one block reduction per step,
//   * Kt, the scaled kinase input of the current bucket (LDS, rewritten only at bucket edges).
// Same method, same Jacobian approximation, same step control as the LDS kernel => same results up to summation order.
// Eligibility: every protein has <= MAXS sites, N <= 256, topology 0 / 1 / 4.  Otherwise the LDS kernel is used.
// Padding entries (site slots j >= n_sites) hold exact zeros in every vector and in Sr, which makes them inert in all formulas.
#pragma once
#include "pk_network_solve.hpp"
#include "pk_wave.hpp"

namespace pk {

template <int MODEL, int MAXS>
__global__ __launch_bounds__(256, 2) void net_solve_reg_kernel(const NetDev n, const NetSolveArgs A) {
  using namespace rosw;
  extern __shared__ __align__(16) double lds[];
  const int N = n.N, S = n.S;
  double* Kt = lds;                       // [n_K]
  double* Pv = Kt + n.n_K;                // [2][N]
  double* red = Pv + 2 * N;               // [24]
  const int nnzT = n.TF_indptr[N];
  double* tf_dat = red + 24;              // [nnzT]
  int32_t* tf_idx = reinterpret_cast<int32_t*>(tf_dat + nnzT);
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

  // ---- this thread's protein
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
  double yR = own ? y0[st] : 0.0, yP = own ? y0[st + 1] : 0.0, ys[MAXS];
#pragma unroll
  for (int j = 0; j < MAXS; ++j) ys[j] = (j < ns) ? y0[st + 2 + j] : 0.0;
  auto write_row = [&](int row) {
    if (!own) return;
    double* o = Yout + (size_t)row * S + st;
    o[0] = yR; o[1] = yP;
#pragma unroll
    for (int j = 0; j < MAXS; ++j) if (j < ns) o[2 + j] = ys[j];
  };
  write_row(0);

  // ---- per-bucket: Kt (shared) and this protein's site rates S_all = W . Kt
  auto set_bucket = [&](const int jb) {
    __syncthreads();                                         // nobody may still be reading the old Kt
    for (int k = tid; k < n.n_K; k += nt) Kt[k] = n.kin_Kmat[(size_t)k * n.n_grid + jb] * (A.x_is_raw ? softplus(xb[sl.ck + k]) : xb[sl.ck + k]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MAXS; ++j) {
      double acc = 0.0;
      if (j < ns) for (int q = n.W_indptr[ss + j]; q < n.W_indptr[ss + j + 1]; ++q) acc += n.W_data[q] * Kt[n.W_indices[q]];
      Sr[j] = acc;
    }
  };

  // ---- f(Y) for the whole block.  buf selects the P_vec buffer; ends after ONE barrier.
  int buf = 0;
  auto rhs_block = [&](const double YR, const double YP, const double (&Ysv)[MAXS], double& fR, double& fP, double (&fs)[MAXS]) {
    double tot;
    if (drv >= 0) tot = Kt[drv];
    else { tot = YP;
#pragma unroll
      for (int j = 0; j < MAXS; ++j) tot += Ysv[j]; }         // padding entries are zero
    if (own) Pv[buf * N + i] = tot;
    __syncthreads();
    double acc = 0.0;
    for (int e = tf0; e < tf1; ++e) acc += tf_dat[e] * Pv[buf * N + tf_idx[e]];
    buf ^= 1;
    double v = acc * tfdeg_inv;
    if (MODEL != 4) v = v * net_rcp(1.0 + fabs(v));
    fR = synth_rate_fast(Ai, ts, v) - Bi * YR;
    if (MODEL == 0) {
      double sumS = 0.0, back = 0.0;
#pragma unroll
      for (int j = 0; j < MAXS; ++j) { sumS += Sr[j]; back += Ei * Ysv[j]; fs[j] = Sr[j] * YP - (Ei + Dp[j] + Di) * Ysv[j]; }
      fP = Ci * YR - (Di + sumS) * YP + back;
    } else if (MODEL == 4) {
      const double q = YP * net_rcp(1.0 + YP);
      double f = 0.0, back = 0.0;
#pragma unroll
      for (int j = 0; j < MAXS; ++j) { const double fw = Sr[j] * q; f += fw; back += Ei * Ysv[j]; fs[j] = fw - (Dp[j] + Di) * Ysv[j] - Ei * Ysv[j]; }
      fP = (Ci * YR) * net_rcp(1.0 + YR) - Di * YP - f + back;
    } else {
      // sequential chain P0 -> P1 -> ... -> Pns (models.py:216-306); zero padding makes the "last level" case automatic
      fP = Ci * YR - Di * YP - Sr[0] * YP + Ei * Ysv[0];
      static_for<MAXS>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        double prev, next = 0.0, knext = 0.0;
        if constexpr (j == 0) prev = YP; else prev = Ysv[j - 1];
        if constexpr (j + 1 < MAXS) { next = Ysv[j + 1]; knext = Sr[j + 1]; }
        fs[j] = Sr[j] * prev + Ei * next - (knext + Ei + Dp[j] + Di) * Ysv[j];
      });
    }
  };

  // ---- block factors of g I - J_blockdiag(y) and the thread-local block solve
  double winvR = 1.0, sinv = 1.0, cRv = 0.0, gPv = 1.0, wv[MAXS + 1];
  auto factor = [&](const double g) {
    winvR = net_rcp(g + Bi);
    if (MODEL == 1) {
      // Thomas pivots over P0, P1..P_MAXS (rows beyond n_sites decouple by themselves: their Sr is 0)
      cRv = Ci; gPv = 1.0;
      double d = g + Di + Sr[0];
      wv[0] = net_rcp(d);
      static_for<MAXS>([&](auto jc) {
        constexpr int j = decltype(jc)::value;            // row q = j + 1
        double knext = 0.0;
        if constexpr (j + 1 < MAXS) knext = Sr[j + 1];
        d = (g + Ei + Dp[j] + Di + knext) - (Sr[j] * Ei) * wv[j];
        wv[j + 1] = net_rcp(d);
      });
    } else {
      const bool sat = MODEL == 4;
      gPv = sat ? net_rcp((1.0 + yP) * (1.0 + yP)) : 1.0;
      cRv = sat ? Ci * net_rcp((1.0 + yR) * (1.0 + yR)) : Ci;
      double sumS = 0.0, acc = 0.0;
#pragma unroll
      for (int j = 0; j < MAXS; ++j) {
        const double w = net_rcp(g + Ei + Dp[j] + Di);
        wv[j] = w; sumS += Sr[j]; acc += Ei * (Sr[j] * gPv) * w;       // Sr is 0 on padding entries
      }
      sinv = net_rcp(g + Di + sumS * gPv - acc);
    }
  };
  auto block_solve = [&](const double rR, const double rP, const double (&rs)[MAXS], double& xR, double& xP, double (&xs)[MAXS]) {
    xR = rR * winvR;
    if (MODEL == 1) {
      double fw[MAXS + 1];
      fw[0] = rP + cRv * xR;
      static_for<MAXS>([&](auto jc) { constexpr int j = decltype(jc)::value; fw[j + 1] = rs[j] + Sr[j] * fw[j] * wv[j]; });
      double xn = 0.0;                                    // x_{MAXS+1} = 0; zero padding keeps every x beyond n_sites at 0
      static_for<MAXS>([&](auto jc) { constexpr int q = MAXS - decltype(jc)::value; xn = (fw[q] + Ei * xn) * wv[q]; xs[q - 1] = xn; });
      xP = (fw[0] + Ei * xn) * wv[0];
    } else {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < MAXS; ++j) { const double t = rs[j] * wv[j]; xs[j] = t; acc += Ei * t; }
      xP = (rP + cRv * xR + acc) * sinv;
#pragma unroll
      for (int j = 0; j < MAXS; ++j) xs[j] += (Sr[j] * gPv) * wv[j] * xP;
    }
  };

  __syncthreads();
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  double tc = A.t0;
  int jb = net_bucket(tc, n.kin_grid, n.n_grid);
  set_bucket(jb);
  double UR[4], UP[4], Us[4][MAXS];
  double h;
  {
    double fR, fP, fs[MAXS];
    rhs_block(yR, yP, ys, fR, fP, fs);
    auto q = [&](double v, double yv) { return fabs(v) / (A.atol + A.rtol * fabs(yv)); };
    double d0 = own ? fmax(q(yR, yR), q(yP, yP)) : 0.0, d1 = own ? fmax(q(fR, yR), q(fP, yP)) : 0.0;
#pragma unroll
    for (int j = 0; j < MAXS; ++j) if (j < ns) { d0 = fmax(d0, q(ys[j], ys[j])); d1 = fmax(d1, q(fs[j], ys[j])); }
    d0 = block_max(d0, red); d1 = block_max(d1, red);
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
    if (A.h0 > 0.0) h = A.h0;
    if (!(h > 0.0) || h != h) h = 1e-6;
  }
  bool after_reject = false;
  for (int si = 0; si < A.n_stops && status == PK_ST_OK; ++si) {
    const double te = stops[si];
    while (true) {
      if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; break; }
      const bool last = (tc + 1.0001 * h >= te);
      const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
      if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; break; }
      const double hinv = net_rcp(hs);
      factor(hinv * (1.0 / GAM));
      double YR = yR, YP = yP, Ysv[MAXS];
#pragma unroll
      for (int j = 0; j < MAXS; ++j) Ysv[j] = ys[j];
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        if (sg > 0) {
          YR = yR; YP = yP;
#pragma unroll
          for (int j = 0; j < MAXS; ++j) Ysv[j] = ys[j];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            if (u < sg) {
              const double a = (sg == 1) ? A21 : (sg == 2 ? (u == 0 ? A31 : A32) : (u == 0 ? A41 : (u == 1 ? A42 : A43)));
              YR = __builtin_fma(a, UR[u], YR); YP = __builtin_fma(a, UP[u], YP);
#pragma unroll
              for (int j = 0; j < MAXS; ++j) Ysv[j] = __builtin_fma(a, Us[u][j], Ysv[j]);
            }
          }
        }
        double fR, fP, fs[MAXS];
        rhs_block(YR, YP, Ysv, fR, fP, fs);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          if (u < sg) {
            const double c = ((sg == 1) ? C21 : (sg == 2 ? (u == 0 ? C31 : C32) : (u == 0 ? C41 : (u == 1 ? C42 : C43)))) * hinv;
            fR = __builtin_fma(c, UR[u], fR); fP = __builtin_fma(c, UP[u], fP);
#pragma unroll
            for (int j = 0; j < MAXS; ++j) fs[j] = __builtin_fma(c, Us[u][j], fs[j]);
          }
        }
        block_solve(fR, fP, fs, UR[sg], UP[sg], Us[sg]);
      }
      // y1 = Y4 + U4 ; err = sum E_i U_i
      const double nR = YR + UR[3], nP = YP + UP[3];
      double nS[MAXS];
      auto q = [&](double ev, double ya, double yb) { return fabs(ev) * net_rcp(A.atol + A.rtol * fmax(fabs(ya), fabs(yb))); };
      auto mx = [](double a, double c) { return (a > c || a != a) ? a : c; };
      const bool rms = A.err_rms;
      double e = 0.0;
      if (own) {
        e = err_acc(err_acc(0.0, q(E1 * UR[0] + E2 * UR[1] + E3 * UR[2] + E4 * UR[3], yR, nR), rms), q(E1 * UP[0] + E2 * UP[1] + E3 * UP[2] + E4 * UP[3], yP, nP), rms);
      }
#pragma unroll
      for (int j = 0; j < MAXS; ++j) {
        nS[j] = Ysv[j] + Us[3][j];
        if (j < ns) e = err_acc(e, q(E1 * Us[0][j] + E2 * Us[1][j] + E3 * Us[2][j] + E4 * Us[3][j], ys[j], nS[j]), rms);
      }
      const double err = err_reduce(e, rms, S, red);
      if (err != err || err > 1e300) {
        ++nrej; after_reject = true; h = 0.1 * hs;
        double bad = (nonfinite(yR) || nonfinite(yP) || nonfinite(Ai) || nonfinite(Bi) || nonfinite(Ci) || nonfinite(Di) || nonfinite(Ei) ||
                      nonfinite(ts)) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < MAXS; ++j) if (nonfinite(ys[j]) || nonfinite(Dp[j]) || nonfinite(Sr[j])) bad = 1.0;
        if (block_max(bad, red) != 0.0) { status |= PK_ST_NONFINITE; break; }
        continue;
      }
      double fac = cbrt(err) * (1.0 / 0.9);
      fac = fmax(1.0 / 6.0, fmin(5.0, fac));
      double hnew = hs * net_rcp(fac);
      if (err <= 1.0) {
        ++nacc;
        yR = nR; yP = nP;
#pragma unroll
        for (int j = 0; j < MAXS; ++j) ys[j] = nS[j];
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

__host__ inline size_t net_solve_reg_lds_bytes(const NetDev& n, int nnzT) {
  return ((size_t)n.n_K + 2 * (size_t)n.N + 24 + nnzT) * 8 + ((size_t)nnzT * 4 + 7) / 8 * 8;
}

}  // namespace pk
