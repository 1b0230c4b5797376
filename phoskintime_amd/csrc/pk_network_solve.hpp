// pk_network_solve.hpp -- batched integration of the network ODE: replaces global_model.simulate.simulate_odeint
// (simulate.py:34-80: odeint(rhs_odeint, y0, t, Dfun = fd_jacobian_odeint, rtol, atol, mxstep)) for B candidates.
//
// One workgroup per candidate; state, stage vectors and per-block factors in LDS.
//
// Integrator: ROS34PW2 (Rang & Angermann 2005), a 4-stage, order-3, stiffly accurate, L-stable Rosenbrock-W method --
// its order conditions hold for ANY approximation of the Jacobian (table verified in 50-digit arithmetic,
// tools/check_ros34pw2.py).  That licence is used to keep only the per-protein diagonal blocks of J in the linear
// systems (arrow blocks for the distributive / saturating topologies, tridiagonal blocks for the sequential one): all
// stiffness of this system lives inside those blocks (phosphorylation / de-phosphorylation / decay), while the coupling
// between proteins -- transcription-factor input to the mRNA rows -- is slow and bounded; the numpy model
// (tools/proto_rosw_network.py) shows identical step counts with the full and the block-diagonal Jacobian.  So a "linear
// solve" is N independent tiny block solves (one thread per protein) instead of an S x S factorisation
// (the reference hands LSODA a dense finite-difference Jacobian: S + 1 right-hand sides per refresh).
//
// The kinase forcing is piecewise constant (jacspeedup.py:149-172): a step never straddles a bucket edge (every edge is a
// forced landing point) and uses the bucket of its START time throughout, so each step sees an autonomous system.
#pragma once
#include "pk_network.hpp"
#include "../../include/phoskin.h"

namespace pk {

namespace rosw {
constexpr double GAM = 0.435866521508459;
constexpr double A21 = 2.0, A31 = 1.41921731745576465, A32 = -0.25923221167296971378;
constexpr double A41 = 4.1847604823191607312, A42 = -0.2851920173554959137, A43 = 2.2942803602790417167;
constexpr double C21 = -4.5885607205580834861, C31 = -4.1847604823191607312, C32 = 0.2851920173554959137;
constexpr double C41 = -6.3681792001283577635, C42 = -6.7956209444668361844, C43 = 2.8700986043310560892;
// y1 = Y4 + U4 (stiffly accurate: m = (A41, A42, A43, 1)); error estimate = sum E_i U_i
constexpr double E1 = 0.27774994764796811038, E2 = -1.4032398951759990242, E3 = 1.7726301276675507452, E4 = 0.5;
__device__ constexpr double TA[4][3] = {{0, 0, 0}, {A21, 0, 0}, {A31, A32, 0}, {A41, A42, A43}};
__device__ constexpr double TC[4][3] = {{0, 0, 0}, {C21, 0, 0}, {C31, C32, 0}, {C41, C42, C43}};
}  // namespace rosw

struct NetSolveArgs {
  const double* x; int x_is_raw;
  const double* y0; int y0_batched;
  // landing times (output times + kinase-bucket edges) and the output row each one fills (-1: none); passed by value when short
  double stops_v[64]; int32_t stop_out_v[64];
  const double* stops_p; const int32_t* stop_out_p; int n_stops;
  double t0; int T;
  double* Y;                       // [B, T, S]
  int32_t* status; int32_t* n_steps;
  double rtol, atol, h0; int max_steps;
  double ctl_safety, ctl_grow;     // step-size controller of the additive kernels: h_new = h * min(ctl_grow, ctl_safety / err^(1/4)) (0: defaults 0.9, 6)
  int err_rms;                     // 1: ODEPACK's weighted root-mean-square error norm (what the reference's LSODA controls); 0: max norm
  // fused objective (pk_network_simulate_objective_batch; the pair kernel on the register diet): dense observation tables [T, S] of the
  // loss handle, read at every output time; the trajectory itself is written only if Y != null.  loss_obs == null: plain simulate
  const double* loss_obs; const double* loss_w; const double* loss_defaults;
  double loss_lam[4], loss_norm[3], loss_fail; int loss_mode, loss_rna_base;
  double* loss_sums; double* loss_F;
};

// block-wide NaN-propagating max; `red` holds >= 17 doubles of LDS
__device__ __forceinline__ double block_max(double v, double* red) {
  auto mx = [](double a, double b) { return (a > b || a != a) ? a : b; };
  for (int off = 32; off > 0; off >>= 1) v = mx(v, __shfl_xor(v, off));
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double r = red[0];
  for (int i = 1; i < nw; ++i) r = mx(r, red[i]);
  return r;
}

// block-wide sum (NaN / inf propagate by themselves); `red` holds >= 17 doubles of LDS
__device__ __forceinline__ double block_sum(double v, double* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double r = red[0];
  for (int i = 1; i < nw; ++i) r += red[i];
  return r;
}
// local error of a step from the per-thread partial value `e`: max over all states, or sqrt(mean of squares) (ODEPACK vnorm)
__device__ __forceinline__ double err_reduce(double e, const bool rms, const int S, double* red) {
  return rms ? sqrt(block_sum(e, red) / S) : block_max(e, red);
}
__device__ __forceinline__ double err_acc(double a, double q, const bool rms) {
  return rms ? __builtin_fma(q, q, a) : ((q > a || q != q) ? q : a);
}

template <int MODEL>
__global__ __launch_bounds__(256, 3) void net_solve_kernel(const NetDev n, const NetSolveArgs A) {
  using namespace rosw;
  constexpr int model = MODEL;
  extern __shared__ __align__(16) double lds[];
  NetLds L(lds, n);
  double* base = lds + NetLds::doubles(n);
  const int S = n.S, N = n.N;
  double* y = L.y;                        // current state
  double* Ys = base;                      // stage point
  double* U1 = Ys + S; double* U2 = U1 + S; double* U3 = U2 + S; double* U4 = U3 + S;
  double* R_ = U4 + S;                    // right-hand side of the stage system
  double* winv = R_ + S;                  // per state: 1 / pivot of its row in the block factorisation
  double* sinv = winv + S;                // per protein: 1 / Schur pivot of the P row (arrow blocks)
  double* cR = sinv + N;                  // per protein: d f_P / d R
  double* gP = cR + N;                    // per protein: saturating-kinetics factor 1 / (1 + P)^2 (1 otherwise)
  double* red = gP + N;                   // 24 doubles: reductions
  // static topology the inner loop touches, cached in LDS: TF CSR (data, degree, indptr, indices)
  const int nnzT = n.TF_indptr[N];
  double* tf_dat = red + 24;
  double* tf_degl = tf_dat + nnzT;
  int32_t* tf_ptr = reinterpret_cast<int32_t*>(tf_degl + N);
  int32_t* tf_idx = tf_ptr + (N + 1);
  const NetSlices sl(n.n_K, N, n.sites);
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const double* stops = A.stops_p ? A.stops_p : A.stops_v;
  const int32_t* stop_out = A.stop_out_p ? A.stop_out_p : A.stop_out_v;

  for (int k = tid; k < nnzT; k += nt) { tf_dat[k] = n.TF_data[k]; tf_idx[k] = n.TF_indices[k]; }
  for (int k = tid; k <= N; k += nt) tf_ptr[k] = n.TF_indptr[k];
  for (int k = tid; k < N; k += nt) tf_degl[k] = n.tf_deg[k];
  // per-thread contexts in registers: up to KS states and KP proteins per thread (host guarantees S <= KS*nt, N <= KP*nt)
  constexpr int KS = 4, KP = 2;
  int s_i[KS], s_loc[KS], s_st[KS], s_ss[KS], s_ns[KS];
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    const int k = tid + q * nt;
    if (k < S) { const int i = n.state_prot[k]; s_i[q] = i; s_loc[q] = n.state_local[k]; s_st[q] = n.offset_y[i]; s_ss[q] = n.offset_s[i]; s_ns[q] = n.n_sites[i]; }
    else { s_i[q] = 0; s_loc[q] = 0; s_st[q] = 0; s_ss[q] = 0; s_ns[q] = 0; }
  }
  int p_st[KP], p_ss[KP], p_ns[KP], p_drv[KP];
#pragma unroll
  for (int q = 0; q < KP; ++q) {
    const int i = tid + q * nt;
    if (i < N) { p_st[q] = n.offset_y[i]; p_ss[q] = n.offset_s[i]; p_ns[q] = n.n_sites[i]; p_drv[q] = n.driver_map[i]; }
    else { p_st[q] = 0; p_ss[q] = 0; p_ns[q] = 0; p_drv[q] = -1; }
  }
  const double* xb = A.x + b * n.n_var;
  for (int k = tid; k < n.n_var; k += nt) L.p[k] = A.x_is_raw ? softplus(xb[k]) : xb[k];
  const double* y0 = A.y0 + (A.y0_batched ? b * S : 0);
  double* Yout = A.Y + b * (size_t)A.T * S;
  for (int k = tid; k < S; k += nt) { const double v = y0[k]; y[k] = v; Yout[k] = v; }
  __syncthreads();

  // ---- block factorisation of  g I - J_blockdiag(y)  and block solve  x <- W^{-1} r  (in place: r -> x), one thread per protein
  auto factor = [&](const double g) {
#pragma unroll
    for (int q_ = 0; q_ < KP; ++q_) {
      const int i = tid + q_ * nt;
      if (i >= N) break;
      const int st = p_st[q_], ss = p_ss[q_], ns = p_ns[q_];
      const double Bi = L.p[sl.B + i], Ci = L.p[sl.C + i], Di = L.p[sl.D + i], Ei = L.p[sl.E + i];
      const double* Dp = L.p + sl.Dp + ss;
      const double* Sr = L.Sall + ss;
      winv[st] = 1.0 / (g + Bi);
      if (model == 2) {
        // combinatorial block of any size (2^ns bit-pattern states): -diag(loss) + F (phosphorylation, strictly lower in mask order) +
        // K (dephosphorylation, strictly upper).  W-method licence once more: g I - J_block ~= (D_g - F) D_g^-1 (D_g - K), so only the
        // pivots 1 / (g + loss_m) are stored; the two sweeps run in block_solve (same scheme as pk_network_solve_reg2.hpp, any ns)
        cR[i] = Ci; gP[i] = 1.0;
        const int nst = 1 << ns;
        for (int m = 0; m < nst; ++m) {
          double loss = (m == 0) ? Di : 0.0;
          for (int j = 0; j < ns; ++j) loss += ((m >> j) & 1) ? (Ei + Dp[j] + Di) : Sr[j];
          winv[st + 1 + m] = 1.0 / (g + loss);
        }
      } else if (model == 1) {
        // tridiagonal over P0, P1..Pns: Thomas pivots
        cR[i] = Ci; gP[i] = 1.0;
        double d = g + Di + (ns ? Sr[0] : 0.0);
        winv[st + 1] = 1.0 / d;
        for (int q = 1; q <= ns; ++q) {
          const int j = q - 1;
          const double diag = g + Ei + Dp[j] + Di + ((j < ns - 1) ? Sr[j + 1] : 0.0);
          d = diag - (Sr[j] * Ei) * winv[st + q];        // lower entry -k_j, upper entry of the row above -E
          winv[st + 1 + q] = 1.0 / d;
        }
      } else {
        const bool sat = model == 4;
        const double Rv = y[st], Pv = y[st + 1];
        const double g_p = sat ? 1.0 / ((1.0 + Pv) * (1.0 + Pv)) : 1.0;
        cR[i] = sat ? Ci / ((1.0 + Rv) * (1.0 + Rv)) : Ci;
        gP[i] = g_p;
        double sumS = 0.0, acc = 0.0;
        for (int j = 0; j < ns; ++j) {
          const double wj = 1.0 / (g + Ei + Dp[j] + Di);
          winv[st + 2 + j] = wj;
          sumS += Sr[j];
          acc += Ei * (Sr[j] * g_p) * wj;
        }
        sinv[i] = 1.0 / (g + Di + sumS * g_p - acc);
      }
    }
    __syncthreads();
  };
  auto block_solve = [&](const double* r, double* x) {
#pragma unroll
    for (int q_ = 0; q_ < KP; ++q_) {
      const int i = tid + q_ * nt;
      if (i >= N) break;
      const int st = p_st[q_], ss = p_ss[q_], ns = p_ns[q_];
      const double Ei = L.p[sl.E + i];
      const double* Sr = L.Sall + ss;
      const double xR = r[st] * winv[st];
      x[st] = xR;
      if (model == 2) {
        const int nst = 1 << ns;
        for (int m = 0; m < nst; ++m) {                              // (D_g - F) u = r : ascending masks
          double a = r[st + 1 + m] + ((m == 0) ? cR[i] * xR : 0.0);
          for (int mm = m; mm; mm &= mm - 1) { const int bit = mm & -mm; a = __builtin_fma(Sr[__builtin_ctz(bit)], x[st + 1 + (m ^ bit)], a); }
          x[st + 1 + m] = a * winv[st + 1 + m];
        }
        for (int m = nst - 2; m >= 0; --m) {                         // (D_g - K) x = D_g u : descending masks
          double hi = 0.0;
          for (int mm = ~m & (nst - 1); mm; mm &= mm - 1) hi += x[st + 1 + (m | (mm & -mm))];
          x[st + 1 + m] = __builtin_fma(Ei * hi, winv[st + 1 + m], x[st + 1 + m]);
        }
      } else if (model == 1) {
        // Thomas: forward sweep (lower entries -k_{q-1}), back substitution (upper entries -E); x doubles as work space
        double prev = r[st + 1] + cR[i] * xR;
        x[st + 1] = prev;
        for (int q = 1; q <= ns; ++q) { prev = r[st + 1 + q] + Sr[q - 1] * prev * winv[st + q]; x[st + 1 + q] = prev; }
        double xn = x[st + 1 + ns] * winv[st + 1 + ns];
        x[st + 1 + ns] = xn;
        for (int q = ns - 1; q >= 0; --q) { xn = (x[st + 1 + q] + Ei * xn) * winv[st + 1 + q]; x[st + 1 + q] = xn; }
      } else {
        const double g_p = gP[i];
        double acc = 0.0;
        for (int j = 0; j < ns; ++j) { const double t = r[st + 2 + j] * winv[st + 2 + j]; x[st + 2 + j] = t; acc += Ei * t; }
        const double xP = (r[st + 1] + cR[i] * xR + acc) * sinv[i];
        x[st + 1] = xP;
        for (int j = 0; j < ns; ++j) x[st + 2 + j] += (Sr[j] * g_p) * winv[st + 2 + j] * xP;
      }
    }
    __syncthreads();
  };
  // P_vec -> TF input -> synthesis rate for the state L.y points at (net_prepare_state with the LDS-cached topology)
  auto prepare_state = [&]() {
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      const int i = tid + q * nt;
      if (i >= N) break;
      double tot;
      if (model != 2 && p_drv[q] >= 0) tot = L.Kt[p_drv[q]];        // the combinatorial RHS ignores driver_map (jacspeedup.py:319-327)
      else { tot = 0.0; const int cnt = (model == 2) ? (1 << p_ns[q]) : 1 + p_ns[q]; for (int m_ = 0; m_ < cnt; ++m_) tot += L.y[p_st[q] + 1 + m_]; }
      L.Pvec[i] = tot;
    }
    __syncthreads();
    const double ts = L.p[sl.tf];
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      const int i = tid + q * nt;
      if (i >= N) break;
      double acc = 0.0;
      for (int e_ = tf_ptr[i]; e_ < tf_ptr[i + 1]; ++e_) acc += tf_dat[e_] * L.Pvec[tf_idx[e_]];
      double v = acc / tf_degl[i];
      if (model != 4) v = v / (1.0 + fabs(v));
      L.synth[i] = synth_rate(L.p[sl.A + i], ts, v, nullptr);
    }
    __syncthreads();
  };
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  double tc = A.t0;
  int jb = net_bucket(tc, n.kin_grid, n.n_grid);
  net_prepare_bucket(n, L, jb);
  double h;
  {
    // first step from the max-norm of y / sc and f / sc
    L.y = y;
    prepare_state();
    double d0 = 0.0, d1 = 0.0;
    for (int k = tid; k < S; k += nt) {
      const double sc = A.atol + A.rtol * fabs(y[k]);
      d0 = fmax(d0, fabs(y[k]) / sc); d1 = fmax(d1, fabs(net_state_rhs(n, L, k)) / sc);
    }
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
      const double hinv = 1.0 / hs;
      factor(hinv * (1.0 / GAM));
      // four stages, one loop body (kept rolled: the body is large and register-hungry when replicated)
      double* const Us[4] = {U1, U2, U3, U4};
#pragma unroll 1
      for (int sg = 0; sg < 4; ++sg) {
        if (sg > 0) {
          for (int k = tid; k < S; k += nt) {
            double v = y[k];
            for (int j = 0; j < sg; ++j) v = __builtin_fma(TA[sg][j], Us[j][k], v);
            Ys[k] = v;
          }
          __syncthreads();
        }
        L.y = (sg == 0) ? y : Ys;
        prepare_state();
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          const int k = tid + q * nt;
          if (k >= S) break;
          double v = net_state_rhs_ctx<MODEL>(n, L, s_i[q], s_loc[q], s_st[q], s_ss[q], s_ns[q]);
          for (int j = 0; j < sg; ++j) v = __builtin_fma(TC[sg][j] * hinv, Us[j][k], v);
          R_[k] = v;
        }
        __syncthreads();
        block_solve(R_, Us[sg]);
      }
      // y1 = Ys + U4 ; err
      double e = 0.0;
      for (int k = tid; k < S; k += nt) {
        const double yn = Ys[k] + U4[k];
        const double ev = E1 * U1[k] + E2 * U2[k] + E3 * U3[k] + E4 * U4[k];
        const double q = fabs(ev) / (A.atol + A.rtol * fmax(fabs(y[k]), fabs(yn)));
        e = err_acc(e, q, A.err_rms);
        R_[k] = yn;
      }
      const double err = err_reduce(e, A.err_rms, S, red);
      if (err != err || err > 1e300) {
        ++nrej; after_reject = true; h = 0.1 * hs;
        double bad = 0.0;
        for (int k = tid; k < S; k += nt) if (nonfinite(y[k])) bad = 1.0;
        for (int k = tid; k < n.n_var; k += nt) if (nonfinite(L.p[k])) bad = 1.0;
        if (block_max(bad, red) != 0.0) { status |= PK_ST_NONFINITE; break; }
        continue;
      }
      double fac = cbrt(err) * (1.0 / 0.9);
      fac = fmax(1.0 / 6.0, fmin(5.0, fac));
      double hnew = hs / fac;
      if (err <= 1.0) {
        ++nacc;
        for (int k = tid; k < S; k += nt) y[k] = R_[k];
        __syncthreads();
        tc += hs;
        if (after_reject) hnew = fmin(hnew, hs);
        after_reject = false;
        if (last) {
          tc = te;
          h = (hs < h) ? fmax(hnew, h) : hnew;
          break;
        }
        h = hnew;
      } else {
        ++nrej; after_reject = true;
        h = hnew;
      }
    }
    if (status != PK_ST_OK) break;
    const int row = stop_out[si];
    if (row >= 0) for (int k = tid; k < S; k += nt) Yout[(size_t)row * S + k] = y[k];
    const int jn = net_bucket(tc, n.kin_grid, n.n_grid);
    if (jn != jb) { jb = jn; net_prepare_bucket(n, L, jb); }
  }
  if (status != PK_ST_OK) {
    // flagged candidate: every output row that was not reached is NaN (never garbage)
    const double qnan = __builtin_nan("");
    for (int si = 0; si < A.n_stops; ++si) {
      const int row = stop_out[si];
      if (row >= 0 && !(stops[si] <= tc)) for (int k = tid; k < S; k += nt) Yout[(size_t)row * S + k] = qnan;
    }
  }
  if (tid == 0) {
    if (A.status) A.status[b] = status;
    if (A.n_steps) { A.n_steps[2 * b] = nacc; A.n_steps[2 * b + 1] = nrej; }
  }
}

__device__ __host__ inline size_t net_solve_lds_doubles(const NetDev& n) {
  return ((size_t)n.n_var + n.S + n.n_K + n.sites + 3 * (size_t)n.N) + 7 * (size_t)n.S + 3 * (size_t)n.N + 24;
}
// + the LDS copy of the TF CSR: nnz doubles + N doubles + (N + 1 + nnz) int32 (rounded up to doubles)
__host__ inline size_t net_solve_lds_bytes(const NetDev& n, int nnzT) {
  return (net_solve_lds_doubles(n) + (size_t)nnzT + n.N) * 8 + (((size_t)n.N + 1 + nnzT) * 4 + 7) / 8 * 8;
}

}  // namespace pk
