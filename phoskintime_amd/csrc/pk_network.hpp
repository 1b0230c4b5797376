// pk_network.hpp -- the global_model NETWORK right-hand side and analytic Jacobian, one workgroup per candidate.
//
// Reference: global_model/jacspeedup.py:176-388 (rhs_nb_distributive / _sequential / _combinatorial / _saturating) on top of
// global_model/models.py (per-protein kernels), global_model/utils.py:211-225 (time_bucket), params.py:106-132 (softplus unpack).
//
//   Kt      = kin_Kmat[:, bucket(t)] * c_k                       (piecewise-constant kinase input)
//   S_all   = W (CSR, sites x kinases) . Kt                       (phosphorylation rate of every site)
//   P_vec_i = Kt[driver_map[i]] if protein i is a driven kinase / proxy, else P_i + sum of its phospho states
//   v_i     = (TF (CSR, N x N) . P_vec)_i / tf_deg_i ; models 0/1/2 squash once here: v / (1 + |v|)
//   synth_i = calculate_synthesis_rate(A_i, tf_scale, v_i)        (squashes again: the reference's double squash is kept)
//   then the per-protein kinetic block (distributive / sequential / combinatorial / saturating).
//
// Data model: the static topology lives in HBM once per network (NetDev); a candidate is ONE row of the row-major
// [B, n_var] matrix x = [c_k (n_K) | A (N) | B (N) | C (N) | D (N) | Dp (sites) | E (N) | tf_scale], the optimiser's decision
// vector (params.py:60-96), raw (softplus applied on load) or physical.  The workgroup stages x, y, Kt, S_all, P_vec and
// synth in LDS; stages are separated by workgroup barriers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "pk_wave.hpp"

namespace pk {

struct NetDev {
  int model, N, n_K, sites, S, n_grid, n_var;
  const int32_t *offset_y, *offset_s, *n_sites, *state_prot, *state_local;
  const int32_t *W_indptr, *W_indices; const double* W_data;
  const int32_t *TF_indptr, *TF_indices; const double* TF_data;
  const double* tf_deg; const int32_t* driver_map; const double* kin_grid; const double* kin_Kmat;
  // dense lane layout of the additive integrator (pk_network_solve_arkp.hpp; topologies 0 / 1 / 4 with <= 8 sites, else null / 0):
  // entry = (protein << 2) | (paired << 1) | half, pairs on (even, odd) lanes first, then the single-lane proteins
  const int32_t* lane_unit; int n_lanes;
};

struct NetSlices {                      // offsets into the candidate vector
  int ck, A, B, C, D, Dp, E, tf;
  __host__ __device__ explicit NetSlices(int n_K, int N, int sites)
      : ck(0), A(n_K), B(n_K + N), C(n_K + 2 * N), D(n_K + 3 * N), Dp(n_K + 4 * N), E(n_K + 4 * N + sites), tf(n_K + 5 * N + sites) {}
};

__device__ __forceinline__ double softplus(double x) { return x > 20.0 ? x : log1p(exp(x)); }     // utils.py:229-241

__device__ __forceinline__ int net_bucket(const double t, const double* grid, const int n) {    // utils.py:211-225
  if (t <= grid[0]) return 0;
  if (t >= grid[n - 1]) return n - 1;
  int lo = 0, hi = n;                   // first index with grid[idx] > t  (searchsorted side='right')
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (grid[mid] <= t) lo = mid + 1; else hi = mid; }
  int j = lo - 1;
  return j < 0 ? 0 : (j >= n ? n - 1 : j);
}

// synthesis rate and its derivative with respect to the (already once-squashed for models 0/1/2) TF input u_raw
__device__ __forceinline__ double synth_rate(const double Ai, const double ts, const double u_raw, double* d_du_raw) {
  const double den = 1.0 + fabs(u_raw);
  const double u = u_raw / den;
  const double du = 1.0 / (den * den);
  if (u >= 0.0) {
    const double q = 1.0 + u + 1e-6;
    if (d_du_raw) *d_du_raw = Ai * ts * (1.0 + 1e-6) / (q * q) * du;
    return Ai * (1.0 + (ts * u) / q);
  }
  const double q = 1.0 + ts * fabs(u);
  if (d_du_raw) *d_du_raw = Ai * ts / (q * q) * du;
  return Ai / q;
}

// 1/x to full double precision from v_rcp_f64 + two Newton steps: ~6 VALU instructions instead of the ~35 of an IEEE division
__device__ __forceinline__ double net_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}

// synthesis rate only (integrator hot loop): same formulas as synth_rate with reciprocal-multiplies (relative error ~1e-16)
__device__ __forceinline__ double synth_rate_fast(const double Ai, const double ts, const double u_raw) {
  const double u = u_raw * net_rcp(1.0 + fabs(u_raw));
  if (u >= 0.0) return Ai * (1.0 + (ts * u) * net_rcp(1.0 + u + 1e-6));
  return Ai * net_rcp(1.0 + ts * fabs(u));
}

// [r3] synthesis rate of the integrator kernels from the raw TF input v0 = (TF . P_vec)_i / tf_deg_i: topologies 0 / 1 / 2 squash before
// calculate_synthesis_rate squashes again -- s(s(v)) = v / (1 + 2 |v|) -- and the two branches of the rate share ONE reciprocal chain
// (den = 1 + u + 1e-6 for u >= 0, 1 + ts |u| below): two v_rcp_f64 + Newton chains on the critical path of a stage instead of four
__device__ __forceinline__ double synth_rate_squashed(const double Ai, const double ts, const double v0, const bool squash_twice) {
  const double u = v0 * net_rcp(1.0 + (squash_twice ? 2.0 : 1.0) * fabs(v0));
  const bool pos = u >= 0.0;
  const double rden = net_rcp(pos ? 1.0 + u + 1e-6 : __builtin_fma(ts, fabs(u), 1.0));
  return pos ? Ai * __builtin_fma(ts * u, rden, 1.0) : Ai * rden;
}

// the eight LOSS_MODE point losses (global_model/lossfn.py:28-110) and the floored fold change of LOSS_FN (floor 1e-9)
__device__ __forceinline__ double point_loss(const int mode, double diff, const double obs, const double pred) {
  constexpr double EPS = 1e-9;
  switch (mode) {
    case 0: return diff * diff;
    case 1: { const double a = fabs(diff), d = 0.5; return a <= d ? 0.5 * diff * diff : d * (a - 0.5 * d); }
    case 2: { diff = log(diff + EPS) - log(obs + EPS); const double x = diff / 0.5; return 0.25 * (sqrt(1.0 + x * x) - 1.0); }
    case 3: { const double s = fabs(diff); return s > 20.0 ? s - 0.69314718056 : log(cosh(diff)); }
    case 4: return log(1.0 + diff * diff);
    case 5: return (diff * diff) / (fabs(pred) + 1e-6);
    case 6: { const double x2 = diff * diff; return x2 / (x2 + 1.0); }
    default: return sqrt(diff * diff + 1e-3 * 1e-3) - 1e-3;
  }
}

__device__ __forceinline__ double fold_change(const double a, const double b) {
  constexpr double EPS = 1e-9;
  return (a > EPS ? a : EPS) / (b > EPS ? b : EPS);
}

// LDS work area of one candidate
struct NetLds {
  double *p, *y, *Kt, *Sall, *Pvec, *synth, *dsyn;     // dsyn: d synth_i / d (TF . P_vec)_i   (Jacobian only)
  __device__ static size_t doubles(const NetDev& n) { return (size_t)n.n_var + n.S + n.n_K + n.sites + 3 * (size_t)n.N; }
  __device__ NetLds() {}
  __device__ NetLds(double* base, const NetDev& n) {
    p = base; y = p + n.n_var; Kt = y + n.S; Sall = Kt + n.n_K; Pvec = Sall + n.sites; synth = Pvec + n.N; dsyn = synth + n.N;
  }
};

// Stage 1-2 (depends on the candidate and the kinase bucket only): Kt and S_all.  All threads call; ends with a barrier.
__device__ __forceinline__ void net_prepare_bucket(const NetDev& n, const NetLds& L, const int jb) {
  const NetSlices s(n.n_K, n.N, n.sites);
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int k = tid; k < n.n_K; k += nt) L.Kt[k] = n.kin_Kmat[(size_t)k * n.n_grid + jb] * L.p[s.ck + k];
  __syncthreads();
  for (int r = tid; r < n.sites; r += nt) {
    double acc = 0.0;
    for (int q = n.W_indptr[r]; q < n.W_indptr[r + 1]; ++q) acc += n.W_data[q] * L.Kt[n.W_indices[q]];
    L.Sall[r] = acc;
  }
  __syncthreads();
}

// Stage 3-4 (depends on the state in L.y): P_vec -> TF input -> synthesis rate (and its derivative).  Ends with a barrier.
template <bool WITH_DERIV>
__device__ __forceinline__ void net_prepare_state(const NetDev& n, const NetLds& L) {
  const NetSlices s(n.n_K, n.N, n.sites);
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < n.N; i += nt) {
    const int st = n.offset_y[i];
    double tot;
    const int d = (n.model == 2) ? -1 : n.driver_map[i];            // the combinatorial RHS ignores driver_map
    if (d >= 0) tot = L.Kt[d];
    else {
      const int cnt = (n.model == 2) ? (1 << n.n_sites[i]) : 1 + n.n_sites[i];
      tot = 0.0;
      for (int m = 0; m < cnt; ++m) tot += L.y[st + 1 + m];
    }
    L.Pvec[i] = tot;
  }
  __syncthreads();
  const double ts = L.p[s.tf];
  for (int i = tid; i < n.N; i += nt) {
    double acc = 0.0;
    for (int q = n.TF_indptr[i]; q < n.TF_indptr[i + 1]; ++q) acc += n.TF_data[q] * L.Pvec[n.TF_indices[q]];
    double v = acc / n.tf_deg[i];
    double dv = 1.0 / n.tf_deg[i];
    if (n.model != 4) { const double den = 1.0 + fabs(v); dv *= 1.0 / (den * den); v = v / den; }
    double d = 0.0;
    L.synth[i] = synth_rate(L.p[s.A + i], ts, v, WITH_DERIV ? &d : nullptr);
    if (WITH_DERIV) L.dsyn[i] = d * dv;
  }
  __syncthreads();
}

template <bool WITH_DERIV>
__device__ __forceinline__ void net_prepare(const NetDev& n, const NetLds& L, const int jb) {
  net_prepare_bucket(n, L, jb);
  net_prepare_state<WITH_DERIV>(n, L);
}

// dy/dt of state `st_idx` (stage 5).  Reads only LDS.
// (i, loc, st, ss, ns) = protein, position inside its block, block start, site offset, site count of the state
// MODEL >= 0 fixes the kinetic topology at compile time (the integrator kernels); -1 reads it from the network
template <int MODEL = -1>
__device__ __forceinline__ double net_state_rhs_ctx(const NetDev& n, const NetLds& L, const int i, const int loc, const int st,
                                                    const int ss, const int ns) {
  const int model = (MODEL >= 0) ? MODEL : n.model;
  const NetSlices s(n.n_K, n.N, n.sites);
  const double* y = L.y + st;
  const double Bi = L.p[s.B + i], Ci = L.p[s.C + i], Di = L.p[s.D + i], Ei = L.p[s.E + i];
  const double* Dp = L.p + s.Dp + ss;
  const double* Sr = L.Sall + ss;
  const double R = y[0];
  if (loc == 0) return L.synth[i] - Bi * R;
  if (model == 0) {
    const double P = y[1];
    if (loc == 1) {
      if (ns == 0) return Ci * R - Di * P;
      double sumS = 0.0, back = 0.0;
      for (int j = 0; j < ns; ++j) { sumS += Sr[j]; back += Ei * y[2 + j]; }
      return Ci * R - (Di + sumS) * P + back;
    }
    const int j = loc - 2;
    return Sr[j] * P - (Ei + Dp[j] + Di) * y[loc];
  }
  if (model == 4) {
    const double P = y[1];
    if (loc == 1) {
      const double trans = (Ci * R) / (1.0 + R);
      if (ns == 0) return trans - Di * P;
      double f = 0.0, b = 0.0;
      for (int j = 0; j < ns; ++j) { f += (Sr[j] * P) / (1.0 + P); b += Ei * y[2 + j]; }
      return trans - Di * P - f + b;
    }
    const int j = loc - 2;
    const double fwd = (Sr[j] * P) / (1.0 + P);
    return fwd - (Dp[j] + Di) * y[loc] - Ei * y[loc];
  }
  if (model == 1) {
    const double P0 = y[1];
    if (loc == 1) {
      if (ns == 0) return Ci * R - Di * P0;
      return Ci * R - Di * P0 - Sr[0] * P0 + Ei * y[2];
    }
    const int j = loc - 2;                           // phospho level j + 1
    const double prev = y[loc - 1], cur = y[loc];
    if (j == ns - 1) return Sr[j] * prev - (Ei + Dp[j] + Di) * cur;
    return Sr[j] * prev + Ei * y[loc + 1] - (Sr[j + 1] + Ei + Dp[j] + Di) * cur;
  }
  // model 2: loc - 1 = bit mask m of the phospho state, y[1 + m]
  if (ns == 0) return Ci * R - Di * y[1];
  const int m = loc - 1;
  const double Pm = y[1 + m];
  double acc = (m == 0) ? Ci * R - Di * Pm : 0.0;
  double loss = 0.0;
  for (int j = 0; j < ns; ++j) {
    const int bit = 1 << j;
    if (m & bit) {
      loss += Ei + Dp[j] + Di;                       // de-phosphorylation of bit j + state decay (models.py:391-407)
      acc += Sr[j] * y[1 + (m ^ bit)];               // forward inflow from the state lacking bit j
    } else {
      loss += Sr[j];                                 // forward outflow
      acc += Ei * y[1 + (m | bit)];                  // inflow from de-phosphorylation of the state that has bit j
    }
  }
  return acc - loss * Pm;
}

__device__ __forceinline__ double net_state_rhs(const NetDev& n, const NetLds& L, const int sidx) {
  const int i = n.state_prot[sidx];
  return net_state_rhs_ctx(n, L, i, n.state_local[sidx], n.offset_y[i], n.offset_s[i], n.n_sites[i]);
}

// d f_row / d y_col  for row, col inside ONE protein block (the TF coupling is added separately)
__device__ __forceinline__ double net_block_jac(const NetDev& n, const NetLds& L, const int i, const int lr, const int lc) {
  const NetSlices s(n.n_K, n.N, n.sites);
  const int st = n.offset_y[i], ss = n.offset_s[i], ns = n.n_sites[i];
  const double* y = L.y + st;
  const double Bi = L.p[s.B + i], Ci = L.p[s.C + i], Di = L.p[s.D + i], Ei = L.p[s.E + i];
  const double* Dp = L.p + s.Dp + ss;
  const double* Sr = L.Sall + ss;
  if (lr == 0) return (lc == 0) ? -Bi : 0.0;
  if (n.model == 0 || n.model == 4) {
    const double P = y[1], R = y[0];
    const bool sat = n.model == 4;
    const double gP = sat ? 1.0 / ((1.0 + P) * (1.0 + P)) : 1.0;      // d/dP of P/(1+P)
    if (lr == 1) {
      if (lc == 0) return sat ? Ci / ((1.0 + R) * (1.0 + R)) : Ci;
      if (lc == 1) { double sumS = 0.0; for (int j = 0; j < ns; ++j) sumS += Sr[j]; return -(Di + sumS * gP); }
      return Ei;
    }
    const int j = lr - 2;
    if (lc == 1) return Sr[j] * gP;
    if (lc == lr) return -(Ei + Dp[j] + Di);
    return 0.0;
  }
  if (n.model == 1) {
    if (lr == 1) {
      if (lc == 0) return Ci;
      if (lc == 1) return ns ? -(Di + Sr[0]) : -Di;
      if (lc == 2) return Ei;
      return 0.0;
    }
    const int j = lr - 2;
    if (lc == lr - 1) return Sr[j];
    if (lc == lr) return (j == ns - 1) ? -(Ei + Dp[j] + Di) : -(Sr[j + 1] + Ei + Dp[j] + Di);
    if (lc == lr + 1 && j < ns - 1) return Ei;
    return 0.0;
  }
  // model 2
  if (ns == 0) { if (lr == 1) return lc == 0 ? Ci : (lc == 1 ? -Di : 0.0); return 0.0; }
  const int m = lr - 1;
  if (lc == 0) return (m == 0) ? Ci : 0.0;
  const int c = lc - 1;
  if (c == m) {
    double loss = (m == 0) ? Di : 0.0;
    for (int j = 0; j < ns; ++j) loss += (m & (1 << j)) ? (Ei + Dp[j] + Di) : Sr[j];
    return -loss;
  }
  const int d = m ^ c;
  if (d & (d - 1)) return 0.0;                        // more than one bit apart
  const int j = __builtin_ctz(d);
  return (m & d) ? Sr[j] : Ei;
}

}  // namespace pk
