// pk_network_rk45.hpp -- the reference's opt-in explicit network integrator, batched: Dormand-Prince 5(4) with a PI step controller,
// bucket-edge landing and cubic-Hermite output (global_model/solvers.py:293-577 for topologies 0 / 1 / 4, :580-758 for the
// combinatorial one; reached in the reference through jacspeedup.solve_custom, jacspeedup.py:31-64).  Selected with
// pk_solver_opts.method = PK_METHOD_DP5.
//
// The algorithm is followed decision for decision (same tableau, same error norm and floors, same controller, same treatment of the
// piecewise-constant kinase input: the bucket index is carried by the integrator and a step that ends on a bucket edge drops FSAL), so a
// candidate takes the same accepted / rejected steps as the reference and the output agrees to round-off (tests/test_gpu_network.py).
// Being explicit with dt <= 1 it needs >= 960 steps on the reference's time grid; the W-method of pk_network_solve*.hpp is the
// production path.  It needs only right-hand sides, so it also serves combinatorial networks of any block size.
//
// One workgroup per candidate; y, the stage point and k1..k7 live in LDS, every thread owns up to 4 states.
#pragma once
#include "pk_network_solve.hpp"

namespace pk {

namespace dp5 {
__device__ constexpr double TA[5][5] = {
    {1.0 / 5, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
__device__ constexpr double B[6] = {35.0 / 384, 0.0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
__device__ constexpr double E[7] = {71.0 / 57600, 0.0, -71.0 / 16695, 71.0 / 1920, -17253.0 / 339200, 22.0 / 525, -1.0 / 40};
constexpr double DT_INIT = 0.05, DT_MIN = 1e-6, DT_MAX = 1.0, SAFETY = 0.9, BETA = 0.04, ALPHA = 0.2 - 0.04;
}  // namespace dp5

__device__ __host__ inline size_t net_rk45_lds_bytes(const NetDev& n) {
  return ((size_t)n.n_var + n.S + n.n_K + n.sites + 3 * (size_t)n.N + 8 * (size_t)n.S + 24) * 8;
}

// A.stops_* holds ALL T output times here (stops[0] = t[0]); A.stop_out_* is unused
template <int MODEL>
__global__ __launch_bounds__(256) void net_rk45_kernel(const NetDev n, const NetSolveArgs A) {
  using namespace dp5;
  extern __shared__ __align__(16) double lds[];
  NetLds L(lds, n);
  const int S = n.S, T = A.T, G = n.n_grid;
  double* const y = L.y;
  double* const ytmp = lds + NetLds::doubles(n);
  double* const K = ytmp + S;                    // k_j at K + j * S, j = 0..6
  double* const red = K + 7 * (size_t)S;
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const double* tev = A.stops_p ? A.stops_p : A.stops_v;
  const double* grid = n.kin_grid;

  constexpr int KS = 4;                          // host guarantees S <= KS * nt
  int s_i[KS], s_loc[KS], s_st[KS], s_ss[KS], s_ns[KS];
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    const int k = tid + q * nt;
    if (k < S) { const int i = n.state_prot[k]; s_i[q] = i; s_loc[q] = n.state_local[k]; s_st[q] = n.offset_y[i]; s_ss[q] = n.offset_s[i]; s_ns[q] = n.n_sites[i]; }
    else { s_i[q] = 0; s_loc[q] = 0; s_st[q] = 0; s_ss[q] = 0; s_ns[q] = 0; }
  }
  const double* xb = A.x + b * n.n_var;
  for (int k = tid; k < n.n_var; k += nt) L.p[k] = A.x_is_raw ? softplus(xb[k]) : xb[k];
  const double* y0 = A.y0 + (A.y0_batched ? b * S : 0);
  double* Yout = A.Y + b * (size_t)T * S;
  for (int k = tid; k < S; k += nt) { const double v = y0[k]; y[k] = v; Yout[k] = v; }
  __syncthreads();

  // k_out <- f(src) in the current bucket; src is y or ytmp (LDS).  Ends with a barrier: src may be overwritten afterwards.
  auto eval = [&](double* src, double* kout) {
    L.y = src;
    net_prepare_state<false>(n, L);
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int k = tid + q * nt;
      if (k >= S) break;
      kout[k] = net_state_rhs_ctx<MODEL>(n, L, s_i[q], s_loc[q], s_st[q], s_ss[q], s_ns[q]);
    }
    __syncthreads();
  };

  int status = PK_ST_OK, nacc = 0, nrej = 0;
  double tcur = tev[0];
  const double tfin = tev[T - 1];
  int jb = 0;
  while (jb + 1 < G && tcur >= grid[jb + 1]) ++jb;
  net_prepare_bucket(n, L, jb);
  int nxt = 1;
  if (T > 1) eval(y, K);
  double dt = A.h0 > 0.0 ? A.h0 : DT_INIT;
  double err_prev = 1.0;
  bool hit = false;
  long long steps = 0;
  while (tcur < tfin && nxt < T) {
    if (++steps > (long long)A.max_steps) { status |= PK_ST_MAXSTEPS; break; }
    bool moved = false;
    while (jb + 1 < G && tcur >= grid[jb + 1]) { ++jb; hit = true; moved = true; }
    if (moved) net_prepare_bucket(n, L, jb);
    if (hit) { eval(y, K); hit = false; err_prev = 1.0; }      // the input jumped: k1 is stale

    double dt_use = dt, dist = 1e9;
    if (jb + 1 < G) {
      dist = grid[jb + 1] - tcur;
      if (dist > 1e-15 && dt_use > dist) dt_use = dist;
    }
    const double rem = tfin - tcur;
    if (dt_use > rem) dt_use = rem;
    if (dt_use < DT_MIN) dt_use = DT_MIN;

#pragma unroll 1
    for (int s = 0; s < 5; ++s) {                               // stages 2..6
      for (int k = tid; k < S; k += nt) {
        double inc = TA[s][0] * K[k];
        for (int j = 1; j <= s; ++j) inc += TA[s][j] * K[(size_t)j * S + k];
        ytmp[k] = y[k] + dt_use * inc;
      }
      __syncthreads();
      eval(ytmp, K + (size_t)(s + 1) * S);
    }
    for (int k = tid; k < S; k += nt)
      ytmp[k] = y[k] + dt_use * (B[0] * K[k] + B[2] * K[2 * (size_t)S + k] + B[3] * K[3 * (size_t)S + k] + B[4] * K[4 * (size_t)S + k] + B[5] * K[5 * (size_t)S + k]);
    __syncthreads();
    eval(ytmp, K + 6 * (size_t)S);                              // k7 = f(y_new): FSAL

    double e = 0.0;
    for (int k = tid; k < S; k += nt) {
      const double diff = dt_use * (E[0] * K[k] + E[2] * K[2 * (size_t)S + k] + E[3] * K[3 * (size_t)S + k] + E[4] * K[4 * (size_t)S + k] +
                                    E[5] * K[5 * (size_t)S + k] + E[6] * K[6 * (size_t)S + k]);
      const double ay = fabs(y[k]), an = fabs(ytmp[k]);
      double sc = A.atol + A.rtol * (ay > an ? ay : an);
      if (sc < 1e-12) sc = 1e-12;
      const double r = fabs(diff) / sc;
      e = (r > e || r != r) ? r : e;
    }
    const double err = block_max(e, red);
    if (err != err || err > 1e300) { status |= PK_ST_NONFINITE; break; }   // the reference would spin to max_steps and raise
    if (err <= 1.0) {
      ++nacc;
      const double t_next = tcur + dt_use;
      while (nxt < T && tev[nxt] <= t_next) {
        const double te = tev[nxt];
        if (te >= tcur) {
          double* row = Yout + (size_t)nxt * S;
          const double h = t_next - tcur;
          if (h < 1e-16) {
            for (int k = tid; k < S; k += nt) row[k] = ytmp[k];
          } else {
            const double tau = (te - tcur) / h, t2 = tau * tau, t3 = t2 * tau;
            const double h00 = 2 * t3 - 3 * t2 + 1, h10 = t3 - 2 * t2 + tau, h01 = -2 * t3 + 3 * t2, h11 = t3 - t2;
            for (int k = tid; k < S; k += nt) row[k] = h00 * y[k] + h10 * h * K[k] + h01 * ytmp[k] + h11 * h * K[6 * (size_t)S + k];
          }
        }
        ++nxt;
      }
      const bool on_edge = fabs(dt_use - dist) < 1e-14;
      for (int k = tid; k < S; k += nt) { y[k] = ytmp[k]; if (!on_edge) K[k] = K[6 * (size_t)S + k]; }
      __syncthreads();
      tcur = t_next;
      if (on_edge) hit = true;
      double fac = (err < 1e-12) ? 5.0 : SAFETY * pow(err, -ALPHA) * pow(err_prev, BETA);
      if (fac > 5.0) fac = 5.0;
      if (fac < 0.2) fac = 0.2;
      dt = dt * fac;
      if (dt > DT_MAX) dt = DT_MAX;
      err_prev = err < 1e-4 ? 1e-4 : err;
    } else {
      ++nrej;
      double fac = SAFETY * pow(err, -0.2);
      if (fac < 0.1) fac = 0.1;
      dt = dt_use * fac;
      if (dt < DT_MIN) dt = DT_MIN;
      err_prev = 1.0;
    }
  }
  if (status != PK_ST_OK) {
    const double qnan = __builtin_nan("");
    for (int r = nxt; r < T; ++r) for (int k = tid; k < S; k += nt) Yout[(size_t)r * S + k] = qnan;
  }
  if (tid == 0) {
    if (A.status) A.status[b] = status;
    if (A.n_steps) { A.n_steps[2 * b] = nacc; A.n_steps[2 * b + 1] = nrej; }
  }
}

}  // namespace pk
