// pk_solve_kernel.hpp -- batched per-protein solve / RHS / Jacobian kernels (gfx950), device templates.
//
// Replaces, for a batch of B parameter vectors, the reference's models.solve_ode
// (models/distmod.py:93-134, succmod.py:114-152, randmod.py:249-305): SciPy odeint(LSODA) -> clip -> flat.
//
// Mapping: one replica (parameter vector) per group of G lanes of a wavefront, lane r <-> state r
// (pk_wave.hpp).  Parameters are read once from the row-major [B, P] matrix (a group reads one contiguous
// row: coalesced), everything else lives in VGPRs; results are written once ([B, T, S], a contiguous
// S-vector per output time).  No LDS storage, no atomics, no inter-workgroup traffic.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/phoskin.h"
#include "pk_linsolve.hpp"

namespace pk {

struct SolveArgs {
  const double* theta; const double* y0; const double* t;
  double* sol; double* flat; double* metric; int32_t* status; int32_t* n_steps;
  long long B; int n_sites; int S; int P; int T; int F; int n_obs; int y0_batched; int metric_id;
  double rtol, atol, h0, rk4_h; int max_steps; int clip; int normalize; int stage_form;
};

// ------------------------------------------------------------------ RODAS4 (Hairer & Wanner, rodas.f METH=1)
// Coefficients verified against the order conditions in 50-digit arithmetic: tools/check_rodas4.py.
namespace r4 {
constexpr double GAM = 0.25;
constexpr double A21 = 0.1544000000000000e+01, A31 = 0.9466785280815826e+00, A32 = 0.2557011698983284e+00;
constexpr double A41 = 0.3314825187068521e+01, A42 = 0.2896124015972201e+01, A43 = 0.9986419139977817e+00;
constexpr double A51 = 0.1221224509226641e+01, A52 = 0.6019134481288629e+01, A53 = 0.1253708332932087e+02, A54 = -0.6878860361058950e+00;
constexpr double C21 = -0.5668800000000000e+01, C31 = -0.2430093356833875e+01, C32 = -0.2063599157091915e+00;
constexpr double C41 = -0.1073529058151375e+00, C42 = -0.9594562251023355e+01, C43 = -0.2047028614809616e+02;
constexpr double C51 = 0.7496443313967647e+01, C52 = -0.1024680431464352e+02, C53 = -0.3399990352819905e+02, C54 = 0.1170890893206160e+02;
constexpr double C61 = 0.8083246795921522e+01, C62 = -0.7981132988064893e+01, C63 = -0.3152159432874371e+02, C64 = 0.1631930543123136e+02, C65 = -0.6058818238834054e+01;
// Resolvent form for affine right-hand sides (tools/rodas4_resolvent.py, 60-digit arithmetic): with M = I - GAM h J,
// z_1 = M^{-1} h f(y_n), z_{k+1} = M^{-1} z_k:  y_{n+1} = y_n + sum RB_k z_k,  err = sum RE_k z_k  (RE_1 = 0).
constexpr double RB1 = 0.25, RB2 = -0.30261563040255475781, RB3 = 2.0437958549435498747, RB4 = -1.1490271157486588988,
                 RB5 = 0.12712918827688397052, RB6 = 0.030717702930779034113;
constexpr double RE2 = -0.27896255898136486925, RE3 = 0.80616997401331598084, RE4 = -0.74473456815175834943,
                 RE5 = 0.18680945018902833694, RE6 = 0.030717702930779034113;
}  // namespace r4

// Resolvent-form one-step methods for affine systems:  M = I - GAM h J, z_1 = M^{-1} h f(y_n), z_{k+1} = M^{-1} z_k,
//   y_{n+1} = y_n + sum_k B[k] z_k ,  err = sum_k E[k] z_k ,  step-size exponent 1 / Q  (Q = order of the estimator + 1).
template <int METHOD> struct ResolventTab;
template <> struct ResolventTab<PK_METHOD_RODAS4> {          // RODAS4 rewritten for affine f (tools/rodas4_resolvent.py)
  static constexpr int NS = 6;
  static constexpr double GAM = 0.25;
  static constexpr double Q = 4.0;
  static constexpr double B[6] = {r4::RB1, r4::RB2, r4::RB3, r4::RB4, r4::RB5, r4::RB6};
  static constexpr double E[6] = {0.0, r4::RE2, r4::RE3, r4::RE4, r4::RE5, r4::RE6};
};
// LRP8: L-stable restricted-Pade approximation of exp with 8 resolvent solves, order 7, embedded order 6 on z_1..z_7
// (both A-stable with R(inf) = 0; gamma = 0.22 lies in the s = 8 window [0.1567, 0.2344] of Hairer & Wanner II, Table IV.6.4;
// weights from the order conditions in 60-digit arithmetic: tools/restricted_pade.py 8 0.22).
template <> struct ResolventTab<PK_METHOD_LRP8> {
  static constexpr int NS = 8;
  static constexpr double GAM = 0.22;
  static constexpr double Q = 7.0;
  static constexpr double B[8] = {0.22, 0.896963014494355498137, -4.05804045796110421602, 10.5343409578260650281,
                                  -11.197471732915163878, 6.15403680296552740685, -1.75732530837770792816, 0.207496723968028089083};
  static constexpr double E[8] = {0.0, 0.207496723968028089083, -1.2449803438081685345, 3.11245085952042133624,
                                  -4.14993447936056178166, 3.11245085952042133624, -1.2449803438081685345, 0.207496723968028089083};
};

// LRP12: 12 resolvent solves, order 11, embedded order 10 on z_1..z_11; gamma = 0.16.  R(inf) = 0 (L-stable); the embedded method is
// strictly A-stable; the propagated one is A(alpha)-stable with alpha = 90 deg to within its own leading error term: on the imaginary
// axis |R(iy)| <= 1 + 4.5e-9, the excess sitting at |y| ~ 1 (2 C_12 y^12 with C_12 = 8.7e-10 > 0) and |R(iy)| < 1 beyond (tools/
// restricted_pade.py 12 0.16; tests/test_integrator_tables.py).  For comparison the reference's LSODA switches to BDF1-5 (alpha = 73 deg
// at order 4, 51 deg at order 5).  The models' Jacobians are (similar to) M-matrices with real spectra, where only |R(x)| <= 1, x <= 0, matters.  Weights alternate in sign up to ~240: three digits of cancellation
// against double precision's sixteen, five orders of magnitude below the parity band.
template <> struct ResolventTab<PK_METHOD_LRP12> {
  static constexpr int NS = 12;
  static constexpr double GAM = 0.16;
  static constexpr double Q = 11.0;
  static constexpr double B[12] = {0.16, 0.610353945449570422947, -5.673096446159684124791, 32.50341628054675047494, -100.569007871288394668,
                                   194.1476904543206939067, -240.6749692517442544225, 197.5042940245258878855, -107.8972269008007496046,
                                   38.02972288281095798187, -7.871184212939898249467, 0.7300070952791203973919};
  static constexpr double E[12] = {0.0, 0.7300070952791203973919, -7.300070952791203973919, 32.85031928756041788264, -87.60085143349444768703,
                                   153.3014900086152834523, -183.9617880103383401428, 153.3014900086152834523, -87.60085143349444768703,
                                   32.85031928756041788264, -7.300070952791203973919, 0.7300070952791203973919};
};

// err^(1/Q) for the step-size controller: single precision is ample (the factor is clamped to [1/6, 5] anyway)
__device__ __forceinline__ double root_q(double err, double Q) {
  const float e = (float)fmin(fmax(err, 1e-30), 1e30);
  return (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf(e) * (float)(1.0 / Q));
}

// Per-replica output / reduction state (one value per lane = per state row).
template <int G>
struct Emitter {
  const SolveArgs& A; long long rep; int row, lane; bool obs; double y0inv;
  double s1 = 0.0, s2 = 0.0, dyn = 0.0, prev = 0.0, shift = 0.0;
  __device__ __forceinline__ Emitter(const SolveArgs& a, long long rep_, int row_, int lane_, double y0)
      : A(a), rep(rep_), row(row_), lane(lane_) {
    obs = row < 2 + A.n_obs;
    y0inv = A.normalize ? 1.0 / y0 : 1.0;
  }
  __device__ __forceinline__ void emit(int k, double y) {
    double v = A.clip ? ((y < 0.0) ? 0.0 : y) : y;   // NaN stays NaN (np.clip semantics)
    v *= y0inv;
    if (row < A.S) {
      if (A.sol) A.sol[(rep * A.T + k) * A.S + row] = v;
      if (A.flat) {
        double* f = A.flat + rep * A.F;
        const int T5 = A.T > 5 ? A.T - 5 : 0;
        if (row == 0) { if (k >= 5) f[k - 5] = v; }
        else if (row == 1) f[T5 + k] = v;
        else if (row < 2 + A.n_obs) f[T5 + A.T + (row - 2) * A.T + k] = v;
      }
    }
    if (A.metric) {
      const double x = obs ? v : 0.0;
      if (k == 0) { shift = gsum<G>(x, lane) / (2 + A.n_obs); prev = x; }
      const double xs = obs ? x - shift : 0.0;
      s1 += x; s2 = __builtin_fma(xs, xs, s2);
      const double d = x - prev; dyn = __builtin_fma(d, d, dyn); prev = x;
    }
  }
  __device__ __forceinline__ void fill_nan(int k_from) {
    const double qnan = __builtin_nan("");
    for (int k = k_from; k < A.T; ++k) emit(k, qnan);
  }
  __device__ __forceinline__ void finish(int status, int acc, int rej) {
    if (A.metric) {
      const double L = 2.0 * A.T + (double)A.T * A.n_obs;
      const double tot = gsum<G>(s1, lane);
      double m;
      switch (A.metric_id) {
        case PK_METRIC_TOTAL_SIGNAL: m = tot; break;
        case PK_METRIC_MEAN_ACTIVITY: m = tot / L; break;
        case PK_METRIC_VARIANCE: {                       // E[(x-c)^2] - (E[x-c])^2 with c = mean at t0
          const double q = gsum<G>(s2, lane);
          const double ms = tot / L - shift;
          m = q / L - ms * ms;
        } break;
        case PK_METRIC_DYNAMICS: m = gsum<G>(dyn, lane); break;
        default: {                                        // l2_norm of the unshifted values
          const double q = gsum<G>(s2, lane);
          // sum x^2 = sum (x-c)^2 + 2 c sum x - L c^2
          m = sqrt(fmax(q + 2.0 * shift * tot - L * shift * shift, 0.0));
        } break;
      }
      if (row == 0) A.metric[rep] = m;
    }
    if (row == 0) {
      if (A.status) A.status[rep] = status;
      if (A.n_steps) { A.n_steps[2 * rep] = acc; A.n_steps[2 * rep + 1] = rej; }
    }
  }
};

// RESOLVENT: RODAS4 only -- use the resolvent form (valid because every per-protein model is affine in y); false =
// the classical 6-stage form that a nonlinear right-hand side needs.
template <int MODEL, int G, int METHOD, bool STRUCTURED, bool RESOLVENT = true>
__global__ __launch_bounds__(256) void solve_kernel(const SolveArgs A) {
  constexpr int RPB = 256 / G;                           // replicas per block
  const int lane = lane_id();
  const int row = threadIdx.x & (G - 1);
  const long long rep = (long long)blockIdx.x * RPB + (threadIdx.x / G);
  if (rep >= A.B) return;                                // whole groups leave together
  const int S = A.S, n = A.n_sites, T = A.T;

  const RowCoef c = load_row<MODEL>(A.theta + rep * A.P, n, S, row);
  const double y_init = (row < S) ? A.y0[(A.y0_batched ? rep * S : 0) + row] : 0.0;
  Emitter<G> out(A, rep, row, lane, (row < S) ? y_init : 1.0);

  double y = y_init;
  out.emit(0, y);
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  if (T < 2) { out.finish(status, 0, 0); return; }

  auto f_of = [&](double yy) { return rhs<MODEL, G>(c, yy, n, S, row, lane); };

  if constexpr (METHOD == PK_METHOD_RK4) {
    // ---------------------------------------------------------------- classical RK4, fixed step
    double tc = A.t[0];
    for (int k = 1; k < T; ++k) {
      const double te = A.t[k];
      const double span = te - tc;
      long long nsub = (long long)ceil(span / A.rk4_h);
      if (nsub < 1) nsub = 1;
      if (nacc + nsub > (long long)A.max_steps) { status |= PK_ST_MAXSTEPS; out.fill_nan(k); break; }
      const double h = span / (double)nsub;
      for (long long s = 0; s < nsub; ++s) {
        const double k1 = f_of(y);
        const double k2 = f_of(__builtin_fma(0.5 * h, k1, y));
        const double k3 = f_of(__builtin_fma(0.5 * h, k2, y));
        const double k4 = f_of(__builtin_fma(h, k3, y));
        y = __builtin_fma(h / 6.0, (k1 + k4) + 2.0 * (k2 + k3), y);
      }
      nacc += (int)nsub;
      tc = te;
      const double bad = gmax<G>((nonfinite(y)) ? 1.0 : 0.0, lane);      // inf or nan
      if (bad != 0.0) { status |= PK_ST_NONFINITE; out.fill_nan(k); break; }
      out.emit(k, y);
    }
    out.finish(status, nacc, nrej);
    return;
  } else {
    using Solver = typename SolverFor<MODEL, G, STRUCTURED>::type;
    Solver ls;
    const double rtol = A.rtol, atol = A.atol;
    double tc = A.t[0];
    int k = 1;
    double te = A.t[1];
    // first step: Hairer's hinit-lite on the max-norm
    double h;
    {
      const double f0 = f_of(y);
      const double sc = __builtin_fma(rtol, fabs(y), atol);
      const double d0 = gmax<G>(fabs(y) / sc, lane), d1 = gmax<G>(fabs(f0) / sc, lane);
      h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
      if (A.h0 > 0.0) h = A.h0;
      if (!(h > 0.0) || h != h) h = 1e-6;
    }

    if constexpr (METHOD == PK_METHOD_RODAS4 || METHOD == PK_METHOD_LRP8 || METHOD == PK_METHOD_LRP12) {
      using namespace r4;
      constexpr bool RES = RESOLVENT || METHOD == PK_METHOD_LRP8 || METHOD == PK_METHOD_LRP12;
      using Tab = ResolventTab<METHOD>;
      constexpr double GAMMA = Tab::GAM;
      bool after_reject = false;
      while (true) {
        if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; out.fill_nan(k); break; }
        const bool last = (tc + 1.0001 * h >= te);
        const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
        if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; out.fill_nan(k); break; }
        const double hinv = 1.0 / hs;
        const double gW = hinv * (1.0 / GAMMA);                          // W = gW I - J = M / (GAMMA h)
        ls.factor(c, gW, S, row, lane);
        double yn, u6;
        if constexpr (RES) {
          // z_1 = M^{-1} h f(y) = W^{-1} f(y) / GAMMA ; z_{k+1} = M^{-1} z_k = gW W^{-1} z_k
          double z = ls.solve(f_of(y), S, row, lane) * (1.0 / GAMMA);
          yn = __builtin_fma(Tab::B[0], z, y);
          u6 = 0.0;
          static_for<Tab::NS - 1>([&](auto kc) {
            constexpr int kk = 1 + decltype(kc)::value;
            z = ls.solve(z, S, row, lane) * gW;
            yn = __builtin_fma(Tab::B[kk], z, yn);
            u6 = __builtin_fma(Tab::E[kk], z, u6);
          });
        } else {
        const double u1 = ls.solve(f_of(y), S, row, lane);
        const double u2 = ls.solve(__builtin_fma(C21 * hinv, u1, f_of(__builtin_fma(A21, u1, y))), S, row, lane);
        const double u3 = ls.solve(f_of(y + (A31 * u1 + A32 * u2)) + hinv * (C31 * u1 + C32 * u2), S, row, lane);
        const double u4 = ls.solve(f_of(y + (A41 * u1 + A42 * u2 + A43 * u3)) + hinv * (C41 * u1 + C42 * u2 + C43 * u3), S, row, lane);
        yn = y + (A51 * u1 + A52 * u2 + A53 * u3 + A54 * u4);
        const double u5 = ls.solve(f_of(yn) + hinv * (C51 * u1 + C52 * u2 + C53 * u3 + C54 * u4), S, row, lane);
        yn += u5;
        u6 = ls.solve(f_of(yn) + hinv * (C61 * u1 + C62 * u2 + C63 * u3 + C64 * u4 + C65 * u5), S, row, lane);
        yn += u6;
        }
        const double sc = __builtin_fma(rtol, fmax(fabs(y), fabs(yn)), atol);
        const double err = gmax<G>(fabs(u6) / sc, lane);                // NaN-propagating
        if (err != err || err > 1e300) {
          // non-finite stage: retry with a much smaller step; give up through the HMIN test above
          ++nrej; after_reject = true; h = 0.1 * hs;
          if (gmax<G>((nonfinite(y)) ? 1.0 : 0.0, lane) != 0.0) { status |= PK_ST_NONFINITE; out.fill_nan(k); break; }
          continue;
        }
        double fac = root_q(err, Tab::Q) * (1.0 / 0.9);
        fac = fmax(1.0 / 6.0, fmin(5.0, fac));
        double hnew = hs / fac;
        if (err <= 1.0) {
          ++nacc;
          y = yn; tc += hs;
          if (after_reject) hnew = fmin(hnew, hs);
          after_reject = false;
          if (last) {
            tc = te;
            out.emit(k, y);
            ++k;
            h = (hs < h) ? fmax(hnew, h) : hnew;                        // keep the untruncated proposal
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
      return;
    } else {
      // ---------------------------------------------------------------- variable-step BDF2 (BDF1 start-up)
      // y_{n+1} - a1 y_n + a2 y_{n-1} = beta h f(y_{n+1}),  w = h_n / h_{n-1}:
      //   a1 = (1+w)^2/(1+2w), a2 = w^2/(1+2w), beta = (1+w)/(1+2w).
      // f is affine on this path (J constant), so one Newton step from the predictor is exact:
      //   (1/(beta h) I - J) d = f(p) - (p - a1 y_n + a2 y_{n-1})/(beta h),  y_{n+1} = p + d.
      // Local error estimate: LTE ~ C (y_{n+1} - p) with the predictor p the quadratic extrapolation through
      // y_n, y_{n-1}, y_{n-2} (linear while fewer points exist).
      double ym1 = y, ym2 = y;          // y_{n-1}, y_{n-2}
      double hm1 = 0.0, hm2 = 0.0;      // h_{n-1}, h_{n-2}
      int hist = 0;                     // accepted steps so far (capped at 2); < 2: BDF1 start-up
      while (true) {
        if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; out.fill_nan(k); break; }
        if (hist >= 1) h = fmin(h, 2.0 * hm1);                          // step ratio <= 2 (zero-stability of BDF2)
        const bool last = (tc + 1.0001 * h >= te);
        const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
        if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; out.fill_nan(k); break; }
        double yn, err;
        if (hist < 2) {
          // BDF1: (1/h I - J) d = f(p) - (p - y)/h from the explicit-Euler predictor p; LTE ~ (y1 - p)/2
          const double g = 1.0 / hs;
          ls.factor(c, g, S, row, lane);
          const double p = __builtin_fma(hs, f_of(y), y);
          const double d = ls.solve(f_of(p) - g * (p - y), S, row, lane);
          yn = p + d;
          const double sc = __builtin_fma(rtol, fmax(fabs(y), fabs(yn)), atol);
          err = gmax<G>(0.5 * fabs(d) / sc, lane);
        } else {
          const double w = hs / hm1;
          const double a1 = (1.0 + w) * (1.0 + w) / (1.0 + 2.0 * w);
          const double a2 = w * w / (1.0 + 2.0 * w);
          const double beta = (1.0 + w) / (1.0 + 2.0 * w);
          // predictor: quadratic through (t_n, y), (t_n - hm1, ym1), (t_n - hm1 - hm2, ym2) evaluated at t_n + hs
          const double d1 = (y - ym1) / hm1;
          const double d2 = (ym1 - ym2) / hm2;
          const double dd = (d1 - d2) / (hm1 + hm2);
          const double p = y + hs * d1 + hs * (hs + hm1) * dd;
          const double g = 1.0 / (beta * hs);
          ls.factor(c, g, S, row, lane);
          const double d = ls.solve(f_of(p) - g * (p - a1 * y + a2 * ym1), S, row, lane);
          yn = p + d;
          // predictor error PE = h (h+hm1) (h+hm1+hm2) y3/6, BDF2 local error LE = beta h^2 (h+hm1) y3/6 (y3 = third
          // derivative), d = PE - LE  =>  LE = d * beta h / (h + hm1 + hm2 - beta h)
          const double ecoef = beta * hs / (hs + hm1 + hm2 - beta * hs);
          const double sc = __builtin_fma(rtol, fmax(fabs(y), fabs(yn)), atol);
          err = gmax<G>(ecoef * fabs(d) / sc, lane);
        }
        if (err != err || err > 1e300) {
          ++nrej; h = 0.1 * hs;
          if (gmax<G>((nonfinite(y)) ? 1.0 : 0.0, lane) != 0.0) { status |= PK_ST_NONFINITE; out.fill_nan(k); break; }
          continue;
        }
        double fac = ((hist < 2) ? sqrt(err) : cbrt(err)) * (1.0 / 0.9);
        fac = fmax(0.5, fmin(5.0, fac));                                // growth <= 2x per step
        const double hnew = hs / fac;
        if (err <= 1.0) {
          ++nacc;
          ym2 = ym1; hm2 = hm1; ym1 = y; hm1 = hs; y = yn; tc += hs;
          if (hist < 2) ++hist;
          if (last) {
            tc = te;
            out.emit(k, y);
            ++k;
            h = (hs < h) ? fmax(hnew, h) : hnew;
            if (k >= T) break;
            te = A.t[k];
          } else {
            h = hnew;
          }
        } else {
          ++nrej;
          h = hnew;
        }
      }
      out.finish(status, nacc, nrej);
      return;
    }
  }
}

// ---------------------------------------------------------------------------- RHS / Jacobian batch kernels
template <int MODEL, int G>
__global__ __launch_bounds__(256) void rhs_kernel(const double* __restrict__ theta, const double* __restrict__ y,
                                                  double* __restrict__ dydt, long long B, int n, int S, int P) {
  const int lane = lane_id();
  const int row = threadIdx.x & (G - 1);
  const long long rep = (long long)blockIdx.x * (256 / G) + (threadIdx.x / G);
  if (rep >= B) return;
  const RowCoef c = load_row<MODEL>(theta + rep * P, n, S, row);
  const double yy = (row < S) ? y[rep * S + row] : 0.0;
  const double f = rhs<MODEL, G>(c, yy, n, S, row, lane);
  if (row < S) dydt[rep * S + row] = f;
}

// Steady state of the affine system: J y* = -b, i.e. W y* = b with W = 0 * I - J -- the same factor / solve pair as an implicit stage
// with g = 0 (arrow / tridiagonal elimination, dense inverse for the random model).  Replaces the SLSQP feasibility problem of
// steady/initdist.py:9-50, initsucc.py:9-55, initrand.py:9-77 (which fixes all rates to 1) for ARBITRARY per-replica theta.
// A singular J (no degradation) yields non-finite values: flagged PK_ST_NONFINITE, row filled with NaN.
template <int MODEL, int G>
__global__ __launch_bounds__(256) void steady_kernel(const double* __restrict__ theta, double* __restrict__ yss, int32_t* __restrict__ status,
                                                     long long B, int n, int S, int P) {
  const int lane = lane_id();
  const int row = threadIdx.x & (G - 1);
  const long long rep = (long long)blockIdx.x * (256 / G) + (threadIdx.x / G);
  if (rep >= B) return;
  const RowCoef c = load_row<MODEL>(theta + rep * P, n, S, row);
  typename SolverFor<MODEL, G, true>::type solver;
  solver.factor(c, 0.0, S, row, lane);
  const double x = solver.solve(c.bias, S, row, lane);
  const double bad = gmax<G>((row < S && (nonfinite(x))) ? 1.0 : 0.0, lane);
  if (row < S) yss[rep * S + row] = (bad != 0.0) ? __builtin_nan("") : x;
  if (row == 0 && status) status[rep] = (bad != 0.0) ? PK_ST_NONFINITE : PK_ST_OK;
}

template <int MODEL, int G>
__global__ __launch_bounds__(256) void jac_kernel(const double* __restrict__ theta, double* __restrict__ J,
                                                  long long B, int n, int S, int P) {
  const int row = threadIdx.x & (G - 1);
  const long long rep = (long long)blockIdx.x * (256 / G) + (threadIdx.x / G);
  if (rep >= B) return;
  const RowCoef c = load_row<MODEL>(theta + rep * P, n, S, row);
  if (row >= S) return;
  double* out = J + (rep * S + row) * S;
  static_for<G>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if (j < S) out[j] = jac_entry<MODEL, j>(c, S, row);
  });
}

}  // namespace pk

