/* TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- plain-C, scalar restatement for the distributive model of
 *   (1) the reference right-hand side  models/distmod.py:7-65 (ode_core), and
 *   (2) the ALGORITHM the HIP throughput kernel runs (phoskintime_amd/csrc/pk_dist_fast.hpp): adaptive LRP8 / LRP12 in resolvent form with
 *       arrow elimination, same coefficients, same step controller, same landing rule.
 * Purpose: check the GPU kernel against an independent CPU implementation of the same method (agreement far tighter than the
 * parity band), test the method itself on the CPU-only suite, and time "same algorithm on the host cores" next to the GPU
 * (bench.py: cpu_same_algorithm).  Built by __graft_entry__.build() with gcc into oracle/_build/. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* resolvent-form tables (phoskintime_amd/csrc/pk_solve_kernel.hpp ResolventTab; weights from tools/restricted_pade.py) */
static const double GAM8 = 0.22;
static const double LB8[8] = {0.22, 0.896963014494355498137, -4.05804045796110421602, 10.5343409578260650281,
                              -11.197471732915163878, 6.15403680296552740685, -1.75732530837770792816, 0.207496723968028089083};
static const double LE8[8] = {0.0, 0.207496723968028089083, -1.2449803438081685345, 3.11245085952042133624,
                              -4.14993447936056178166, 3.11245085952042133624, -1.2449803438081685345, 0.207496723968028089083};
static const double GAM12 = 0.16;
static const double LB12[12] = {0.16, 0.610353945449570422947, -5.673096446159684124791, 32.50341628054675047494, -100.569007871288394668,
                                194.1476904543206939067, -240.6749692517442544225, 197.5042940245258878855, -107.8972269008007496046,
                                38.02972288281095798187, -7.871184212939898249467, 0.7300070952791203973919};
static const double LE12[12] = {0.0, 0.7300070952791203973919, -7.300070952791203973919, 32.85031928756041788264, -87.60085143349444768703,
                                153.3014900086152834523, -183.9617880103383401428, 153.3014900086152834523, -87.60085143349444768703,
                                32.85031928756041788264, -7.300070952791203973919, 0.7300070952791203973919};

/* models/distmod.py:7-65 */
void oracle_dist_rhs(const double* y, const double* th, int n, double* dy) {
  const double A = th[0], B = th[1], C = th[2], D = th[3];
  const double* S = th + 4; const double* Dr = th + 4 + n;
  const double R = y[0], P = y[1];
  double sumS = 0.0, sumP = 0.0;
  for (int i = 0; i < n; ++i) sumS += S[i];
  for (int i = 0; i < n; ++i) sumP += y[2 + i];
  dy[0] = A - B * R;
  dy[1] = C * R - (D + sumS) * P + sumP;
  for (int i = 0; i < n; ++i) dy[2 + i] = S[i] * P - (1.0 + Dr[i]) * y[2 + i];
}

/* x = (I - q J)^{-1} r for the arrow matrix of the distributive model */
static void arrow_solve(const double* r, double* x, const double* th, int n, double q) {
  const double B = th[1], C = th[2], D = th[3];
  const double* S = th + 4; const double* Dr = th + 4 + n;
  double sumS = 0.0, scw = 0.0, st = 0.0;
  for (int i = 0; i < n; ++i) sumS += S[i];
  const double xR = r[0] / (1.0 + q * B);
  for (int i = 0; i < n; ++i) {
    const double w = 1.0 / (1.0 + q * (1.0 + Dr[i]));
    const double t = r[2 + i] * w;
    x[2 + i] = t; st += t; scw += q * S[i] * w;
  }
  const double xP = (r[1] + q * (C * xR + st)) / (1.0 + q * (D + sumS) - q * scw);
  x[0] = xR; x[1] = xP;
  for (int i = 0; i < n; ++i) x[2 + i] += (q * S[i] / (1.0 + q * (1.0 + Dr[i]))) * xP;
}

/* One replica.  Returns status bits (1 non-finite, 2 max steps, 4 step underflow) like the HIP kernels; sol is [T, S] (raw, unclipped). */
int oracle_lrp8_dist_one(const double* th, int n, const double* y0, const double* t, int T, double rtol, double atol, int max_steps,
                         int stages, double* sol, int* n_acc, int* n_rej) {
  const int S = n + 2;
  const int NS = (stages == 12) ? 12 : 8;                 /* LRP12 (default of the library) or LRP8 */
  const double GAM = (NS == 12) ? GAM12 : GAM8;
  const double* LB = (NS == 12) ? LB12 : LB8;
  const double* LE = (NS == 12) ? LE12 : LE8;
  double y[66], yn[66], e[66], z[66], f[66];
  memcpy(y, y0, S * sizeof(double));
  memcpy(sol, y, S * sizeof(double));
  int acc = 0, rej = 0, status = 0, after_reject = 0;
  if (T < 2) { *n_acc = 0; *n_rej = 0; return 0; }
  double tc = t[0], h;
  {
    oracle_dist_rhs(y, th, n, f);
    double d0 = 0.0, d1 = 0.0;
    for (int i = 0; i < S; ++i) { const double sc = atol + rtol * fabs(y[i]); d0 = fmax(d0, fabs(y[i]) / sc); d1 = fmax(d1, fabs(f[i]) / sc); }
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
  }
  for (int k = 1; k < T && !status; ++k) {
    const double te = t[k];
    for (;;) {
      if (acc + rej >= max_steps) { status |= 2; break; }
      const int last = (tc + 1.0001 * h >= te);
      const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
      if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= 4; break; }
      const double q = GAM * hs;
      oracle_dist_rhs(y, th, n, f);
      for (int i = 0; i < S; ++i) f[i] *= hs;
      arrow_solve(f, z, th, n, q);
      for (int i = 0; i < S; ++i) { yn[i] = y[i] + LB[0] * z[i]; e[i] = 0.0; }
      for (int s = 1; s < NS; ++s) {
        memcpy(f, z, S * sizeof(double));
        arrow_solve(f, z, th, n, q);
        for (int i = 0; i < S; ++i) { yn[i] += LB[s] * z[i]; e[i] += LE[s] * z[i]; }
      }
      double err = 0.0; int bad = 0;
      for (int i = 0; i < S; ++i) {
        const double v = fabs(e[i]) / (atol + rtol * fmax(fabs(y[i]), fabs(yn[i])));
        if (v != v) bad = 1;
        if (v > err) err = v;
      }
      if (bad || err > 1e300) {
        ++rej; after_reject = 1; h = 0.1 * hs;
        int nf = 0;
        for (int i = 0; i < S; ++i) if (y[i] - y[i] != 0.0) nf = 1;
        for (int i = 0; i < 4 + 2 * n; ++i) if (th[i] - th[i] != 0.0) nf = 1;
        if (nf) { status |= 1; break; }
        continue;
      }
      double fac = pow(fmin(fmax(err, 1e-30), 1e30), 1.0 / (NS - 1.0)) / 0.9;
      fac = fmax(1.0 / 6.0, fmin(5.0, fac));
      double hnew = hs / fac;
      if (err <= 1.0) {
        ++acc;
        memcpy(y, yn, S * sizeof(double)); tc += hs;
        if (after_reject) hnew = fmin(hnew, hs);
        after_reject = 0;
        if (last) { tc = te; h = (hs < h) ? fmax(hnew, h) : hnew; break; }
        h = hnew;
      } else { ++rej; after_reject = 1; h = hnew; }
    }
    if (status) { for (int kk = k; kk < T; ++kk) for (int i = 0; i < S; ++i) sol[kk * S + i] = NAN; break; }
    memcpy(sol + (size_t)k * S, y, S * sizeof(double));
  }
  *n_acc = acc; *n_rej = rej;
  return status;
}

/* Batch over replicas [b0, b1): theta [B, P], shared y0 [S]; sol [B, T, S]; status / steps may be NULL. */
void oracle_lrp8_dist_batch(const double* theta, long b0, long b1, int n, const double* y0, const double* t, int T, double rtol, double atol,
                            int max_steps, int stages, double* sol, int32_t* status, int32_t* n_steps) {
  const int S = n + 2, P = 4 + 2 * n;
  for (long b = b0; b < b1; ++b) {
    int a = 0, r = 0;
    const int st = oracle_lrp8_dist_one(theta + b * P, n, y0, t, T, rtol, atol, max_steps, stages, sol + (size_t)b * T * S, &a, &r);
    if (status) status[b] = st;
    if (n_steps) { n_steps[2 * b] = a; n_steps[2 * b + 1] = r; }
  }
}
