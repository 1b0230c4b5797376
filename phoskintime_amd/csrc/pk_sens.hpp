// pk_sens.hpp -- forward parameter sensitivities of the per-protein solves: sens_kernel<Sys, GP>.
//
// What the reference does: scipy.optimize.curve_fit (paramest/normest.py:167-326, paramest/toggle.py) differentiates models.solve_ode
// by forward differences, 1 + P solves per Jacobian.  This kernel returns flat(theta) AND d flat / d theta [F, P] from ONE integration:
// the models are affine in theta (y' = A(theta) y + b(theta)), so the tangent y'_c = dy / dtheta_c obeys
//     y'_c' = A y'_c + A'_c y + b'_c ,   A'_c = dA / dtheta_c (constant, sparse),   y'_c(0) = 0   (the initial condition is data),
// and the resolvent method (M = I - g h A;  M z_1 = h (A y + b),  M z_{k+1} = z_k,  y+ = y + sum B_k z_k) differentiates stage by stage
// with the SAME matrix M (internal differentiation: exact derivative of the discrete solution for the step sequence taken):
//     M z'_1 = h (A y' + b' + A' (y + g z_1)) ,     M z'_{k+1} = z'_k + g h A' z_{k+1} ,     y'+ = y' + sum B_k z'_k .
// Mapping: one COLUMN per lane.  A replica owns a group of GP lanes (GP = 8, 16, 32, 64 >= 1 + P): lane 0 integrates y, lane c >= 1 the
// tangent for theta_{c-1}; every lane keeps its column (S values) in registers and runs the same O(S) arrow / Thomas (or 2^n-row
// Gauss-Jordan) solve as the thread-per-replica kernels (pk_tpr.hpp, pk_tpr_rand.hpp) with the factors of the one shared M.  The only
// coupling, A'_c z_k, needs the base lane's stage vector: the base publishes it in LDS (S doubles per group) and the tangent lanes run
// ONE STAGE BEHIND (NS + 1 rounds of solves for NS stages), so that z_{k+1} is there when z'_{k+1} is formed.  A'_c v is evaluated with
// the model's own right-hand-side routine on the derivative coefficients (coefficients of the unit vector e_c with the structural
// constants dropped): no per-model tables of sparse derivative patterns.
// The error norm takes the maximum over ALL columns (the tangents are held to the same rtol / atol as the states), the step-size
// sequence is therefore shared by the group; clipping (y < 0 -> 0) zeroes the derivative like it zeroes the value.
// One wave per workgroup, NG = 64 / GP replicas per wave; LDS: two thread-private coefficient sets + 2 S doubles per group.
#pragma once
#include "pk_solve_kernel.hpp"

namespace pk {

struct SensArgs {
  SolveArgs s;
  double* dflat;        // [B, F, P]
};

// LDS pointers carry their address space explicitly: a pointer that has been through a struct member or a select otherwise degrades to
// a generic one, and its loads / stores to `flat_*` instructions (measured on the first version of the cube kernels: 7x slower)
using lds_f64 = __attribute__((address_space(3))) double;

// compiler-level ordering of the wave's LDS traffic (the hardware executes a wave's LDS instructions in order)
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Park {            // thread-private LDS slots, slot-major (stride 64: conflict-free)
  lds_f64* p;
  __device__ __forceinline__ double ld(int k) const { return p[k * 64]; }
  __device__ __forceinline__ void st(int k, double v) const { p[k * 64] = v; }
};

// ---------------------------------------------------------------------------------------------- distributive / successive model
// rows: 0 = R, 1 = P, 2 + j = site (dist) / level (succ) j.  Coefficients: c1[NR] (coupling to P / to the previous row), dgn[NR] (-diag), cA.
template <int MODEL, int NS>
struct ChainSys {
  static_assert(MODEL == M_DIST || MODEL == M_SUCC, "chain systems");
  static constexpr int NR = NS + 2;
  static constexpr int NCOEF = 2 * NR + 1;
  static constexpr int K_C1 = 0, K_DG = NR, K_A = 2 * NR;
  static constexpr int MIN_WAVES = NS <= 3 ? 3 : NS <= 9 ? 2 : 1;      // measured: a third wave at NS = 5 costs spills and time
  double winv[NR], fx[NR], qq, cC;

  static __host__ __device__ constexpr int n_params(int n) { return 4 + 2 * n; }

  // coefficients of the parameter vector tv(.), `one` = the structural constants (1 for the model itself, 0 for a derivative)
  template <class TV>
  static __device__ __forceinline__ void build(const Park& pk, const int base, TV&& tv, const double one, const int n) {
    double sumS = 0.0;
    for (int j = 0; j < n; ++j) sumS += tv(4 + j);
    pk.st(base + K_A, tv(0));
    pk.st(base + K_C1 + 0, 0.0); pk.st(base + K_DG + 0, tv(1));
    pk.st(base + K_C1 + 1, tv(2));
    pk.st(base + K_DG + 1, MODEL == M_DIST ? tv(3) + sumS : tv(3) + (n > 0 ? tv(4) : 0.0));
    static_for<NS>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const bool ok = j < n;
      pk.st(base + K_C1 + 2 + j, ok ? tv(4 + j) : 0.0);
      double d;
      if (MODEL == M_DIST) d = ok ? one + tv(4 + n + j) : one;
      else d = ok ? ((j == n - 1) ? one + tv(4 + n + j) : one + tv(4 + j + 1) + tv(4 + n + j)) : one;
      pk.st(base + K_DG + 2 + j, d);
    });
  }
  // f = A(coefficients at `base`) Y + constA e_R
  static __device__ __forceinline__ void apply(const Park& pk, const int base, const double one, const double constA, const double (&Y)[NR],
                                               double (&f)[NR], const int S) {
    f[0] = __builtin_fma(-pk.ld(base + K_DG), Y[0], constA);
    if (MODEL == M_DIST) {
      double sv[NS];
      static_for<NS>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        sv[j] = Y[2 + j];
        f[2 + j] = __builtin_fma(pk.ld(base + K_C1 + 2 + j), Y[1], -pk.ld(base + K_DG + 2 + j) * Y[2 + j]);
      });
      f[1] = __builtin_fma(pk.ld(base + K_C1 + 1), Y[0], __builtin_fma(-pk.ld(base + K_DG + 1), Y[1], one * tree_sum(sv)));
    } else {
      static_for<NR - 1>([&](auto ic) {
        constexpr int i = 1 + decltype(ic)::value;
        double v = __builtin_fma(pk.ld(base + K_C1 + i), Y[i - 1], -pk.ld(base + K_DG + i) * Y[i]);
        if constexpr (i + 1 < NR) { if (i + 1 < S) v = __builtin_fma(one, Y[i + 1], v); }
        f[i] = (i < S) ? v : 0.0;
      });
    }
  }
  // factors of M = I - q A (coefficient set 0), as in pk_tpr.hpp
  __device__ __forceinline__ void factor(const Park& pk, const double q, const int S) {
    qq = q; cC = pk.ld(K_C1 + 1);
    winv[0] = fast_rcp(__builtin_fma(q, pk.ld(K_DG), 1.0));
    fx[0] = 0.0;
    if (MODEL == M_DIST) {
      double cwv[NS];
      static_for<NS>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        winv[2 + j] = fast_rcp(__builtin_fma(q, pk.ld(K_DG + 2 + j), 1.0));
        cwv[j] = q * pk.ld(K_C1 + 2 + j) * winv[2 + j];
        fx[2 + j] = cwv[j];
      });
      winv[1] = fast_rcp(__builtin_fma(q, pk.ld(K_DG + 1) - tree_sum(cwv), 1.0));
      fx[1] = 0.0;
    } else {
      static_for<NR - 1>([&](auto ic) {
        constexpr int i = 1 + decltype(ic)::value;
        const double up_prev = (i - 1 >= 1 && i < S) ? -q : 0.0;
        const double lo = (i < S) ? (-q * pk.ld(K_C1 + i)) * winv[i - 1] : 0.0;
        fx[i] = lo;
        winv[i] = fast_rcp(__builtin_fma(q, pk.ld(K_DG + i), 1.0) - lo * up_prev);
      });
    }
  }
  __device__ __forceinline__ void solve(const double (&r)[NR], double (&x)[NR], const int S) const {
    if (MODEL == M_DIST) {
      const double xR = r[0] * winv[0];
      double t[NS];
      static_for<NS>([&](auto jc) { constexpr int j = decltype(jc)::value; t[j] = r[2 + j] * winv[2 + j]; });
      const double xP = __builtin_fma(qq, __builtin_fma(cC, xR, tree_sum(t)), r[1]) * winv[1];
      x[0] = xR; x[1] = xP;
      static_for<NS>([&](auto jc) { constexpr int j = decltype(jc)::value; x[2 + j] = __builtin_fma(fx[2 + j], xP, t[j]); });
    } else {
      double g[NR];
      g[0] = r[0];
      static_for<NR - 1>([&](auto ic) { constexpr int i = 1 + decltype(ic)::value; g[i] = __builtin_fma(-fx[i], g[i - 1], r[i]); });
      x[NR - 1] = g[NR - 1] * winv[NR - 1];
      static_for<NR - 1>([&](auto ic) {
        constexpr int i = NR - 2 - decltype(ic)::value;
        const double up = (i >= 1 && i + 1 < S) ? -qq : 0.0;
        x[i] = __builtin_fma(-up, x[i + 1], g[i]) * winv[i];
      });
    }
  }
  __device__ __forceinline__ bool coef_nonfinite(const Park& pk) const {
    bool nf = nonfinite(pk.ld(K_A));
    static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; nf = nf || nonfinite(pk.ld(K_C1 + i)) || nonfinite(pk.ld(K_DG + i)); });
    return nf;
  }
  // ---- the interface sens_kernel uses
  Park pk; int S_;
  static constexpr size_t lds_doubles(int) { return (size_t)2 * NCOEF * 64; }
  __device__ __forceinline__ void init(lds_f64* lds, const int lane, const int, const int c, const double* __restrict__ th, const int n, const int S) {
    pk = Park{lds + lane}; S_ = S;
    build(pk, 0, [&](int i) { return th[i]; }, 1.0, n);
    build(pk, NCOEF, [&](int i) { return (i == c - 1) ? 1.0 : 0.0; }, 0.0, n);
  }
  __device__ __forceinline__ double cA() const { return pk.ld(K_A); }
  __device__ __forceinline__ double dA() const { return pk.ld(NCOEF + K_A); }
  // Y(ic) = element ic::value of the operand, out(ic, v) receives row ic::value of the product
  template <class YF, class OF>
  __device__ __forceinline__ void apply_base(YF&& Y, OF&& out, const double constA) const {
    double a[NR], f[NR];
    static_for<NR>([&](auto ic) { a[decltype(ic)::value] = Y(ic); });
    apply(pk, 0, 1.0, constA, a, f, S_);
    static_for<NR>([&](auto ic) { out(ic, f[decltype(ic)::value]); });
  }
  template <class YF, class OF>
  __device__ __forceinline__ void apply_deriv(YF&& W, OF&& out, const double constA) const {
    double a[NR], f[NR];
    static_for<NR>([&](auto ic) { a[decltype(ic)::value] = W(ic); });
    apply(pk, NCOEF, 0.0, constA, a, f, S_);
    static_for<NR>([&](auto ic) { out(ic, f[decltype(ic)::value]); });
  }
  __device__ __forceinline__ void factor(const double q) { factor(pk, q, S_); }
  template <class OF>
  __device__ __forceinline__ void solve(const double (&r)[NR], OF&& out) const {
    double x[NR];
    solve(r, x, S_);
    static_for<NR>([&](auto ic) { out(ic, x[decltype(ic)::value]); });
  }
  __device__ __forceinline__ bool coef_nonfinite() const { return coef_nonfinite(pk); }
};

// ---------------------------------------------------------------------------------------------- random model, n = NB <= 3 sites
// rows: 0 = R, 1 + m = bit mask m (m = 0: the unphosphorylated protein).  Coefficients: dg[NM], ci[NM], cA, cB, cC
// (models/randmod.py:122-247 incl. the lowest-set-bit rate quirk at :201, as in pk_tpr_rand.hpp).
template <int NB>
struct CubeSys {
  static constexpr int NM = 1 << NB;
  static constexpr int NR = NM + 1;
  static constexpr int NCOEF = 2 * NM + 3;
  static constexpr int MIN_WAVES = NB <= 2 ? 3 : 1;                     // NB = 3 holds the 8 x 8 inverse in registers
  static constexpr int K_DG = 0, K_CI = NM, K_A = 2 * NM, K_B = 2 * NM + 1, K_C = 2 * NM + 2;
  double a[NM][NM], winvR, qC;

  static __host__ __device__ constexpr int n_params(int n) { return 4 + n + (1 << n) - 1; }

  template <class TV>
  static __device__ __forceinline__ void build(const Park& pk, const int base, TV&& tv, const double one, const int) {
    pk.st(base + K_A, tv(0)); pk.st(base + K_B, tv(1)); pk.st(base + K_C, tv(2));
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      if constexpr (m == 0) {
        double sumS = 0.0;
        for (int j = 0; j < NB; ++j) sumS += tv(4 + j);
        pk.st(base + K_DG, tv(3) + sumS); pk.st(base + K_CI, 0.0);
      } else {
        constexpr int lsb = __builtin_ctz(m);
        pk.st(base + K_CI + m, tv(4 + lsb));
        double out = 0.0;
        for (int j = 0; j < NB; ++j) out += (m & (1 << j)) ? one : tv(4 + (j < lsb ? j : lsb));
        pk.st(base + K_DG + m, out + tv(4 + NB + m - 1));
      }
    });
  }
  static __device__ __forceinline__ void apply(const Park& pk, const int base, const double one, const double constA, const double (&Y)[NR],
                                               double (&f)[NR], const int) {
    f[0] = __builtin_fma(-pk.ld(base + K_B), Y[0], constA);
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const double ci = pk.ld(base + K_CI + m);
      double v = -pk.ld(base + K_DG + m) * Y[1 + m];
      static_for<NB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        v = __builtin_fma((m & (1 << j)) ? ci : one, Y[1 + (m ^ (1 << j))], v);
      });
      f[1 + m] = v;
    });
    f[1] = __builtin_fma(pk.ld(base + K_C), Y[0], f[1]);
  }
  __device__ __forceinline__ void factor(const Park& pk, const double q, const int) {
    winvR = fast_rcp(__builtin_fma(q, pk.ld(K_B), 1.0));
    qC = q * pk.ld(K_C);
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const double ci = pk.ld(K_CI + m), dgs = pk.ld(K_DG + m);
      static_for<NM>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int d = m ^ c;
        if constexpr (c == m) a[m][c] = __builtin_fma(q, dgs, 1.0);
        else if constexpr ((d & (d - 1)) == 0) a[m][c] = -q * ((m & d) ? ci : 1.0);
        else a[m][c] = 0.0;
      });
    });
    static_for<NM>([&](auto kc) {                              // in-place Gauss-Jordan inverse (no pivoting: M-matrix)
      constexpr int kk = decltype(kc)::value;
      const double rp = fast_rcp(a[kk][kk]);
      static_for<NM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i != kk) {
          const double ml = a[i][kk] * rp;
          static_for<NM>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j != kk) a[i][j] = __builtin_fma(-ml, a[kk][j], a[i][j]);
          });
          a[i][kk] = -ml;
        }
      });
      static_for<NM>([&](auto jc) { constexpr int j = decltype(jc)::value; if constexpr (j != kk) a[kk][j] *= rp; });
      a[kk][kk] = rp;
    });
  }
  __device__ __forceinline__ void solve(const double (&r)[NR], double (&x)[NR], const int) const {
    const double zR = r[0] * winvR;
    double rr[NM];
    static_for<NM>([&](auto mc) { constexpr int m = decltype(mc)::value; rr[m] = r[1 + m]; });
    rr[0] = __builtin_fma(qC, zR, rr[0]);
    x[0] = zR;
    static_for<NM>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      double v = a[i][0] * rr[0];
      static_for<NM - 1>([&](auto jc) { constexpr int j = 1 + decltype(jc)::value; v = __builtin_fma(a[i][j], rr[j], v); });
      x[1 + i] = v;
    });
  }
  __device__ __forceinline__ bool coef_nonfinite(const Park& pk) const {
    bool nf = nonfinite(pk.ld(K_A)) || nonfinite(pk.ld(K_B)) || nonfinite(pk.ld(K_C));
    static_for<NM>([&](auto mc) { constexpr int m = decltype(mc)::value; nf = nf || nonfinite(pk.ld(K_DG + m)) || nonfinite(pk.ld(K_CI + m)); });
    return nf;
  }
  // ---- the interface sens_kernel uses
  Park pk; int S_;
  static constexpr size_t lds_doubles(int) { return (size_t)2 * NCOEF * 64; }
  __device__ __forceinline__ void init(lds_f64* lds, const int lane, const int, const int c, const double* __restrict__ th, const int n, const int S) {
    pk = Park{lds + lane}; S_ = S;
    build(pk, 0, [&](int i) { return th[i]; }, 1.0, n);
    build(pk, NCOEF, [&](int i) { return (i == c - 1) ? 1.0 : 0.0; }, 0.0, n);
  }
  __device__ __forceinline__ double cA() const { return pk.ld(K_A); }
  __device__ __forceinline__ double dA() const { return pk.ld(NCOEF + K_A); }
  // Y(ic) = element ic::value of the operand, out(ic, v) receives row ic::value of the product
  template <class YF, class OF>
  __device__ __forceinline__ void apply_base(YF&& Y, OF&& out, const double constA) const {
    double a[NR], f[NR];
    static_for<NR>([&](auto ic) { a[decltype(ic)::value] = Y(ic); });
    apply(pk, 0, 1.0, constA, a, f, S_);
    static_for<NR>([&](auto ic) { out(ic, f[decltype(ic)::value]); });
  }
  template <class YF, class OF>
  __device__ __forceinline__ void apply_deriv(YF&& W, OF&& out, const double constA) const {
    double a[NR], f[NR];
    static_for<NR>([&](auto ic) { a[decltype(ic)::value] = W(ic); });
    apply(pk, NCOEF, 0.0, constA, a, f, S_);
    static_for<NR>([&](auto ic) { out(ic, f[decltype(ic)::value]); });
  }
  __device__ __forceinline__ void factor(const double q) { factor(pk, q, S_); }
  template <class OF>
  __device__ __forceinline__ void solve(const double (&r)[NR], OF&& out) const {
    double x[NR];
    solve(r, x, S_);
    static_for<NR>([&](auto ic) { out(ic, x[decltype(ic)::value]); });
  }
  __device__ __forceinline__ bool coef_nonfinite() const { return coef_nonfinite(pk); }
};

// ---------------------------------------------------------------------------------------------- random model, n = NB = 4, 5 sites
// Here the inverse of M = I - q A (2^n x 2^n, dense after elimination) no longer fits one lane -- and it need not: it is the SAME matrix
// for all 1 + P columns.  The group builds it ONCE per step in LDS (cooperative Gauss-Jordan, 2^2n / GP elements per lane and pivot,
// wave-level ordering only) and every lane multiplies its own column by it, the matrix entries arriving as LDS broadcast reads.  This
// is where forward sensitivities beat differencing outright: the differenced Jacobian inverts the matrix once per step in EACH of its
// 1 + P replicas.  The derivative coefficients of a lane (d ci / d theta_c in {0, 1}, d dg / d theta_c in {0 .. n}) are packed into a
// few registers (one bit / one nibble per row); the coefficients of theta live once per group in LDS.
template <int NB, int GP_>
struct CubeLdsSys {
  static constexpr int NM = 1 << NB, NR = NM + 1, GP = GP_;
  static constexpr int MIN_WAVES = 1;
  static_assert(GP >= NM, "one lane per matrix row in the pivot-row / pivot-column update");
  static constexpr int GD = NM * NM + 2 * NM + 4;               // per group: inverse, dg[NM], ci[NM], cA cB cC (+ pad)
  static constexpr size_t lds_doubles(int NG) { return (size_t)NG * GD; }
  lds_f64* ainv; const lds_f64* dgp; const lds_f64* cip;
  double cA_, cB, cC, dA_, dBc, dCc, winvR, qC;
  uint32_t dcim, ddgw[(NM + 7) / 8];
  int c_;

  __device__ __forceinline__ void init(lds_f64* lds, const int, const int g, const int c, const double* __restrict__ th, const int, const int) {
    lds_f64* G = lds + g * GD;
    ainv = G; lds_f64* dgw = G + NM * NM; lds_f64* ciw = dgw + NM; lds_f64* abc = ciw + NM;
    dgp = dgw; cip = ciw; c_ = c;
    if (c < NM) {                                               // lane m builds row m (models/randmod.py:122-247, lowest-set-bit quirk at :201)
      const int m = c;
      if (m == 0) {
        double sumS = 0.0;
        for (int j = 0; j < NB; ++j) sumS += th[4 + j];
        dgw[0] = th[3] + sumS; ciw[0] = 0.0;
        abc[0] = th[0]; abc[1] = th[1]; abc[2] = th[2];
      } else {
        const int lsb = __builtin_ctz(m);
        ciw[m] = th[4 + lsb];
        double out = 0.0;
        for (int j = 0; j < NB; ++j) out += ((m >> j) & 1) ? 1.0 : th[4 + (j < lsb ? j : lsb)];
        dgw[m] = out + th[4 + NB + m - 1];
      }
    }
    wave_sync_lds();
    cA_ = abc[0]; cB = abc[1]; cC = abc[2];
    const int pidx = c - 1;                                     // -1 (base lane) and indices beyond P: all derivative coefficients zero
    dA_ = (pidx == 0) ? 1.0 : 0.0; dBc = (pidx == 1) ? 1.0 : 0.0; dCc = (pidx == 2) ? 1.0 : 0.0;
    dcim = 0;
    for (int w = 0; w < (NM + 7) / 8; ++w) ddgw[w] = 0;
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      uint32_t ddg = 0, dci = 0;
      if constexpr (m == 0) {
        ddg = (pidx >= 3 && pidx < 4 + NB) ? 1u : 0u;            // dg[0] = D + sum_j S_j
      } else {
        constexpr int lsb = __builtin_ctz(m);
        dci = (pidx == 4 + lsb) ? 1u : 0u;
        static_for<NB>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if constexpr (!((m >> j) & 1)) ddg += (pidx == 4 + (j < lsb ? j : lsb)) ? 1u : 0u;
        });
        ddg += (pidx == 4 + NB + m - 1) ? 1u : 0u;
      }
      dcim |= dci << m;
      ddgw[m / 8] |= ddg << (4 * (m % 8));
    });
  }
  __device__ __forceinline__ double cA() const { return cA_; }
  __device__ __forceinline__ double dA() const { return dA_; }
  template <int I> using IC = std::integral_constant<int, I>;
  template <class YF, class OF>
  __device__ __forceinline__ void apply_base(YF&& Y, OF&& out, const double constA) const {
    const double Y0 = Y(IC<0>{});
    out(IC<0>{}, __builtin_fma(-cB, Y0, constA));
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const double civ = cip[m];
      double v = -dgp[m] * Y(IC<1 + m>{});
      static_for<NB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (m & (1 << j)) v = __builtin_fma(civ, Y(IC<1 + (m ^ (1 << j))>{}), v);
        else v += Y(IC<1 + (m ^ (1 << j))>{});
      });
      if constexpr (m == 0) v = __builtin_fma(cC, Y0, v);
      out(IC<1 + m>{}, v);
    });
  }
  template <class YF, class OF>
  __device__ __forceinline__ void apply_deriv(YF&& W, OF&& out, const double constA) const {
    const double W0 = W(IC<0>{});
    out(IC<0>{}, __builtin_fma(-dBc, W0, constA));
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const double ddg = (double)((ddgw[m / 8] >> (4 * (m % 8))) & 15u);
      double v = -ddg * W(IC<1 + m>{});
      if constexpr (m != 0) {
        double lo = 0.0;
        static_for<NB>([&](auto jc) { constexpr int j = decltype(jc)::value; if constexpr (m & (1 << j)) lo += W(IC<1 + (m ^ (1 << j))>{}); });
        v += ((dcim >> m) & 1u) ? lo : 0.0;
      } else {
        v = __builtin_fma(dCc, W0, v);
      }
      out(IC<1 + m>{}, v);
    });
  }
  __device__ __forceinline__ void factor(const double q) {
    winvR = fast_rcp(__builtin_fma(q, cB, 1.0));
    qC = q * cC;
    wave_sync_lds();                                            // the previous step's products have read the old inverse
    // element (i, j) with j = c mod 2^n and i = c / 2^n + (GP / 2^n) m, m = 0 .. EL - 1: a lane stays in ONE column, so the pivot row
    // contributes one value per lane and pivot; the m-loops are unrolled (independent loads in flight)
    constexpr int EL = NM * NM / GP, RS = GP / NM;
    const int j = c_ & (NM - 1), ib = c_ >> NB;
    static_for<EL>([&](auto mc) {
      const int i = ib + RS * decltype(mc)::value, d = i ^ j;
      double v;
      if (d == 0) v = __builtin_fma(q, dgp[i], 1.0);
      else if ((d & (d - 1)) == 0) v = -q * ((i & d) ? cip[i] : 1.0);
      else v = 0.0;
      ainv[i * NM + j] = v;
    });
    wave_sync_lds();
#pragma unroll 1
    for (int k = 0; k < NM; ++k) {                              // in-place Gauss-Jordan inverse, no pivoting (M-matrix)
      const double rp = fast_rcp(ainv[k * NM + k]);
      const double akj = ainv[k * NM + j];
      double nv[EL];
      static_for<EL>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const int i = ib + RS * m;
        nv[m] = __builtin_fma(-(ainv[i * NM + k] * rp), akj, ainv[i * NM + j]);
      });
      static_for<EL>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const int i = ib + RS * m;
        // off the pivot row / column: the update; pivot column: -a_ik / a_kk; pivot row: a_kj / a_kk; pivot: 1 / a_kk
        double w = nv[m];
        if (i == k) w = (j == k) ? rp : akj * rp;
        else if (j == k) w = -(ainv[i * NM + k] * rp);
        nv[m] = w;
      });
      wave_sync_lds();                                          // every lane has read the pivot row and column
      static_for<EL>([&](auto mc) { constexpr int m = decltype(mc)::value; ainv[(ib + RS * m) * NM + j] = nv[m]; });
      wave_sync_lds();
    }
  }
  template <class OF>
  __device__ __forceinline__ void solve(const double (&r)[NR], OF&& out) const {
    const double zR = r[0] * winvR;
    out(IC<0>{}, zR);
    const double r0 = __builtin_fma(qC, zR, r[1]);              // the -q C z_R coupling of row P moved to the right-hand side
    static_for<NM / 4>([&](auto bc) {                           // four rows at a time: four independent FMA chains per pass over r
      constexpr int i0 = 4 * decltype(bc)::value;
      double v[4];
      static_for<4>([&](auto kc) { constexpr int k = decltype(kc)::value; v[k] = ainv[(i0 + k) * NM] * r0; });
      static_for<NM - 1>([&](auto jc) {
        constexpr int j = 1 + decltype(jc)::value;
        static_for<4>([&](auto kc) { constexpr int k = decltype(kc)::value; v[k] = __builtin_fma(ainv[(i0 + k) * NM + j], r[1 + j], v[k]); });
      });
      static_for<4>([&](auto kc) { constexpr int k = decltype(kc)::value; out(IC<1 + i0 + k>{}, v[k]); });
      __builtin_amdgcn_sched_barrier(0);                        // one block's loads in flight at a time
    });
  }
  // the same product with the row loop ROLLED, the result going to thread-private LDS slots (stride 64): for the sizes whose columns live
  // in LDS anyway -- one row's 2^n loads and FMAs in the code instead of 2^2n, registers bounded by one row
  __device__ __forceinline__ void solve_slots(const double (&r)[NR], lds_f64* xs) const {
    const double zR = r[0] * winvR;
    xs[0] = zR;
    const double r0 = __builtin_fma(qC, zR, r[1]);
#pragma unroll 1
    for (int i = 0; i < NM; i += 4) {                           // four rows per trip: four independent FMA chains hide the latency of one wave
      const lds_f64* row = ainv + i * NM;
      double v[4];
      static_for<4>([&](auto kc) { constexpr int k = decltype(kc)::value; v[k] = row[k * NM] * r0; });
      static_for<NM - 1>([&](auto jc) {
        constexpr int j = 1 + decltype(jc)::value;
        static_for<4>([&](auto kc) { constexpr int k = decltype(kc)::value; v[k] = __builtin_fma(row[k * NM + j], r[1 + j], v[k]); });
      });
      static_for<4>([&](auto kc) { constexpr int k = decltype(kc)::value; xs[(1 + i + k) * 64] = v[k]; });
    }
  }
  __device__ __forceinline__ bool coef_nonfinite() const {
    bool nf = nonfinite(cA_) || nonfinite(cB) || nonfinite(cC);
    for (int m = 0; m < NM; ++m) nf = nf || nonfinite(dgp[m]) || nonfinite(cip[m]);
    return nf;
  }
};

// a column of NR doubles owned by one lane: in registers, or (for the largest systems) in thread-private LDS slots
template <int NR, bool IN_LDS> struct Col;
template <int NR> struct Col<NR, false> {
  double v[NR];
  __device__ __forceinline__ explicit Col(lds_f64*) {}
  template <int I> __device__ __forceinline__ double get() const { return v[I]; }
  template <int I> __device__ __forceinline__ void set(double x) { v[I] = x; }
};
template <int NR> struct Col<NR, true> {
  lds_f64* p;
  __device__ __forceinline__ explicit Col(lds_f64* slots) : p(slots) {}
  template <int I> __device__ __forceinline__ double get() const { return p[I * 64]; }
  template <int I> __device__ __forceinline__ void set(double x) { p[I * 64] = x; }
};
template <class Sys> constexpr bool sens_cols_in_lds() { return Sys::NR > 24; }

template <class Sys, int GP> constexpr size_t sens_lds_bytes() {
  return (Sys::lds_doubles(64 / GP) + (size_t)(64 / GP) * 2 * Sys::NR + (sens_cols_in_lds<Sys>() ? (size_t)2 * Sys::NR * 64 : 0)) * sizeof(double);
}

// waves per SIMD the register allocation must leave room for: the kernel is a chain of dependent f64 operations (a lone wave keeps its SIMD
// busy 38 % of the time: profiles/r02_g_sens_*), so the small systems trade registers for a second / third resident wave
template <class Sys> constexpr int sens_min_waves() { return Sys::MIN_WAVES; }

template <class Sys, int GP>
__global__ __launch_bounds__(64, sens_min_waves<Sys>()) void sens_kernel(const SensArgs SA) {
  using Tab = ResolventTab<PK_METHOD_LRP12>;
  constexpr int NR = Sys::NR, NG = 64 / GP;
  const SolveArgs& A = SA.s;
  const int lane = threadIdx.x, g = lane / GP, c = lane % GP;
  long long rep = (long long)blockIdx.x * NG + g;
  const bool live = rep < A.B;                               // surplus groups of the last wave shadow the last replica and write nothing
  if (!live) rep = A.B - 1;
  const int n = A.n_sites, S = A.S, T = A.T, P = A.P, F = A.F;
  const bool is_base = (c == 0);
  const bool is_tan = (c >= 1 && c <= P);                    // lanes beyond 1 + P carry a zero column

  extern __shared__ __align__(16) double sens_lds[];
  lds_f64* const lds0 = (lds_f64*)sens_lds;
  lds_f64* const by = lds0 + Sys::lds_doubles(NG) + g * 2 * NR;              // the base lane's y_n ...  (ordering: wave_sync_lds, not volatile --
  lds_f64* const bz = by + NR;                                               // ... and its latest stage vector     volatile loads would each wait out the LDS latency)

  Sys sys;
  sys.init(lds0, lane, g, c, A.theta + rep * P, n, S);    // coefficients of theta (shared by the group) and of the unit vector e_c
  const double cA = sys.cA();                                 // b = cA e_R
  const double dA = sys.dA();                                 // b'_c

  const double* y0p = A.y0 + (A.y0_batched ? rep * S : 0);
  constexpr bool CL = sens_cols_in_lds<Sys>();                // y_n and the solve's output column leave the registers where 5 NR doubles do not fit
  lds_f64* const colmem = lds0 + Sys::lds_doubles(NG) + NG * 2 * NR + lane;
  Col<NR, CL> y(colmem), x(colmem + (size_t)NR * 64);
  static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; y.template set<i>((is_base && i < S) ? y0p[i] : 0.0); });
  auto publish_y = [&]() __attribute__((always_inline)) {
    wave_sync_lds();
    if (is_base) static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; by[i] = y.template get<i>(); });
    wave_sync_lds();
  };

  const int T5 = T > 5 ? T - 5 : 0;
  double* const fl = A.flat + rep * F;
  double* const dfl = SA.dflat + rep * (long long)F * P + (c - 1);
  auto emit = [&](const int k, const bool nan_fill) __attribute__((always_inline)) {
    static_for<NR>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (i < S && i < 2 + n) {
        const int fi = (i == 0) ? (k >= 5 ? k - 5 : -1) : (i == 1 ? T5 + k : T5 + T + (i - 2) * T + k);
        if (fi >= 0 && live) {
          const double sc = A.normalize ? 1.0 / y0p[i] : 1.0;
          const bool clipped = A.clip && (by[i] < 0.0);
          double r;
          if (nan_fill) r = __builtin_nan("");
          else r = clipped ? 0.0 : y.template get<i>() * sc;
          if (is_base) fl[fi] = r;
          else if (is_tan) dfl[(long long)fi * P] = r;
        }
      }
    });
  };
  auto finish = [&](const int status, const int acc, const int rej) {
    if (is_base && live) {
      if (A.status) A.status[rep] = status;
      if (A.n_steps) { A.n_steps[2 * rep] = acc; A.n_steps[2 * rep + 1] = rej; }
    }
  };
  auto fail_from = [&](int k) __attribute__((always_inline)) {
#pragma unroll 1
    for (; k < T; ++k) emit(k, true);
  };

  publish_y();
  emit(0, false);
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  if (T < 2) { finish(status, 0, 0); return; }

  const double rtol = A.rtol, atol = A.atol;
  auto norm = [&](auto&& ef, auto&& yaf, auto&& ybf) __attribute__((always_inline)) {
    double m = 0.0;
    static_for<NR>([&](auto ic) {
      const double q = fabs(ef(ic)) * approx_rcp(__builtin_fma(rtol, fmax(fabs(yaf(ic)), fabs(ybf(ic))), atol));
      m = (q > m || q != q) ? q : m;
    });
    return m;
  };
  auto y_at = [&](auto ic) __attribute__((always_inline)) { return y.template get<decltype(ic)::value>(); };

  double tc = A.t[0];
  int k = 1;
  double te = A.t[1];
  double h;
  {
    double f0[NR];
    sys.apply_base(y_at, [&](auto ic, double v) { f0[decltype(ic)::value] = v; }, cA);
    const double d0 = norm(y_at, y_at, y_at), d1 = norm([&](auto ic) { return f0[decltype(ic)::value]; }, y_at, y_at);
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
    if (A.h0 > 0.0) h = A.h0;
    if (!(h > 0.0) || h != h) h = 1e-6;
    h = bcast<GP, 0>(h);                                      // the base lane's estimate, for the whole group
  }

  const double* const kB = Tab::B;                            // indexed by the (uniform) round counter: scalar loads from constant memory
  const double* const kE = Tab::E;
  bool after_reject = false;
  while (true) {
    if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; fail_from(k); break; }
    const bool last = (tc + 1.0001 * h >= te);
    const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
    if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; fail_from(k); break; }
    sys.factor(Tab::GAM * hs);

    double z[NR], yn[NR], e[NR];
    static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; z[i] = 0.0; yn[i] = y.template get<i>(); e[i] = 0.0; });
    // round `it`: the base lane forms stage it + 1, the tangent lanes stage it (they need the base's z_it, published in round it - 1).
    // Rounds 0 and 1 are special (right-hand sides from y), rounds 2 .. NS are one rolled loop: a single copy of the solve in the code,
    // registers bounded by one round
    auto solve_to_x = [&](const double (&r)[NR]) __attribute__((always_inline)) {          // x = M^-1 r
      if constexpr (CL) sys.solve_slots(r, x.p);
      else sys.solve(r, [&](auto ic, double v) { x.template set<decltype(ic)::value>(v); });
    };
    auto finish_round = [&](const int it) __attribute__((always_inline)) {
      wave_sync_lds();                                          // every tangent lane has read bz
      if (is_base) static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; bz[i] = z[i]; });
      wave_sync_lds();
      const double bB = it < Tab::NS ? kB[it] : 0.0, eB = it < Tab::NS ? kE[it] : 0.0;
      const double bT = it >= 1 ? kB[it - 1] : 0.0, eT = it >= 1 ? kE[it - 1] : 0.0;
      const double bk = is_base ? bB : bT, ek = is_base ? eB : eT;
      static_for<NR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        yn[i] = __builtin_fma(bk, z[i], yn[i]);
        e[i] = __builtin_fma(ek, z[i], e[i]);
      });
    };
    {                                                           // round 0
      double r[NR];
      sys.apply_base(y_at, [&](auto ic, double v) { r[decltype(ic)::value] = is_base ? v * hs : 0.0; }, cA);
      solve_to_x(r);
      static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; z[i] = x.template get<i>(); });
      finish_round(0);
    }
    {                                                           // round 1
      double r[NR];
      sys.apply_base(y_at, [&](auto ic, double v) { r[decltype(ic)::value] = v; }, 0.0);                  // A y'_c
      sys.apply_deriv([&](auto ic) { constexpr int i = decltype(ic)::value; return __builtin_fma(Tab::GAM, bz[i], by[i]); },
                      [&](auto ic, double v) { constexpr int i = decltype(ic)::value; r[i] = is_base ? z[i] : hs * (r[i] + v); }, dA);   // + A'_c (y + g z_1) + b'_c
      solve_to_x(r);
      static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; z[i] = x.template get<i>(); });
      finish_round(1);
    }
    const double gh = is_base ? 0.0 : Tab::GAM * hs;
#pragma unroll 1
    for (int it = 2; it <= Tab::NS; ++it) {
      sys.apply_deriv([&](auto ic) { return bz[decltype(ic)::value]; },
                      [&](auto ic, double v) { constexpr int i = decltype(ic)::value; z[i] = __builtin_fma(gh, v, z[i]); }, 0.0);              // z'_it + g h A'_c z_it
      solve_to_x(z);
      static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; z[i] = x.template get<i>(); });
      finish_round(it);
    }

    double err = norm([&](auto ic) { return e[decltype(ic)::value]; }, y_at, [&](auto ic) { return yn[decltype(ic)::value]; });
    err = gmax<GP>(err, lane);
    if (err != err || err > 1e300) {
      ++nrej; after_reject = true; h = 0.1 * hs;
      bool nf = sys.coef_nonfinite();
      static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; nf = nf || nonfinite(by[i]); });
      if (nf) { status |= PK_ST_NONFINITE; fail_from(k); break; }
      continue;
    }
    double fac = root_q(err, Tab::Q) * (1.0 / 0.9);
    fac = fmax(1.0 / 6.0, fmin(5.0, fac));
    double hnew = hs * fast_rcp(fac);
    if (err <= 1.0) {
      ++nacc;
      static_for<NR>([&](auto ic) { constexpr int i = decltype(ic)::value; y.template set<i>(yn[i]); });
      publish_y();
      tc += hs;
      if (after_reject) hnew = fmin(hnew, hs);
      after_reject = false;
      if (last) {
        tc = te;
        emit(k, false);
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
  finish(status, nacc, nrej);
}

}  // namespace pk
