// pk_rand_level.hpp -- random model, n = 8 sites (257 states): rand_level_kernel<8>, EXACT resolvent solves, one workgroup per replica.
//
// Reference: models/randmod.py:122-247 (ode_system incl. the lowest-set-bit rate quirk at :201), solve_ode at :249-305; the reference's
// default model (config.toml:186) has no size limit.
//
// Why: beyond n = 7 the dense inverse of M = I - q J no longer fits a CU's registers, and the approximate factorisation of pk_wide.hpp pays
// for it with W-method step counts (500 .. 8 000 per replica) and the narrowest parity margin of the library.  But the matrix has
// structure: order the 2^n bit-mask states by POPCOUNT level.  Phosphorylation raises the level by one, dephosphorylation lowers it by one,
// degradation is diagonal -- so M is block tridiagonal over the levels 0 .. n with DIAGONAL diagonal blocks
//     row a of level l:   (1 + q dg_a) x_a  -  q ci_a  sum_{i in a} x_{a \ i}  -  q  sum_{j not in a} x_{a | j}  =  r_a
// and block elimination needs only the Schur complements of the levels: dense C(n,l) x C(n,l) matrices, sum_l C(n,l)^2 = C(2n,n)
// = 12 870 doubles = 103 KB at n = 8 -- they fit the 160 KB of LDS.  So the kernel runs the SAME LRP12 resolvent method as every exact
// kernel of the library (25 .. 60 steps for every parameter draw, twelve solves per factorisation):
//   factor   TWISTED block elimination: the chain from level 0 upwards and the chain from level n downwards advance together (levels l and
//            n - l have the same size) and meet in the middle level n / 2, whose Schur complement takes a correction from both sides.  Each
//            Schur complement is formed in registers (a 4 x 4 / 5 x 5 tile per thread of a 16 x 16 thread grid; its entries are q^2 times
//            short sums over the previous inverse: l^2 terms), inverted in place by Gauss-Jordan without pivoting (M-matrix; Schur
//            complements of M-matrices are M-matrices) with the pivot row / column published in LDS, double-buffered: ONE barrier per pivot,
//            1 + 8 + 28 + 56 + 70 = 163 pivots for both chains together instead of 256 -- and stored to LDS as explicit inverses.
//   solve    forward sweeps of both chains inwards (a sparse level-coupling gather + a dense block mat-vec per level), the middle level,
//            back-substitution outwards: 4 n/2 + 2 = 18 barrier phases per solve instead of the 2 n + ... of a one-sided sweep.
// The mRNA row is decoupled (lower triangular) and handled as a scalar, as in pk_rand_dense.hpp.  Controller, landing rule, outputs, fused
// metric and flags are those of the other workgroup-per-replica kernels (WideOut of pk_wide.hpp).
#pragma once
#include "pk_wide.hpp"

namespace pk {

constexpr int binom_c(int n, int k) { return (k < 0 || k > n) ? 0 : (k == 0 ? 1 : binom_c(n - 1, k - 1) * n / k); }

template <int NB>
struct LevelTab {
  static_assert(NB % 2 == 0 && NB >= 2 && NB <= 8, "twisted elimination pairs level l with n - l: even n; LDS holds n <= 8");
  static constexpr int NM = 1 << NB, MID = NB / 2;
  static constexpr int C(int l) { return binom_c(NB, l); }
  static constexpr int off(int l) { int s = 0; for (int i = 0; i < l; ++i) s += C(i); return s; }            // first position of level l
  static constexpr int blk(int l) { int s = 0; for (int i = 0; i < l; ++i) s += (C(i) * C(i) + 1) & ~1; return s; }   // first double of its inverse (even: 16-byte rows)
  static constexpr int TOTB = blk(NB + 1);
  static constexpr int CMAX = C(MID);
  static constexpr int CPAIR = MID > 0 ? C(MID - 1) : 1;                                                      // largest paired level
  static constexpr int TILE_MID = (CMAX + 15) / 16, TILE_PAIR = (CPAIR + 15) / 16;
};

// level geometry for a run-time level: a few scalar operations on a uniform argument (a private array indexed at run time would live in
// scratch memory; the recursive constexpr binomial must not be called at run time either)
struct LevelGeo { int C, off, blk; };
template <int NB>
__device__ __forceinline__ LevelGeo level_geo(const int l) {
  LevelGeo g{1, 0, 0};
  for (int i = 0; i < l; ++i) { g.off += g.C; g.blk += (g.C * g.C + 1) & ~1; g.C = g.C * (NB - i) / (i + 1); }
  return g;
}

template <int NB>
__host__ __device__ constexpr size_t rand_level_lds_doubles() {
  using L = LevelTab<NB>;
  // inverses | y yn u6 zs zd | dg ci | gp wv | pivot rows / columns: 2 chains x 2 buffers x (row, col) | prevv | red | pos_of, mask_at (int32 pairs)
  // + three byte tables [NM][NB]: ranks of the down / up neighbours of every position inside their level, masks of the up neighbours
  return (size_t)L::TOTB + 5 * (L::NM + 2) + 2 * L::NM + 2 * L::NM + 8 * L::CMAX + (2 + NB) + 24 + L::NM + (3 * L::NM * NB + 7) / 8;
}
template <int NB> __host__ __device__ constexpr size_t rand_level_lds_bytes() { return rand_level_lds_doubles<NB>() * sizeof(double); }

template <int NB>
__global__ __launch_bounds__(256) void rand_level_kernel(const SolveArgs A) {
  using Tab = ResolventTab<PK_METHOD_LRP12>;
  using L = LevelTab<NB>;
  constexpr int NM = L::NM, MID = L::MID, NT = 256;
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x, nt = NT, lane = tid & 63;
  const int ti = tid >> 4, tj = tid & 15;                     // 16 x 16 grid over the entries of a Schur complement
  const int S = A.S, T = A.T;
  const long long rep = blockIdx.x;
  if (rep >= A.B) return;
  const double* __restrict__ th = A.theta + rep * A.P;
  double* binv = lds;
  double* y = binv + L::TOTB;     double* yn = y + (NM + 2);      double* u6 = yn + (NM + 2);      // vectors padded to an even length:
  double* zs = u6 + (NM + 2);     double* zd = zs + (NM + 2);                                       // gp below stays 16-byte aligned
  double* dg = zd + (NM + 2);     double* ci = dg + NM;
  double* gp = ci + NM;           double* wv = gp + NM;           // gp: right-hand side of a block mat-vec, POSITION order; wv: forward results, MASK order
  double* piv = wv + NM;                                          // [chain][buffer][row | col][CMAX]
  double* prevv = piv + 8 * L::CMAX;   double* red = prevv + (2 + NB);
  int* pos_of = (int*)(red + 24);      int* mask_at = pos_of + NM;
  unsigned char* dnr = (unsigned char*)(mask_at + NM);           // [position][k]: rank inside level l - 1 of (mask minus its k-th set bit)
  unsigned char* upr = dnr + NM * NB;                            // [position][k]: rank inside level l + 1 of (mask plus its k-th clear bit)
  unsigned char* upm = upr + NM * NB;                            // [position][k]: that mask itself (its inflow rate ci enters the correction from above)
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
    // level tables: position = first position of the level + rank among the masks of equal popcount, in increasing mask order
    const int lv = __builtin_popcount(m);
    int rank = 0;
    for (int mm = 0; mm < m; ++mm) rank += (__builtin_popcount(mm) == lv) ? 1 : 0;
    const int o = level_geo<NB>(lv).off;
    pos_of[m] = o + rank; mask_at[o + rank] = m;
  }
  for (int row = tid; row < S; row += nt) y[row] = y0p[row];
  __syncthreads();
  if (tid < NM) {                                                // neighbour tables of position `tid` (needs pos_of / mask_at of every mask)
    const int p = tid, a = mask_at[p], lv = __builtin_popcount(a);
    const int od = level_geo<NB>(lv > 0 ? lv - 1 : 0).off, ou = level_geo<NB>(lv + 1).off;
    int kd = 0, ku = 0;
    for (int b = 0; b < NB; ++b) {
      if ((a >> b) & 1) dnr[p * NB + kd++] = (unsigned char)(pos_of[a ^ (1 << b)] - od);
      else { const int c = a | (1 << b); upr[p * NB + ku] = (unsigned char)(pos_of[c] - ou); upm[p * NB + ku] = (unsigned char)c; ++ku; }
    }
  }
  __syncthreads();
  WideOut out(A, rep, y0p, prevv, red);
  out.emit(0, y, false);
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  if (T < 2) { out.finish(status, 0, 0); return; }
  const double rtol = A.rtol, atol = A.atol;
  const int dbg = A.stage_form;            // dev timing switches (PK_LEVEL_DEBUG, host side): 1 skip pivots, 2 skip Schur sums, 4 skip solves, 8 skip mat-vecs

  auto rhs_into = [&](const double* Y, double* dst, const double scale) __attribute__((always_inline)) {       // f(Y) * scale (dst != Y); ends with a barrier
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
  auto err_norm = [&](const double* e, const double* ya, const double* yb) __attribute__((always_inline)) {
    auto mx = [](double p, double r) { return (p > r || p != p) ? p : r; };
    double m = 0.0;
    for (int row = tid; row < S; row += nt) m = mx(m, fabs(e[row]) / __builtin_fma(rtol, fmax(fabs(ya[row]), fabs(yb[row])), atol));
    return wg_max(m, red);
  };

  // ------------------------------------------------------------------------------------------------------------------------ factor
  // couplings seen from level l: "down" neighbours a \ i (level l - 1, entry -q ci_a), "up" neighbours a | j (level l + 1, entry -q)
  double winvR = 1.0, qC = 0.0;
  using Geo = LevelGeo;
  auto geo = [&](const int l) __attribute__((always_inline)) { return level_geo<NB>(l); };

  // entry (ia, ib) of the Schur complement of level l: the diagonal of M minus the fill of the eliminated neighbour level(s),
  //   from below:  (L Sinv U)[a][b] = q^2 ci_a sum_{i in a} sum_{j in b} Sinv_{l-1}[a \ i][b \ j]
  //   from above:  (U Sinv L)[a][b] = q^2 sum_{i not in a} sum_{j not in b} Sinv_{l+1}[a | i][b | j] ci_{b | j}
  // the neighbour ranks come from the byte tables, so the loops are uniform across the workgroup (l is)
  auto schur_entry = [&](const int l, const bool from_below, const bool from_above, const int ia, const int ib, const double q) __attribute__((always_inline)) {
    const Geo gl = geo(l);
    const int pa = gl.off + ia, pb = gl.off + ib;
    const int a = mask_at[pa];
    double v = (ia == ib) ? __builtin_fma(q, dg[a], 1.0) : 0.0;
    if (dbg & 2) return v;
    if (from_below && l > 0) {
      const Geo gm = geo(l - 1);
      const double* Bp = binv + gm.blk;
      const int Cp = gm.C;
      const unsigned char* ra = dnr + pa * NB; const unsigned char* rb = dnr + pb * NB;
      double s = 0.0;
      for (int i = 0; i < l; ++i) {
        const double* rowp = Bp + (int)ra[i] * Cp;
        for (int j = 0; j < l; ++j) s += rowp[rb[j]];
      }
      v = __builtin_fma(-(q * q) * ci[a], s, v);
    }
    if (from_above && l < NB) {
      const Geo gn = geo(l + 1);
      const double* Bn = binv + gn.blk;
      const int Cn = gn.C, cnt = NB - l;
      const unsigned char* ra = upr + pa * NB; const unsigned char* rb = upr + pb * NB; const unsigned char* mb = upm + pb * NB;
      double s = 0.0;
      for (int i = 0; i < cnt; ++i) {
        const double* rown = Bn + (int)ra[i] * Cn;
        for (int j = 0; j < cnt; ++j) s = __builtin_fma(rown[rb[j]], ci[mb[j]], s);
      }
      v = __builtin_fma(-(q * q), s, v);
    }
    return v;
  };

  // Gauss-Jordan inversion of NCH (1 or 2) C x C matrices held as TI x TI register tiles (rows ti + 16 ii, columns tj + 16 jj), in lockstep:
  // one barrier per pivot serves all of them.  Publishes the inverses to LDS (row-major C x C at dst[ch]).
  auto invert_tiles = [&](auto tile_c, auto nch_c, double (&a)[2][5][5], const int C, double* const (&dst)[2]) __attribute__((always_inline)) {
    constexpr int TI = decltype(tile_c)::value, NCH = decltype(nch_c)::value;
    auto rowbuf = [&](int ch, int p) __attribute__((always_inline)) { return piv + ((ch * 2 + p) * 2 + 0) * L::CMAX; };
    auto colbuf = [&](int ch, int p) __attribute__((always_inline)) { return piv + ((ch * 2 + p) * 2 + 1) * L::CMAX; };
    // publish pivot row / column 0
    static_for<NCH>([&](auto cc) {
      constexpr int ch = decltype(cc)::value;
      static_for<TI>([&](auto ic) {
        constexpr int ii = decltype(ic)::value;
        static_for<TI>([&](auto jc) {
          constexpr int jj = decltype(jc)::value;
          const int i = ti + 16 * ii, j = tj + 16 * jj;
          if (i < C && j < C) { if (i == 0) rowbuf(ch, 0)[j] = a[ch][ii][jj]; if (j == 0) colbuf(ch, 0)[i] = a[ch][ii][jj]; }
        });
      });
    });
    __syncthreads();
    // pivot k = 16 KQ + kr lives in tile row / column KQ (compile time) of the threads with ti == kr / tj == kr: every register index
    // below is static, the pivot-row / pivot-column special cases are selects on two per-thread flags, and the only branches are the two
    // publishing blocks (the NEXT pivot's row and column, into the other buffer)
    auto pivot_step = [&](auto kq_c, auto nkq_c, const int kr, const int nkr) __attribute__((always_inline)) {
      constexpr int KQ = decltype(kq_c)::value, NKQ = decltype(nkq_c)::value;
      const int k = 16 * KQ + kr, p = k & 1;
      const bool isr = (ti == kr), isc = (tj == kr), nxr = (ti == nkr), nxc = (tj == nkr);
      static_for<NCH>([&](auto cc) {
        constexpr int ch = decltype(cc)::value;
        const double* rb = rowbuf(ch, p); const double* cb = colbuf(ch, p);
        const double rp = fast_rcp(rb[k]);
        double rowv[TI], ml[TI];
        static_for<TI>([&](auto jc) { constexpr int jj = decltype(jc)::value; rowv[jj] = rb[tj + 16 * jj]; });      // beyond C: stale values, they only
        static_for<TI>([&](auto ic) { constexpr int ii = decltype(ic)::value; ml[ii] = cb[ti + 16 * ii] * rp; });   // reach padding entries of the tile
        static_for<TI>([&](auto ic) {
          constexpr int ii = decltype(ic)::value;
          static_for<TI>([&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            double v = __builtin_fma(-ml[ii], rowv[jj], a[ch][ii][jj]);
            if constexpr (ii == KQ && jj == KQ) v = isr ? (isc ? rp : rowv[jj] * rp) : (isc ? -ml[ii] : v);
            else if constexpr (ii == KQ) v = isr ? rowv[jj] * rp : v;            // pivot row: a_kj / a_kk
            else if constexpr (jj == KQ) v = isc ? -ml[ii] : v;                  // pivot column: -a_ik / a_kk
            a[ch][ii][jj] = v;
          });
        });
        if constexpr (NKQ < TI) {
          double* rbn = rowbuf(ch, p ^ 1); double* cbn = colbuf(ch, p ^ 1);
          if (nxr) static_for<TI>([&](auto jc) { constexpr int jj = decltype(jc)::value; const int j = tj + 16 * jj; if (j < C) rbn[j] = a[ch][NKQ][jj]; });
          if (nxc) static_for<TI>([&](auto ic) { constexpr int ii = decltype(ic)::value; const int i = ti + 16 * ii; if (i < C) cbn[i] = a[ch][ii][NKQ]; });
        }
      });
      __syncthreads();
    };
    if (!(dbg & 1)) {
      static_for<TI>([&](auto qc) {
        constexpr int KQ = decltype(qc)::value;
        const int kmax = C - 16 * KQ;                               // pivots left when this tile row starts (uniform)
        if (kmax > 0) {
#pragma unroll 1
          for (int kr = 0; kr < (kmax < 15 ? kmax : 15); ++kr) pivot_step(std::integral_constant<int, KQ>{}, std::integral_constant<int, KQ>{}, kr, kr + 1);
          if (kmax >= 16) pivot_step(std::integral_constant<int, KQ>{}, std::integral_constant<int, KQ + 1>{}, 15, 0);
        }
      });
    }
    static_for<NCH>([&](auto cc) {
      constexpr int ch = decltype(cc)::value;
      static_for<TI>([&](auto ic) {
        constexpr int ii = decltype(ic)::value;
        static_for<TI>([&](auto jc) {
          constexpr int jj = decltype(jc)::value;
          const int i = ti + 16 * ii, j = tj + 16 * jj;
          if (i < C && j < C) dst[ch][i * C + j] = a[ch][ii][jj];
        });
      });
    });
    __syncthreads();
  };

  auto factor = [&](const double q) __attribute__((always_inline)) {
    winvR = fast_rcp(__builtin_fma(q, cB, 1.0));
    qC = q * cC;
    double a[2][5][5];
#pragma unroll 1
    for (int k = 0; k < MID; ++k) {                              // paired levels k (chain from below) and NB - k (chain from above)
      const int lt = k, lb = NB - k, C = geo(k).C;
      static_for<L::TILE_PAIR>([&](auto ic) {
        constexpr int ii = decltype(ic)::value;
        static_for<L::TILE_PAIR>([&](auto jc) {
          constexpr int jj = decltype(jc)::value;
          const int i = ti + 16 * ii, j = tj + 16 * jj;
          const bool in = i < C && j < C;
          a[0][ii][jj] = in ? schur_entry(lt, true, false, i, j, q) : ((i == j) ? 1.0 : 0.0);
          a[1][ii][jj] = in ? schur_entry(lb, false, true, i, j, q) : ((i == j) ? 1.0 : 0.0);
        });
      });
      double* const dst[2] = {binv + geo(lt).blk, binv + geo(lb).blk};
      invert_tiles(std::integral_constant<int, L::TILE_PAIR>{}, std::integral_constant<int, 2>{}, a, C, dst);
    }
    {                                                            // the middle level: corrections from both chains
      constexpr int C = L::C(MID);
      static_for<L::TILE_MID>([&](auto ic) {
        constexpr int ii = decltype(ic)::value;
        static_for<L::TILE_MID>([&](auto jc) {
          constexpr int jj = decltype(jc)::value;
          const int i = ti + 16 * ii, j = tj + 16 * jj;
          a[0][ii][jj] = (i < C && j < C) ? schur_entry(MID, true, true, i, j, q) : ((i == j) ? 1.0 : 0.0);
        });
      });
      double* const dst[2] = {binv + L::blk(MID), nullptr};
      invert_tiles(std::integral_constant<int, L::TILE_MID>{}, std::integral_constant<int, 1>{}, a, C, dst);
    }
  };

  // ------------------------------------------------------------------------------------------------------------------------- solve
  // out(mask, value) receives  base(mask) + sum_b Binv_l[row][b] gp[off_l + b]  for every row of level l; `t` indexes the nth threads that
  // share the block: two threads per row (even / odd columns), partial sums joined by one DPP quad swap
  // gp holds the right-hand side of the block BLOCK-LOCALLY (entry b at gp[g0 + b]; g0 = 0 for the chain from below and the middle level,
  // GB0 for the chain from above): with an even C both the matrix rows and gp are 16-byte aligned and travel as ds_read_b128
  auto block_matvec = [&](const int l, const int g0, const int t, const int nth, auto&& out) __attribute__((always_inline)) {
    const Geo gl = geo(l);
    const int C = gl.C, o = gl.off;
    const double* Bm = binv + gl.blk;
    const double* gv = gp + g0;
    const int part = t & 1;
    if (dbg & 8) { for (int row = t; row < C; row += nth) out(mask_at[o + row], gv[row]); return; }
    for (int row = t >> 1; row < ((C + (nth >> 1) - 1) / (nth >> 1)) * (nth >> 1); row += nth >> 1) {     // whole pairs iterate together
      double s0 = 0.0, s1 = 0.0;
      if (row < C) {
        const double* br = Bm + row * C;
        if ((C & 1) == 0) {
          using d2 = double __attribute__((ext_vector_type(2)));
          const d2* b2 = (const d2*)br; const d2* g2 = (const d2*)gv;
          const int H = C >> 1;
          int c = part;
          for (; c + 6 < H; c += 8) {                               // four independent 16-byte pairs in flight
            const d2 m0 = b2[c], m1 = b2[c + 2], m2 = b2[c + 4], m3 = b2[c + 6];
            const d2 v0 = g2[c], v1 = g2[c + 2], v2 = g2[c + 4], v3 = g2[c + 6];
            s0 = __builtin_fma(m0.x, v0.x, s0); s1 = __builtin_fma(m0.y, v0.y, s1);
            s0 = __builtin_fma(m1.x, v1.x, s0); s1 = __builtin_fma(m1.y, v1.y, s1);
            s0 = __builtin_fma(m2.x, v2.x, s0); s1 = __builtin_fma(m2.y, v2.y, s1);
            s0 = __builtin_fma(m3.x, v3.x, s0); s1 = __builtin_fma(m3.y, v3.y, s1);
          }
          for (; c < H; c += 2) { const d2 m0 = b2[c], v0 = g2[c]; s0 = __builtin_fma(m0.x, v0.x, s0); s1 = __builtin_fma(m0.y, v0.y, s1); }
        } else {
          for (int c = part; c < C; c += 2) s0 = __builtin_fma(br[c], gv[c], s0);
        }
      }
      double s = s0 + s1;
      s += partner<1>(s, lane);
      if (row < C && part == 0) out(mask_at[o + row], s);
    }
  };
  // dst <- M^-1 src (mask order, row 0 = mRNA); ends with a barrier.  dst != src.
  auto solve = [&](const double* src, double* dst, const double q) __attribute__((always_inline)) {
    if (dbg & 4) { for (int row = tid; row < S; row += nt) dst[row] = 0.5 * src[row]; __syncthreads(); return; }
    const double zR = src[0] * winvR;
    if (tid == 0) dst[0] = zR;
    auto rhs_of = [&](const int a) __attribute__((always_inline)) { return (a == 0) ? __builtin_fma(qC, zR, src[1]) : src[1 + a]; };    // the -q C z_R coupling of mask 0 moved to the right
    const int half = tid >> 7, th_ = tid & 127;                 // threads 0..127: the chain from below, 128..255: the chain from above
    constexpr int GB0 = (L::CMAX + 1) & ~1;                     // block-local right-hand side of the chain from above
    const int g0 = half ? GB0 : 0;
#pragma unroll 1
    for (int k = 0; k < MID; ++k) {                              // forward sweeps inwards
      const int l = half ? NB - k : k;
      const Geo gl = geo(l);
      const int C = gl.C, o = gl.off;
      for (int i = th_; i < C; i += 128) {
        const int a = mask_at[o + i];
        double g = rhs_of(a);
        if (k > 0) {
          double s = 0.0;
          if (!half) {
#pragma unroll
            for (int b = 0; b < NB; ++b) if ((a >> b) & 1) s += wv[a ^ (1 << b)];
            g = __builtin_fma(q * ci[a], s, g);
          } else {
#pragma unroll
            for (int b = 0; b < NB; ++b) if (!((a >> b) & 1)) s += wv[a | (1 << b)];
            g = __builtin_fma(q, s, g);
          }
        }
        gp[g0 + i] = g;
      }
      __syncthreads();
      block_matvec(l, g0, th_, 128, [&](const int m, const double v) { wv[m] = v; });
      __syncthreads();
    }
    {                                                            // middle level: couplings to both chains
      constexpr int C = L::C(MID), o = L::off(MID);
      for (int i = tid; i < C; i += nt) {
        const int a = mask_at[o + i];
        double sd = 0.0, su = 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) { if ((a >> b) & 1) sd += wv[a ^ (1 << b)]; else su += wv[a | (1 << b)]; }
        gp[i] = __builtin_fma(q * ci[a], sd, __builtin_fma(q, su, rhs_of(a)));
      }
      __syncthreads();
      block_matvec(MID, 0, tid, nt, [&](const int m, const double v) { dst[1 + m] = v; });
      __syncthreads();
    }
#pragma unroll 1
    for (int k = MID - 1; k >= 0; --k) {                         // back-substitution outwards: x_l = w_l + Sinv_l (coupling to the solved inner level)
      const int l = half ? NB - k : k;
      const Geo gl = geo(l);
      const int C = gl.C, o = gl.off;
      for (int i = th_; i < C; i += 128) {
        const int a = mask_at[o + i];
        double s = 0.0;
        if (!half) {
#pragma unroll
          for (int b = 0; b < NB; ++b) if (!((a >> b) & 1)) s += dst[1 + (a | (1 << b))];
          gp[g0 + i] = q * s;
        } else {
#pragma unroll
          for (int b = 0; b < NB; ++b) if ((a >> b) & 1) s += dst[1 + (a ^ (1 << b))];
          gp[g0 + i] = q * ci[a] * s;
        }
      }
      __syncthreads();
      block_matvec(l, g0, th_, 128, [&](const int m, const double v) { dst[1 + m] = wv[m] + v; });
      __syncthreads();
    }
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
  auto fail_from = [&](int kk) __attribute__((always_inline)) { for (; kk < T; ++kk) out.emit(kk, y, true); };
  bool after_reject = false;
  while (true) {
    if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; fail_from(k); break; }
    const bool last = (tc + 1.0001 * h >= te);
    const double hs = last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h);
    if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; fail_from(k); break; }
    const double q = Tab::GAM * hs;
    factor(q);
    rhs_into(y, zs, hs);
    solve(zs, zd, q);
    { double* t_ = zs; zs = zd; zd = t_; }
    for (int row = tid; row < S; row += nt) { yn[row] = __builtin_fma(Tab::B[0], zs[row], y[row]); u6[row] = 0.0; }
#pragma unroll 1
    for (int kk = 1; kk < Tab::NS; ++kk) {
      solve(zs, zd, q);
      { double* t_ = zs; zs = zd; zd = t_; }
      const double bk = Tab::B[kk], ek = Tab::E[kk];
      for (int row = tid; row < S; row += nt) { yn[row] = __builtin_fma(bk, zs[row], yn[row]); u6[row] = __builtin_fma(ek, zs[row], u6[row]); }
    }
    __syncthreads();
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

namespace pk {
constexpr bool rand_level_available(int n_sites) { return n_sites == 8 || n_sites == 6; }
template <int NB>
static hipError_t launch_rand_level_one(const SolveArgs& a, hipStream_t st) {
  constexpr size_t lds = rand_level_lds_bytes<NB>();
  static_assert(lds <= 160 * 1024, "one workgroup's LDS");
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)rand_level_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((rand_level_kernel<NB>), dim3((unsigned)a.B), dim3(256), lds, st, a);
  return hipGetLastError();
}
}  // namespace pk
