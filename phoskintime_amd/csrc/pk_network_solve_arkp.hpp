// pk_network_solve_arkp.hpp -- [r3] the order-4 additive network integrator (ARK4(3)6L[2]SA, pk_network_solve_ark.hpp) in a DENSE lane
// layout: arrow topologies (distributive 0, saturating 4), a protein's block split over at most TWO adjacent lanes.
//
// Why: round 2's kernel (one thread per protein, 2 + MAXS rows each) is latency-bound, not issue-bound.  Measured on config 5 (8 192
// candidates, N = 100, <= 6 sites; tools/gpu_net5_time.py with PK_ARK_LDS_PAD): 149.6 / 189.7 / 270.7 / 503.1 ms at 4 / 3 / 2 / 1 resident
// workgroups per CU -- throughput is proportional to the waves in flight; a lone workgroup spends ~4 000 clocks per stage of which ~750
// issue VALU work.  Its 8-row block vectors fill all 256 VGPRs (28 spilled) AND need 32 KB of LDS for the stage right-hand sides R_3..R_6,
// whose load -> fma -> store chains (75 LDS operations, 40 s_waitcnt per stage; ds_write_b64 costs ~6 clocks each) are the exposed latency.
//
// Here a lane owns NRL = (2 + site class) / 2 rows: lane A = [mRNA, protein, first NRL - 2 sites], lane B = [the next NRL sites].
// Proteins with <= NRL - 2 sites take ONE lane (no idle partner): N = 100 with 1..6 sites -> ~167 lanes = 3 waves, against the 2 waves x 8
// rows (28 idle lanes, 31 % zero rows) before.  Half the rows per lane -> every block vector of the method (y, Y, hF, hG, g r, both running
// sums AND the four parked right-hand sides) lives in registers: no LDS round trip left in a stage but the TF gather of P_vec.
// The arrow structure needs two exchanges per solve / product between the lanes of a pair (sum of the site terms: quad_perm [1,0,3,2];
// the protein row's value: quad_perm [0,0,2,2]) -- 2 DPP moves each, VALU only.
//
// Same arithmetic as net_solve_ark_kernel up to the order of the site sums; same controller, stops, status and output conventions.
#pragma once
#include "pk_network_solve_ark.hpp"

namespace pk {

// lane table entry: (protein << 2) | (paired << 1) | half.  Pairs sit on (even, odd) lanes, single-lane proteins after them.
__host__ __device__ constexpr int arkp_rows_per_lane(int site_class) { return (2 + site_class) / 2; }
__host__ __device__ constexpr int arkp_park_stride(int rows_per_lane) { return (5 * rows_per_lane + 3) | 1; }  // doubles per thread, odd

// PARK: per-row constants (loss coefficient, site rate, the two solve factors) and the two right-hand sides that are needed last (R_5, R_6)
// live in LDS, thread-private [slot][thread]; the step-size scalars are moved to SGPRs.  That is the register diet for 3 waves per SIMD
// (168 VGPRs): 4 workgroups of 3 waves per CU instead of 2, every SIMD equally loaded.
template <int MODEL, int NRL, bool PARK>
__device__ __forceinline__ void net_solve_arkp_body(const NetDev& n, const NetSolveArgs& A) {
  using namespace ark436;
  static_assert(MODEL == 0 || MODEL == 1 || MODEL == 4, "arrow topologies (0, 4) and the sequential chain (1)");
  constexpr bool CHAIN = MODEL == 1;
  constexpr int NL = NRL - 1;                         // a lane's last row: the interface to its partner in the chain layout
  extern __shared__ __align__(16) double lds[];
  const int N = n.N, S = n.S;
  double* Kt = lds;                       // [n_K]
  double* Pv = Kt + n.n_K;                // [2][N]
  double* red = Pv + 2 * N;               // [24]
  // PARK: [thread][slot] with an ODD slot count per thread: one address VGPR + immediate offsets for any block size, and the lanes of a
  // ds_read_b64 / ds_write_b64 group fall on distinct banks (2 * odd * lane mod 64)
  // arrow: loss coefficient, site rate, solve factor, R_5, R_6;  chain: lower / upper coupling, loss coefficient, multiplier, inverse pivot
  // (its R_5, R_6 stay in registers: it has fewer scalars to keep)
  constexpr int P_LK = 0, P_SR = NRL, P_WV = 2 * NRL, P_R5 = 3 * NRL, P_R6 = 4 * NRL, P_LOSS = 5 * NRL, P_STRIDE = arkp_park_stride(NRL);
  constexpr int P_CA = NRL, P_CB = 2 * NRL, P_CM = 3 * NRL, P_CD = 4 * NRL;
  double* rbase = red + 24;               // PARK: [N] the mRNA baselines of the fused objective (captured at the rna baseline's output time)
  double* const mypark = rbase + (PARK ? N : 0) + (size_t)threadIdx.x * P_STRIDE;
  auto pld = [&](int slot) __attribute__((always_inline)) { return mypark[slot]; };
  auto pst = [&](int slot, double x) __attribute__((always_inline)) { mypark[slot] = x; };
  auto uni = [](double x) __attribute__((always_inline)) {              // a block-uniform value into SGPRs
    if constexpr (!PARK) return x;
    else return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
  };
  const double* tf_dat = n.TF_data;       // rows beyond the register-resident entries (rare): straight from HBM / L2
  const int32_t* tf_idx = n.TF_indices;
  const NetSlices sl(n.n_K, N, n.sites);
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const bool own = tid < n.n_lanes;
  const int unit = own ? n.lane_unit[tid] : 0;
  const int i = unit >> 2;
  const bool paired = own && (unit & 2), hb = own && (unit & 1), la = own && !(unit & 1);      // hb: second lane of a pair; la: lane A (or single)
  const double* stops = A.stops_p ? A.stops_p : A.stops_v;
  const int32_t* stop_out = A.stop_out_p ? A.stop_out_p : A.stop_out_v;
  const double* xb = A.x + b * n.n_var;
  auto par = [&](int off) __attribute__((always_inline)) { const double v = xb[off]; return A.x_is_raw ? softplus(v) : v; };

  const int st = own ? n.offset_y[i] : 0, ss = own ? n.offset_s[i] : 0, ns = own ? n.n_sites[i] : 0, drv = la ? n.driver_map[i] : -1;
  // TF row of the protein: the lanes of a pair take alternate entries; the first TFC entries of a lane live in registers (static topology),
  // so the gather of P_vec is TFC independent LDS reads -- the serial index -> value chain of a CSR loop (two LDS latencies per entry, up
  // to 9 entries) was the longest dependent chain of a stage
  constexpr int TFC = 4;
  const int tfs = paired ? 2 : 1, tfb = own ? n.TF_indptr[i] + (hb ? 1 : 0) : 0, tf1 = own ? n.TF_indptr[i + 1] : 0;
  int tix[TFC]; double tdt[TFC];
#pragma unroll
  for (int c = 0; c < TFC; ++c) {
    const int e = tfb + c * tfs;
    const bool ok = e < tf1;
    tix[c] = ok ? n.TF_indices[e] : 0; tdt[c] = ok ? n.TF_data[e] : 0.0;
  }
  const int tf0 = tfb + TFC * tfs;                    // remainder (rare): from the LDS copy of the CSR arrays
  const double tfdeg_inv = la ? 1.0 / n.tf_deg[i] : 1.0;
  const double Ai = la ? par(sl.A + i) : 0.0, Bi = la ? par(sl.B + i) : 1.0, Ci = la ? par(sl.C + i) : 0.0, Di = own ? par(sl.D + i) : 1.0,
               Ei = own ? par(sl.E + i) : 0.0, ts = uni(par(sl.tf));
  // rows of this lane: site index (or -1), loss coefficient E + Dp + D of a site row, validity
  int sj[NRL]; bool valid[NRL]; double Lk[NRL], Sr[NRL];
#pragma unroll
  for (int k = 0; k < NRL; ++k) {
    // lane B holds the next NRL sites; in the chain layout in REVERSE order, so that both lanes sweep towards their common interface
    const int j = hb ? (CHAIN ? 2 * NRL - 3 - k : NRL - 2 + k) : (k - 2);
    const bool site = own && j >= 0 && j < ns;
    sj[k] = site ? j : -1;
    valid[k] = site || (la && k < 2);
    // loss coefficient of the row's own state: E + Dp + D for a site; lane A's rows 0 / 1 (mRNA, protein) carry B and D, so the generic
    // row formulas  f_k = S_k q - L_k Y_k,  x_k = r_k / (g + L_k) + ...  give the mRNA row outright and the diagonal part of the protein row
    Lk[k] = site ? Ei + par(sl.Dp + ss + j) + Di : (la && k == 0) ? Bi : (la && k == 1) ? Di : 0.0;
    Sr[k] = 0.0;
    if constexpr (PARK) { pst(P_LK + k, Lk[k]); pst(P_SR + k, 0.0); }
  }
  // per-row constants and factors: registers, or (PARK) their LDS slots
  auto LK = [&](int k) __attribute__((always_inline)) { if constexpr (PARK) return pld(P_LK + k); else return Lk[k]; };
  auto SR = [&](int k) __attribute__((always_inline)) { if constexpr (PARK) return pld(P_SR + k); else return Sr[k]; };
  auto yoff = [&](int k) __attribute__((always_inline)) { return st + ((la && k < 2) ? k : 2 + sj[k]); };
  const double mP = paired ? 1.0 : 0.0, mB = hb ? 1.0 : 0.0, mA = la ? 1.0 : 0.0;
  auto mB_ = [&](double v) __attribute__((always_inline)) { return mB * v; };
  auto pair_sum_ = [&](double v) __attribute__((always_inline)) { return __builtin_fma(mP, dpp_mov<0xB1>(v), v); };
  const double* y0 = A.y0 + (A.y0_batched ? b * S : 0);
  double* Yout = A.Y + b * (size_t)A.T * S;
  double y[NRL];
#pragma unroll
  for (int k = 0; k < NRL; ++k) y[k] = valid[k] ? y0[yoff(k)] : 0.0;
  // fused objective (PARK kernel only): at every output time each lane scores the observations of its own states -- mRNA row: rna fold
  // change, protein row: total protein (P + all sites; the pair's site sum by DPP), site rows: phospho -- against the dense [T, S] tables,
  // into three thread-private partial sums parked in LDS (protein, rna, phospho).  Baselines: the initial state for protein / phospho (time
  // index 0); the mRNA value at output row loss_rna_base, kept in LDS by the lane that owns it (observations are never earlier)
  const bool fuse = PARK && A.loss_obs != nullptr;
  bool ybad = false;                                                       // a lane mask in SGPRs: no VGPR across the step loop
  if constexpr (PARK) { pst(P_LOSS, 0.0); pst(P_LOSS + 1, 0.0); pst(P_LOSS + 2, 0.0); }
  auto score_row = [&](int row) __attribute__((always_inline)) {
    double b0[NRL];
#pragma unroll
    for (int k = 0; k < NRL; ++k) b0[k] = valid[k] ? y0[yoff(k)] : 0.0;
    double part = mB_(y[0]), bpart = mB_(b0[0]);
#pragma unroll
    for (int k = 1; k < NRL; ++k) { part += (k >= 2) ? y[k] : mB_(y[k]); bpart += (k >= 2) ? b0[k] : mB_(b0[k]); }
    const double stot = pair_sum_(part), btot = pair_sum_(bpart);
    if (la && row == A.loss_rna_base) rbase[i] = y[0];
    double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < NRL; ++k) {
      if (!valid[k]) continue;
      const size_t at = (size_t)row * S + yoff(k);
      const double wgt = A.loss_w[at];
      const bool prot = la && k == 1, rna = la && k == 0;
      if (wgt != 0.0) {
        const double obs = A.loss_obs[at];
        const double pred = fold_change(prot ? y[1] + stot : y[k], prot ? b0[1] + btot : rna ? rbase[i] : b0[k]);
        const double l = wgt * point_loss(A.loss_mode, obs - pred, obs, pred);
        acc[prot ? 0 : rna ? 1 : 2] += l;
      }
      if (nonfinite(y[k])) ybad = true;                                     // np.all(np.isfinite(Y)) of the reference (optproblem.py:130)
    }
    pst(P_LOSS, pld(P_LOSS) + acc[0]); pst(P_LOSS + 1, pld(P_LOSS + 1) + acc[1]); pst(P_LOSS + 2, pld(P_LOSS + 2) + acc[2]);
  };
  auto write_row = [&](int row) __attribute__((always_inline)) {
    if (A.Y) {
      double* o = Yout + (size_t)row * S;
#pragma unroll
      for (int k = 0; k < NRL; ++k) if (valid[k]) o[yoff(k)] = y[k];
    }
    if constexpr (PARK) if (fuse) score_row(row);
  };

  // pair exchanges (executed by every lane, convergently): sum over the two lanes of a protein; lane A's value seen by both lanes
  // (masks as 0 / 1 factors instead of selects: one fma where a select costs two v_cndmask per double; a stray partner value is finite --
  // every lane of the workgroup belongs to the same candidate, and a non-finite one fails the step's error test anyway)
  auto pair_sum = [&](double v) __attribute__((always_inline)) { return __builtin_fma(mP, dpp_mov<0xB1>(v), v); };
  write_row(0);
  auto from_a = [&](double v) __attribute__((always_inline)) { const double o = dpp_mov<0xA0>(v); return hb ? o : v; };

  double sumS = 0.0;
  double ca[NRL], cb[NRL], cm[NRL], cdi[NRL], cI = 0.0, denI = 1.0;      // chain layout only
  auto CA = [&](int k) __attribute__((always_inline)) { if constexpr (PARK) return pld(P_CA + k); else return ca[k]; };
  auto CB = [&](int k) __attribute__((always_inline)) { if constexpr (PARK) return pld(P_CB + k); else return cb[k]; };
  auto CM = [&](int k) __attribute__((always_inline)) { if constexpr (PARK) return pld(P_CM + k); else return cm[k]; };
  auto CD = [&](int k) __attribute__((always_inline)) { if constexpr (PARK) return pld(P_CD + k); else return cdi[k]; };
  auto set_bucket = [&](const int jb) __attribute__((always_inline)) {
    __syncthreads();
    for (int k = tid; k < n.n_K; k += nt) Kt[k] = n.kin_Kmat[(size_t)k * n.n_grid + jb] * (A.x_is_raw ? softplus(xb[sl.ck + k]) : xb[sl.ck + k]);
    __syncthreads();
    double acc_all = 0.0;
#pragma unroll
    for (int k = 0; k < NRL; ++k) {
      double acc = 0.0;
      if (sj[k] >= 0) for (int q = n.W_indptr[ss + sj[k]]; q < n.W_indptr[ss + sj[k] + 1]; ++q) acc += n.W_data[q] * Kt[n.W_indices[q]];
      if constexpr (CHAIN) Sr[k] = acc;
      else { if constexpr (PARK) pst(P_SR + k, acc); else Sr[k] = acc; }
      acc_all += acc;
    }
    if constexpr (!CHAIN) sumS = pair_sum(acc_all);
    else {
      // sequential chain (models.py sequential_rhs): row of site j:  S_j [previous form] + E [next form] - (S_{j+1} + E + Dp_j + D) [own];
      // protein row: C R + E s_0 - (D + S_0) P.  Per row in SWEEP order: ca = coupling to the row before, cb = to the row after (the last
      // row's: to the partner's last row), L = own loss.  Invalid rows: all zero (they must not feed their neighbours' pivots)
      const double sr_o = mP * dpp_mov<0xB1>(Sr[NL]);                     // the partner's last row's site rate (lane A's last site needs S_{j+1})
#pragma unroll
      for (int k = 0; k < NRL; ++k) {
        double a_, b_, l_;
        const double lbase = (sj[k] >= 0) ? Ei + par(sl.Dp + ss + sj[k]) + Di : 0.0;
        if (hb) {                                                         // sites in reverse order: the row before is the NEXT site
          a_ = (k >= 1 && valid[k]) ? Ei : 0.0;
          b_ = Sr[k];
          l_ = lbase + (k >= 1 ? Sr[k - 1] : 0.0);
        } else {
          a_ = (k == 1) ? Ci : (k >= 2 ? Sr[k] : 0.0);
          b_ = (k >= 1 && valid[k]) ? (k == NL ? mP * Ei : Ei) : 0.0;
          l_ = (k == 0) ? Bi : (k == 1) ? Di + Sr[2 < NRL ? 2 : NL] : lbase + (k < NL ? Sr[k + 1 < NRL ? k + 1 : NL] : sr_o);
        }
        if (!valid[k]) { a_ = 0.0; b_ = 0.0; l_ = 0.0; }
        if constexpr (PARK) { pst(P_CA + k, a_); pst(P_CB + k, b_); pst(P_LK + k, l_); } else { ca[k] = a_; cb[k] = b_; Lk[k] = l_; }
      }
    }
  };

  // frozen block Jacobian of the step (entries that depend on y_n) and the factors of g I - A
  double cRv = 0.0, gPv = 1.0, sinv = 1.0, wv[NRL], cw[NRL];
  auto WV = [&](int k) __attribute__((always_inline)) { if constexpr (PARK) return pld(P_WV + k); else return wv[k]; };
  auto CW = [&](int k) __attribute__((always_inline)) {                 // (S_k gP) w_k: kept (registers) or rebuilt from its parked factors
    if constexpr (!PARK) return cw[k];
    else if constexpr (MODEL == 4) return (pld(P_SR + k) * gPv) * pld(P_WV + k);
    else return pld(P_SR + k) * pld(P_WV + k);
  };
  int buf = 0;
  // f(Y) of the block -> f; with MATVEC also G = A Y (stage 1: Y = y_n).  One barrier.
  auto rhs_block = [&](const double (&Y)[NRL], double (&f)[NRL], double (&G)[NRL], auto mv) __attribute__((always_inline)) {
    constexpr bool MATVEC = decltype(mv)::value;
    double part = mB * Y[0];                            // rows 0, 1 are sites in the second lane only
#pragma unroll
    for (int k = 1; k < NRL; ++k) part = (k >= 2) ? part + Y[k] : __builtin_fma(mB, Y[k], part);
    const double stot = pair_sum(part);                 // sum of the protein's phospho states
    const double Pb = from_a(Y[1]);                     // its protein state, in both lanes
    if (la) Pv[buf * N + i] = (drv >= 0) ? Kt[drv] : Y[1] + stot;
    __syncthreads();
    const double* Pb_ = Pv + buf * N;
    double pv[TFC];
#pragma unroll
    for (int c = 0; c < TFC; ++c) pv[c] = Pb_[tix[c]];
    if constexpr (TFC == 4) asm volatile("" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]));      // all reads in flight before the first use
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < TFC; ++c) acc = __builtin_fma(tdt[c], pv[c], acc);
    for (int e = tf0; e < tf1; e += tfs) acc += tf_dat[e] * Pb_[tf_idx[e]];                              // degree > TFC per lane: rare (host pairs by degree)
    acc = pair_sum(acc);
    buf ^= 1;
    // synthesis rate (calculate_synthesis_rate on the squashed TF input; models 0 / 1 / 2 squash twice: s(s(v)) = v / (1 + 2 |v|)), with ONE
    // reciprocal chain per squash / rate instead of one per branch: den = 1 + u + 1e-6 (u >= 0) or 1 + ts |u| (u < 0)
    const double synth = synth_rate_squashed(Ai, ts, acc * tfdeg_inv, MODEL != 4);
    if constexpr (CHAIN) {
      // the chain is linear in the block: f = A Y + synth e_R, A tridiagonal across the two lanes (the rows after a lane's last one is the
      // partner's last row)
      const double Yo = dpp_mov<0xB1>(Y[NL]);
#pragma unroll
      for (int k = 0; k < NRL; ++k) {
        double lin = -LK(k) * Y[k];
        if (k >= 1) lin = __builtin_fma(CA(k), Y[k - 1 >= 0 ? k - 1 : 0], lin);
        lin = __builtin_fma(CB(k), (k < NL) ? Y[k + 1 < NRL ? k + 1 : NL] : Yo, lin);
        if constexpr (MATVEC) G[k] = lin;
        f[k] = (k == 0) ? __builtin_fma(mA, synth, lin) : lin;
      }
      return;
    }
    // lane A: row 0 = synth - B R (the generic row with S = 0, L = B, plus synth); row 1 = generic (-D P) + [C R - sumS q + E sum(sites)]
    double q = Pb, eP;
    if (MODEL == 4) {
      q = Pb * net_rcp(1.0 + Pb);
      eP = __builtin_fma(Ei, stot, __builtin_fma(-sumS, q, (Ci * Y[0]) * net_rcp(1.0 + Y[0])));
    } else {
      eP = __builtin_fma(Ei, stot, __builtin_fma(-sumS, Y[1], Ci * Y[0]));
    }
#pragma unroll
    for (int k = 0; k < NRL; ++k) {
      const double fs = SR(k) * q - LK(k) * Y[k];
      f[k] = (k == 0) ? __builtin_fma(mA, synth, fs) : (k == 1) ? __builtin_fma(mA, eP, fs) : fs;
    }
    if constexpr (MATVEC) {
      const double gP_ = __builtin_fma(Ei, stot, __builtin_fma(-(sumS * gPv), Y[1], cRv * Y[0]));
#pragma unroll
      for (int k = 0; k < NRL; ++k) {
        const double gs = (SR(k) * gPv) * Pb - LK(k) * Y[k];
        G[k] = (k == 1) ? __builtin_fma(mA, gP_, gs) : gs;
      }
    }
  };
  auto freeze_factor = [&](const double g) __attribute__((always_inline)) {
    if constexpr (CHAIN) {
      // twisted factorisation: each lane eliminates along its own sweep (lane A downwards from the mRNA row, lane B -- reversed rows --
      // upwards from the last site); the two last rows meet in a 2 x 2 system whose inverse pivot replaces the last row's
      double dprev = 0.0, dl = 1.0;
#pragma unroll
      for (int k = 0; k < NRL; ++k) {
        const double m = (k >= 1) ? CA(k) * dprev : 0.0;
        const double d = (k >= 1) ? __builtin_fma(-m, CB(k - 1 >= 0 ? k - 1 : 0), g + LK(k)) : g + LK(k);
        dprev = net_rcp(d);
        if (k == NL) dl = d;
        if constexpr (PARK) { pst(P_CM + k, m); if (k < NL) pst(P_CD + k, dprev); } else { cm[k] = m; if (k < NL) cdi[k] = dprev; }
      }
      const double di_o = dpp_mov<0xB1>(dprev), b_o = dpp_mov<0xB1>(CB(NL));
      cI = CB(NL) * di_o;                                                 // 0 for a single lane (its interface coupling is 0)
      denI = net_rcp(__builtin_fma(-cI, b_o, dl));
      return;
    }
    const bool sat = MODEL == 4;
    const double Pb = from_a(y[1]);
    gPv = sat ? net_rcp((1.0 + Pb) * (1.0 + Pb)) : 1.0;
    cRv = sat ? Ci * net_rcp((1.0 + y[0]) * (1.0 + y[0])) : Ci;
    double part = 0.0;
#pragma unroll
    for (int k = 0; k < NRL; ++k) {                     // Sr = 0 on the mRNA / protein rows: they drop out of the sums by themselves
      double w = net_rcp(g + LK(k));                    // lane A, row 0: 1 / (g + B), the mRNA pivot
      if (k == 1) w = la ? 0.0 : w;                     // lane A, row 1: the protein row is solved through the Schur pivot below, not here
      const double c = (SR(k) * gPv) * w;
      if constexpr (PARK) pst(P_WV + k, w); else { wv[k] = w; cw[k] = c; }
      part += c;
    }
    sinv = net_rcp(g + Di + sumS * gPv - Ei * pair_sum(part));
  };
  // x = (g I - A)^-1 r
  auto block_solve = [&](const double (&r)[NRL], double (&x)[NRL]) __attribute__((always_inline)) {
    if constexpr (CHAIN) {
      double rp[NRL];
      rp[0] = r[0];
#pragma unroll
      for (int k = 1; k < NRL; ++k) rp[k] = __builtin_fma(CM(k), rp[k - 1], r[k]);
      const double rpo = dpp_mov<0xB1>(rp[NL]);
      x[NL] = __builtin_fma(cI, rpo, rp[NL]) * denI;
#pragma unroll
      for (int k = NL - 1; k >= 0; --k) x[k] = CD(k) * __builtin_fma(CB(k), x[k + 1], rp[k]);
      return;
    }
    double t[NRL];
#pragma unroll
    for (int k = 0; k < NRL; ++k) t[k] = r[k] * WV(k);
    double part = mB * t[0];
#pragma unroll
    for (int k = 1; k < NRL; ++k) part = (k >= 2) ? part + t[k] : __builtin_fma(mB, t[k], part);
    const double xP = (r[1] + cRv * t[0] + Ei * pair_sum(part)) * sinv;   // meaningful in lane A (t[0] = the mRNA row's solution there)
    const double xPb = from_a(xP);
#pragma unroll
    for (int k = 0; k < NRL; ++k) {
      const double xs = __builtin_fma(CW(k), xPb, t[k]);                  // lane A: rows 0 / 1 have CW = 0, row 1 also t = 0
      x[k] = (k == 1) ? __builtin_fma(mA, xP, xs) : xs;
    }
  };

  __syncthreads();
  int status = PK_ST_OK, nacc = 0, nrej = 0;
  double tc = A.t0;
  int jb = net_bucket(tc, n.kin_grid, n.n_grid);
  set_bucket(jb);
  double h;
  {
    double f[NRL], dummy[NRL];
    rhs_block(y, f, dummy, std::false_type{});
    auto q = [&](double v, double yv) { return fabs(v) / (A.atol + A.rtol * fabs(yv)); };
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int k = 0; k < NRL; ++k) if (valid[k]) { d0 = fmax(d0, q(y[k], y[k])); d1 = fmax(d1, q(f[k], y[k])); }
    d0 = block_max(d0, red); d1 = block_max(d1, red);
    h = (d0 > 1e-5 && d1 > 1e-5) ? 0.01 * d0 / d1 : 1e-6;
    if (A.h0 > 0.0) h = A.h0;
    if (!(h > 0.0) || h != h) h = 1e-6;
    h = uni(h);
  }
  const bool rms = A.err_rms;
  const double safety_inv = 1.0 / (A.ctl_safety > 0.0 ? A.ctl_safety : 0.9), grow_inv = 1.0 / (A.ctl_grow > 1.0 ? A.ctl_grow : 6.0);
  bool after_reject = false;
  for (int si = 0; si < A.n_stops && status == PK_ST_OK; ++si) {
    const double te = stops[si];
    while (true) {
      if (nacc + nrej >= A.max_steps) { status |= PK_ST_MAXSTEPS; break; }
      const bool last = (tc + 1.0001 * h >= te);
      const double hs = uni(last ? te - tc : ((tc + 2.0 * h > te) ? 0.5 * (te - tc) : h));
      if (!(hs > 1e-14 * fmax(fabs(tc), 1e-3))) { status |= PK_ST_HMIN; break; }
      const double g = uni(net_rcp(hs * GAM));
      freeze_factor(g);
      double Y[NRL], w[NRL], v[NRL], R[4][NRL], sb[NRL], se[NRL];
      auto Rg = [&](auto ic, int k) __attribute__((always_inline)) {          // right-hand side R_{ii + 3}, row k
        constexpr int ii = decltype(ic)::value;
        if constexpr (PARK && !CHAIN && ii >= 2) return pld((ii == 2 ? P_R5 : P_R6) + k); else return R[ii][k];
      };
      auto Rs = [&](auto ic, int k, double x) __attribute__((always_inline)) {
        constexpr int ii = decltype(ic)::value;
        if constexpr (PARK && !CHAIN && ii >= 2) pst((ii == 2 ? P_R5 : P_R6) + k, x); else R[ii][k] = x;
      };
      // ---- stage 1: Y_1 = y_n
      rhs_block(y, w, v, std::true_type{});
#pragma unroll
      for (int k = 0; k < NRL; ++k) {
        w[k] *= hs; v[k] *= hs;
        sb[k] = B[0] * w[k]; se[k] = EB[0] * w[k];
      }
      static_for<4>([&](auto ic) {
        constexpr int ii = decltype(ic)::value;                              // R_{ii + 3}
#pragma unroll
        for (int k = 0; k < NRL; ++k) Rs(ic, k, y[k] + AE[ii + 1][0] * w[k] + DI[ii + 1][0] * v[k]);
      });
      // ---- stages 2 .. 6
      static_for<5>([&](auto sc) {
        constexpr int s = 2 + decltype(sc)::value;
        double gr[NRL];                                                      // g * r_s
#pragma unroll
        for (int k = 0; k < NRL; ++k) {
          double rk;
          if constexpr (s == 2) rk = y[k] + AE[0][0] * w[k] + DI[0][0] * v[k]; else rk = Rg(std::integral_constant<int, (s >= 3 ? s - 3 : 0)>{}, k);
          gr[k] = g * rk;
        }
        block_solve(gr, Y);
#pragma unroll
        for (int k = 0; k < NRL; ++k) v[k] = __builtin_fma(-hs, gr[k], (1.0 / GAM) * Y[k]);      // h G_s = (Y_s - r_s) / gamma
        rhs_block(Y, w, gr, std::false_type{});
#pragma unroll
        for (int k = 0; k < NRL; ++k) {
          w[k] *= hs;
          if constexpr (s != 2) { sb[k] = __builtin_fma(B[s - 1], w[k], sb[k]); se[k] = __builtin_fma(EB[s - 1], w[k], se[k]); }
        }
        static_for<4>([&](auto ic) {
          constexpr int ii = decltype(ic)::value;
          if constexpr (ii + 3 > s) {
#pragma unroll
            for (int k = 0; k < NRL; ++k) Rs(ic, k, Rg(ic, k) + AE[ii + 1][s - 1] * w[k] + DI[ii + 1][s - 1] * v[k]);
          }
        });
      });
      // ---- new value and error estimate
      auto q = [&](double ev, double ya, double yb) { return fabs(ev) * net_rcp(A.atol + A.rtol * fmax(fabs(ya), fabs(yb))); };
      double e = 0.0;
#pragma unroll
      for (int k = 0; k < NRL; ++k) {
        sb[k] += y[k];                                                       // y_{n+1}
        if (valid[k]) e = err_acc(e, q(se[k], y[k], sb[k]), rms);
      }
      const double err = uni(err_reduce(e, rms, S, red));
      if (err != err || err > 1e300) {
        ++nrej; after_reject = true; h = uni(0.1 * hs);
        double bad = (nonfinite(Ai) || nonfinite(Bi) || nonfinite(Ci) || nonfinite(Di) || nonfinite(Ei) || nonfinite(ts)) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < NRL; ++k) if (nonfinite(y[k]) || nonfinite(LK(k)) || nonfinite(SR(k))) bad = 1.0;
        if (block_max(bad, red) != 0.0) { status |= PK_ST_NONFINITE; break; }
        continue;
      }
      // embedded order 3: err^(1/4) -- in single precision (two v_sqrt_f32 instead of two f64 square-root sequences, ~45 VALU instructions
      // of a step's ~1 100): the step-size factor needs three digits, and err > 1e300 / NaN was handled above (a float overflow gives
      // fac = inf -> the 5.0 clamp, an underflow 0 -> the growth clamp)
      double fac = (double)__builtin_sqrtf(__builtin_sqrtf((float)err)) * safety_inv;
      fac = fmax(grow_inv, fmin(5.0, fac));
      double hnew = uni(hs * net_rcp(fac));
      if (err <= 1.0) {
        ++nacc;
#pragma unroll
        for (int k = 0; k < NRL; ++k) y[k] = sb[k];
        tc = uni(tc + hs);
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
  if (status != PK_ST_OK && A.Y) {
    const double qnan = __builtin_nan("");
    for (int si = 0; si < A.n_stops; ++si) {
      const int row = stop_out[si];
      if (row >= 0 && !(stops[si] <= tc)) {
        double* o = Yout + (size_t)row * S;
#pragma unroll
        for (int k = 0; k < NRL; ++k) if (valid[k]) o[yoff(k)] = qnan;
      }
    }
  }
  if constexpr (PARK) {
    if (fuse) {
      // objective assembly of GlobalODE_MOO._evaluate (optproblem.py:99-160), as net_objective_kernel does it from a stored trajectory
      const double lp = block_sum(pld(P_LOSS), red), lr = block_sum(pld(P_LOSS + 1), red), lph = block_sum(pld(P_LOSS + 2), red);
      double prior = 0.0;
      if (A.loss_defaults) {
        double acc = 0.0;
        for (int k = tid; k < 5 * N; k += nt) {
          const int grp = k / N, ii = k - grp * N;
          const int off = (grp == 0 ? sl.A : grp == 1 ? sl.B : grp == 2 ? sl.C : grp == 3 ? sl.D : sl.E) + ii;
          const double pv = A.x_is_raw ? softplus(xb[off]) : xb[off];
          const double dd = (pv - A.loss_defaults[off]) / (A.loss_defaults[off] + 1e-6);
          acc = __builtin_fma(dd, dd, acc);
        }
        prior = A.loss_lam[3] * (block_sum(acc, red) / (double)(5 * N));
      }
      const bool anybad = block_max(ybad ? 1.0 : 0.0, red) != 0.0;
      if (tid == 0) {
        if (A.loss_sums) { A.loss_sums[3 * b] = lp; A.loss_sums[3 * b + 1] = lr; A.loss_sums[3 * b + 2] = lph; }
        if (A.loss_F) {
          const bool bad = status != PK_ST_OK || anybad;                   // a NaN loss of finite states stays NaN (LOSS_MODE 2 in the reference, too)
          A.loss_F[3 * b] = bad ? A.loss_fail : (lp * A.loss_norm[0]) * A.loss_lam[0] + prior;
          A.loss_F[3 * b + 1] = bad ? A.loss_fail : (lr * A.loss_norm[1]) * A.loss_lam[1] + prior;
          A.loss_F[3 * b + 2] = bad ? A.loss_fail : (lph * A.loss_norm[2]) * A.loss_lam[2] + prior;
        }
      }
    }
  }
  if (tid == 0) {
    if (A.status) A.status[b] = status;
    if (A.n_steps) { A.n_steps[2 * b] = nacc; A.n_steps[2 * b + 1] = nrej; }
  }
}

template <int MODEL, int NRL>
__global__ __launch_bounds__(512) void net_solve_arkp_kernel(const NetDev n, const NetSolveArgs A) { net_solve_arkp_body<MODEL, NRL, false>(n, A); }

// the register diet: 3 waves per SIMD (<= 168 VGPRs)
template <int MODEL, int NRL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(3, 3))) void net_solve_arkp3_kernel(const NetDev n, const NetSolveArgs A) {
  net_solve_arkp_body<MODEL, NRL, true>(n, A);
}

__host__ inline size_t net_solve_arkp_lds_bytes(const NetDev& n, int rows_per_lane, int threads, bool park) {
  return ((size_t)n.n_K + 2 * (size_t)n.N + 24 + (park ? (size_t)n.N + (size_t)arkp_park_stride(rows_per_lane) * threads : 0)) * 8;
}

}  // namespace pk
