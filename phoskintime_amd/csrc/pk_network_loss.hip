// pk_network_loss.hip -- fused three-objective loss of the network path, one workgroup per candidate.
//
// Reference: global_model/lossfn.py:114-382 (loss_function_noncomb / _comb with the eight LOSS_MODE point losses, :28-110) and the
// objective assembly of GlobalODE_MOO._evaluate (global_model/optproblem.py:99-160): prior penalty on A, B, C, D, E, finite check
// of the trajectory (fail_value otherwise), sums normalised by the total weight of each modality, times the user lambdas.
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>
#include "../../include/phoskin.h"
#include "pk_network.hpp"

struct pk_ctx;
struct pk_net;
extern "C" int pk_ctx_device(pk_ctx*);
extern "C" void* pk_ctx_stream(pk_ctx*);
extern "C" int pk_ctx_fail(pk_ctx*, int code, const char* msg);
extern "C" const pk::NetDev* pk_net_dev(const pk_net*);

namespace pk {

struct LossDev {
  int n_prot, n_rna, n_pho, base_prot, base_rna, base_pho;
  const int32_t *p_prot, *t_prot, *p_rna, *t_rna, *p_pho, *s_pho, *t_pho;
  const double *obs_prot, *w_prot, *obs_rna, *w_rna, *obs_pho, *w_pho;
  double norm_p, norm_r, norm_ph;              // 1 / max(1e-6, sum w)   (optproblem.py:83-85)
};

__device__ __forceinline__ double block_sum(double v, double* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double r = 0.0;
  for (int i = 0; i < nw; ++i) r += red[i];
  return r;
}

__global__ __launch_bounds__(256) void net_objective_kernel(const NetDev n, const LossDev L, const double* __restrict__ Y, const int T,
                                                            const int mode, const double* __restrict__ x, const int x_is_raw,
                                                            const double* __restrict__ defaults, const double lam_p, const double lam_r,
                                                            const double lam_ph, const double lam_prior, const double fail_value,
                                                            const int32_t* __restrict__ status, double* __restrict__ sums, double* __restrict__ F) {
  __shared__ double red[8];
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x, S = n.S;
  const double* Yb = Y + b * (size_t)T * S;
  auto at = [&](int t, int s) { return Yb[(size_t)t * S + s]; };
  const bool comb = n.model == 2;
  double lp = 0.0, lr = 0.0, lph = 0.0;
  for (int k = tid; k < L.n_prot; k += nt) {
    const int i = L.p_prot[k], st = n.offset_y[i], t = L.t_prot[k];
    const int cnt = comb ? (1 << n.n_sites[i]) : 1 + n.n_sites[i];
    double tt = 0.0, tb = 0.0;
    for (int m = 0; m < cnt; ++m) { tt += at(t, st + 1 + m); tb += at(L.base_prot, st + 1 + m); }
    const double pred = fold_change(tt, tb);
    lp += L.w_prot[k] * point_loss(mode, L.obs_prot[k] - pred, L.obs_prot[k], pred);
  }
  for (int k = tid; k < L.n_rna; k += nt) {
    const int st = n.offset_y[L.p_rna[k]];
    const double pred = fold_change(at(L.t_rna[k], st), at(L.base_rna, st));
    lr += L.w_rna[k] * point_loss(mode, L.obs_rna[k] - pred, L.obs_rna[k], pred);
  }
  for (int k = tid; k < L.n_pho; k += nt) {
    const int i = L.p_pho[k], st = n.offset_y[i], t = L.t_pho[k], j = L.s_pho[k];
    double a, c;
    if (comb) {
      a = 0.0; c = 0.0;
      const int cnt = 1 << n.n_sites[i];
      for (int m = 0; m < cnt; ++m) if (m & (1 << j)) { a += at(t, st + 1 + m); c += at(L.base_pho, st + 1 + m); }
    } else { a = at(t, st + 2 + j); c = at(L.base_pho, st + 2 + j); }
    const double pred = fold_change(a, c);
    lph += L.w_pho[k] * point_loss(mode, L.obs_pho[k] - pred, L.obs_pho[k], pred);
  }
  lp = block_sum(lp, red); lr = block_sum(lr, red); lph = block_sum(lph, red);
  // np.all(np.isfinite(Y)) over the whole trajectory (optproblem.py:130)
  double bad = 0.0;
  for (size_t k = tid; k < (size_t)T * S; k += nt) { const double v = Yb[k]; if (nonfinite(v)) bad = 1.0; }
  bad = block_sum(bad, red);
  if (status && status[b] != 0) bad = 1.0;
  double prior = 0.0;
  if (x && defaults) {
    const NetSlices sl(n.n_K, n.N, n.sites);
    const double* xb = x + b * n.n_var;
    double acc = 0.0;
    for (int k = tid; k < 5 * n.N; k += nt) {
      const int grp = k / n.N, i = k - grp * n.N;
      const int off = (grp == 0 ? sl.A : grp == 1 ? sl.B : grp == 2 ? sl.C : grp == 3 ? sl.D : sl.E) + i;
      const double p = x_is_raw ? softplus(xb[off]) : xb[off];
      const double d = (p - defaults[off]) / (defaults[off] + 1e-6);
      acc = __builtin_fma(d, d, acc);
    }
    acc = block_sum(acc, red);
    prior = lam_prior * (acc / (double)(5 * n.N));
  }
  if (tid == 0) {
    if (sums) { sums[3 * b] = lp; sums[3 * b + 1] = lr; sums[3 * b + 2] = lph; }
    if (F) {
      if (bad != 0.0) { F[3 * b] = fail_value; F[3 * b + 1] = fail_value; F[3 * b + 2] = fail_value; }
      else {
        F[3 * b] = (lp * L.norm_p) * lam_p + prior;
        F[3 * b + 1] = (lr * L.norm_r) * lam_r + prior;
        F[3 * b + 2] = (lph * L.norm_ph) * lam_ph + prior;
      }
    }
  }
}

// pred[b, :] = fold-change observables for the index lists of L (protein | rna | phospho, in that order), floor `eps`:
// the array form of simulate_and_measure's pred_fc columns (simulate.py:119-202, floor 1e-12) / of LOSS_FN's pred_fc (floor 1e-9).
__global__ __launch_bounds__(256) void net_observables_kernel(const NetDev n, const LossDev L, const double* __restrict__ Y, const int T,
                                                              const double eps, double* __restrict__ pred) {
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x, S = n.S;
  const double* Yb = Y + b * (size_t)T * S;
  auto at = [&](int t, int s) { return Yb[(size_t)t * S + s]; };
  auto fc = [&](double a, double c) { return (a > eps ? a : eps) / (c > eps ? c : eps); };
  const bool comb = n.model == 2;
  double* out = pred + b * (size_t)(L.n_prot + L.n_rna + L.n_pho);
  for (int k = tid; k < L.n_prot; k += nt) {
    const int i = L.p_prot[k], st = n.offset_y[i], t = L.t_prot[k];
    const int cnt = comb ? (1 << n.n_sites[i]) : 1 + n.n_sites[i];
    double tt = 0.0, tb = 0.0;
    for (int m = 0; m < cnt; ++m) { tt += at(t, st + 1 + m); tb += at(L.base_prot, st + 1 + m); }
    out[k] = fc(tt, tb);
  }
  for (int k = tid; k < L.n_rna; k += nt) {
    const int st = n.offset_y[L.p_rna[k]];
    out[L.n_prot + k] = fc(at(L.t_rna[k], st), at(L.base_rna, st));
  }
  for (int k = tid; k < L.n_pho; k += nt) {
    const int i = L.p_pho[k], st = n.offset_y[i], t = L.t_pho[k], j = L.s_pho[k];
    double a, c;
    if (comb) {
      a = 0.0; c = 0.0;
      const int cnt = 1 << n.n_sites[i];
      for (int m = 0; m < cnt; ++m) if (m & (1 << j)) { a += at(t, st + 1 + m); c += at(L.base_pho, st + 1 + m); }
    } else { a = at(t, st + 2 + j); c = at(L.base_pho, st + 2 + j); }
    out[L.n_prot + L.n_rna + k] = fc(a, c);
  }
}

// LOSS_FN without a network handle: the state offsets come from the caller's prot_map [N, 2] = (block start, n_sites | n_states), exactly
// the table global_model.lossfn.loss_function_noncomb / _comb index (lossfn.py:114-121, 250-257).  One workgroup per trajectory.
__global__ __launch_bounds__(256) void loss_fn_kernel(const LossDev L, const int32_t* __restrict__ prot_map, const int comb,
                                                      const double* __restrict__ Y, const int T, const int S, const int mode,
                                                      double* __restrict__ sums) {
  __shared__ double red[8];
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const double* Yb = Y + b * (size_t)T * S;
  auto at = [&](int t, int s) { return Yb[(size_t)t * S + s]; };
  double lp = 0.0, lr = 0.0, lph = 0.0;
  for (int k = tid; k < L.n_prot; k += nt) {
    const int i = L.p_prot[k], st = prot_map[2 * i], t = L.t_prot[k];
    const int cnt = comb ? prot_map[2 * i + 1] : 1 + prot_map[2 * i + 1];          // all 2^n states | P + its n site states
    double tt = 0.0, tb = 0.0;
    for (int m = 0; m < cnt; ++m) { tt += at(t, st + 1 + m); tb += at(L.base_prot, st + 1 + m); }
    const double pred = fold_change(tt, tb);
    lp += L.w_prot[k] * point_loss(mode, L.obs_prot[k] - pred, L.obs_prot[k], pred);
  }
  for (int k = tid; k < L.n_rna; k += nt) {
    const int st = prot_map[2 * L.p_rna[k]];
    const double pred = fold_change(at(L.t_rna[k], st), at(L.base_rna, st));
    lr += L.w_rna[k] * point_loss(mode, L.obs_rna[k] - pred, L.obs_rna[k], pred);
  }
  for (int k = tid; k < L.n_pho; k += nt) {
    const int i = L.p_pho[k], st = prot_map[2 * i], t = L.t_pho[k], j = L.s_pho[k];
    double a, c;
    if (comb) {
      a = 0.0; c = 0.0;
      const int cnt = prot_map[2 * i + 1];
      for (int m = 0; m < cnt; ++m) if (m & (1 << j)) { a += at(t, st + 1 + m); c += at(L.base_pho, st + 1 + m); }
    } else { a = at(t, st + 2 + j); c = at(L.base_pho, st + 2 + j); }
    const double pred = fold_change(a, c);
    lph += L.w_pho[k] * point_loss(mode, L.obs_pho[k] - pred, L.obs_pho[k], pred);
  }
  lp = block_sum(lp, red); lr = block_sum(lr, red); lph = block_sum(lph, red);
  if (tid == 0) { sums[3 * b] = lp; sums[3 * b + 1] = lr; sums[3 * b + 2] = lph; }
}

}  // namespace pk

struct pk_loss {
  pk::LossDev d;
  std::vector<void*> allocs;
  int T;
  // the same observations as dense [T, S] tables (value, weight; weight 0 = none), indexed like a trajectory row: what the fused
  // simulate + objective launch reads at each output time (pk_network_simulate_objective_batch).  Null when the lists cannot be fused:
  // a protein / phospho baseline other than time index 0, an rna observation before the rna baseline (the reference's production data:
  // baselines at t = 0, 4, 0 and rna observed from t = 4 on, runner.py:545-547), or two observations of one (state, time).
  const double* dense_obs = nullptr; const double* dense_w = nullptr;
};

namespace {
template <class T>
const T* up(pk_loss* l, const T* host, size_t count, bool& ok) {
  if (!ok) return nullptr;
  void* p = nullptr;
  if (hipMalloc(&p, (count ? count : 1) * sizeof(T)) != hipSuccess) { ok = false; return nullptr; }
  l->allocs.push_back(p);
  if (count && hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { ok = false; return nullptr; }
  return (const T*)p;
}
double norm_of(const double* w, int n) { double s = 0.0; for (int i = 0; i < n; ++i) s += w[i]; return 1.0 / (s > 1e-6 ? s : 1e-6); }
}  // namespace

extern "C" {

pk_loss* pk_network_loss_create(pk_ctx* c, pk_net* net, const pk_loss_data* d, int T) {
  if (!c || !net || !d) return nullptr;
  const pk::NetDev& n = *pk_net_dev(net);
  if (T < 1 || d->n_prot < 0 || d->n_rna < 0 || d->n_pho < 0) { pk_ctx_fail(c, PK_ERR_ARG, "bad loss-data sizes"); return nullptr; }
  auto bad_t = [&](int t) { return t < 0 || t >= T; };
  if (bad_t(d->prot_base_idx) || bad_t(d->rna_base_idx) || bad_t(d->pho_base_idx)) { pk_ctx_fail(c, PK_ERR_ARG, "baseline index outside the time grid"); return nullptr; }
  // index validation on the host (a bad index would be an out-of-bounds read on the GPU); n_sites is needed for s_pho
  std::vector<int32_t> ns(n.N);
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess || hipMemcpy(ns.data(), n.n_sites, n.N * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) {
    pk_ctx_fail(c, PK_ERR_HIP, "hipMemcpy"); return nullptr;
  }
  for (int k = 0; k < d->n_prot; ++k) if (d->p_prot[k] < 0 || d->p_prot[k] >= n.N || bad_t(d->t_prot[k])) { pk_ctx_fail(c, PK_ERR_ARG, "protein observation index out of range"); return nullptr; }
  for (int k = 0; k < d->n_rna; ++k) if (d->p_rna[k] < 0 || d->p_rna[k] >= n.N || bad_t(d->t_rna[k])) { pk_ctx_fail(c, PK_ERR_ARG, "rna observation index out of range"); return nullptr; }
  for (int k = 0; k < d->n_pho; ++k) {
    if (d->p_pho[k] < 0 || d->p_pho[k] >= n.N || bad_t(d->t_pho[k]) || d->s_pho[k] < 0 || d->s_pho[k] >= ns[d->p_pho[k]]) {
      pk_ctx_fail(c, PK_ERR_ARG, "phospho observation index out of range"); return nullptr;
    }
  }
  pk_loss* l = new pk_loss();
  l->T = T;
  bool ok = true;
  pk::LossDev& v = l->d;
  v.n_prot = d->n_prot; v.n_rna = d->n_rna; v.n_pho = d->n_pho;
  v.base_prot = d->prot_base_idx; v.base_rna = d->rna_base_idx; v.base_pho = d->pho_base_idx;
  v.p_prot = up(l, d->p_prot, d->n_prot, ok); v.t_prot = up(l, d->t_prot, d->n_prot, ok);
  v.obs_prot = up(l, d->obs_prot, d->n_prot, ok); v.w_prot = up(l, d->w_prot, d->n_prot, ok);
  v.p_rna = up(l, d->p_rna, d->n_rna, ok); v.t_rna = up(l, d->t_rna, d->n_rna, ok);
  v.obs_rna = up(l, d->obs_rna, d->n_rna, ok); v.w_rna = up(l, d->w_rna, d->n_rna, ok);
  v.p_pho = up(l, d->p_pho, d->n_pho, ok); v.s_pho = up(l, d->s_pho, d->n_pho, ok); v.t_pho = up(l, d->t_pho, d->n_pho, ok);
  v.obs_pho = up(l, d->obs_pho, d->n_pho, ok); v.w_pho = up(l, d->w_pho, d->n_pho, ok);
  v.norm_p = norm_of(d->w_prot, d->n_prot); v.norm_r = norm_of(d->w_rna, d->n_rna); v.norm_ph = norm_of(d->w_pho, d->n_pho);
  bool rna_after_base = true;
  for (int k = 0; k < d->n_rna; ++k) if (d->t_rna[k] < d->rna_base_idx) rna_after_base = false;
  if (ok && n.model != 2 && d->prot_base_idx == 0 && d->pho_base_idx == 0 && rna_after_base) {
    std::vector<int32_t> oy(n.N);
    if (hipMemcpy(oy.data(), n.offset_y, n.N * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) ok = false;
    std::vector<double> ob((size_t)T * n.S, 0.0), wt((size_t)T * n.S, 0.0);
    std::vector<char> seen((size_t)T * n.S, 0);
    bool dup = false;
    auto put = [&](int t, int s, double o, double w) {
      const size_t at = (size_t)t * n.S + s;
      if (seen[at]) dup = true;
      seen[at] = 1; ob[at] = o; wt[at] = w;
    };
    if (ok) {
      for (int k = 0; k < d->n_prot; ++k) put(d->t_prot[k], oy[d->p_prot[k]] + 1, d->obs_prot[k], d->w_prot[k]);
      for (int k = 0; k < d->n_rna; ++k) put(d->t_rna[k], oy[d->p_rna[k]], d->obs_rna[k], d->w_rna[k]);
      for (int k = 0; k < d->n_pho; ++k) put(d->t_pho[k], oy[d->p_pho[k]] + 2 + d->s_pho[k], d->obs_pho[k], d->w_pho[k]);
      if (!dup) { l->dense_obs = up(l, ob.data(), ob.size(), ok); l->dense_w = up(l, wt.data(), wt.size(), ok); }
    }
  }
  if (!ok) { pk_ctx_fail(c, PK_ERR_NOMEM, "hipMalloc / hipMemcpy failed"); pk_network_loss_destroy(l); return nullptr; }
  return l;
}

// for the fused launch (pk_network.hip): the dense tables, the normalisations and the grid length; 0 when the lists cannot be fused
int pk_loss_fused_tables(const pk_loss* l, const double** obs, const double** w, double* norms, int* T, int* rna_base) {
  if (!l || !l->dense_obs || !l->dense_w) return 0;
  *obs = l->dense_obs; *w = l->dense_w; norms[0] = l->d.norm_p; norms[1] = l->d.norm_r; norms[2] = l->d.norm_ph; *T = l->T;
  *rna_base = l->d.base_rna;
  return 1;
}

void pk_network_loss_destroy(pk_loss* l) {
  if (!l) return;
  for (void* p : l->allocs) (void)hipFree(p);
  delete l;
}

int pk_network_objective_batch(pk_ctx* c, pk_net* net, pk_loss* l, int64_t B, const double* Y, int T, int loss_mode, const double* x,
                               int x_is_raw, const double* defaults, const double* lambdas, double fail_value, const int32_t* status,
                               double* loss_sums, double* F) {
  if (!c || !net || !l) return PK_ERR_ARG;
  if (B < 0) return pk_ctx_fail(c, PK_ERR_ARG, "B must be >= 0");
  if (T != l->T) return pk_ctx_fail(c, PK_ERR_ARG, "T differs from the grid the loss data were created for");
  if (B == 0) return PK_OK;
  if (!Y || (!loss_sums && !F)) return pk_ctx_fail(c, PK_ERR_ARG, "null pointer");
  if (F && !lambdas) return pk_ctx_fail(c, PK_ERR_ARG, "lambdas (protein, rna, phospho, prior) are required for F");
  if (B > 0x7fffffffLL) return pk_ctx_fail(c, PK_ERR_ARG, "batch too large for one launch");
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return pk_ctx_fail(c, PK_ERR_HIP, "hipSetDevice");
  const double lp = lambdas ? lambdas[0] : 1.0, lr = lambdas ? lambdas[1] : 1.0, lph = lambdas ? lambdas[2] : 1.0, lpr = lambdas ? lambdas[3] : 0.0;
  hipLaunchKernelGGL(pk::net_objective_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)pk_ctx_stream(c), *pk_net_dev(net), l->d, Y, T,
                     loss_mode, x, x_is_raw, defaults, lp, lr, lph, lpr, fail_value, status, loss_sums, F);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PK_OK : pk_ctx_fail(c, PK_ERR_HIP, hipGetErrorString(e));
}

int pk_network_observables_batch(pk_ctx* c, pk_net* net, pk_loss* l, int64_t B, const double* Y, int T, double eps, double* pred) {
  if (!c || !net || !l) return PK_ERR_ARG;
  if (B < 0) return pk_ctx_fail(c, PK_ERR_ARG, "B must be >= 0");
  if (T != l->T) return pk_ctx_fail(c, PK_ERR_ARG, "T differs from the grid the index lists were created for");
  if (B == 0) return PK_OK;
  if (!Y || !pred) return pk_ctx_fail(c, PK_ERR_ARG, "null pointer");
  if (B > 0x7fffffffLL) return pk_ctx_fail(c, PK_ERR_ARG, "batch too large for one launch");
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return pk_ctx_fail(c, PK_ERR_HIP, "hipSetDevice");
  hipLaunchKernelGGL(pk::net_observables_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)pk_ctx_stream(c), *pk_net_dev(net), l->d, Y, T, eps, pred);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PK_OK : pk_ctx_fail(c, PK_ERR_HIP, hipGetErrorString(e));
}

// LOSS_FN(Y, p_prot, t_prot, obs_prot, w_prot, p_rna, ..., w_pho, prot_map, prot_base_idx, rna_base_idx, pho_base_idx) for B trajectories,
// host pointers throughout (the reference's call shape: numpy arrays): index-checked on the host, staged to HBM, one launch, synchronised.
int pk_loss_fn_batch_host(pk_ctx* c, int combinatorial, int loss_mode, int64_t B, const double* Y, int T, int S, const pk_loss_data* d,
                          const int32_t* prot_map, int N, double* loss_sums) {
  if (!c) return PK_ERR_ARG;
  if (B < 0 || T < 1 || S < 1 || N < 0) return pk_ctx_fail(c, PK_ERR_ARG, "B >= 0, T >= 1, S >= 1, N >= 0 required");
  if (B == 0) return PK_OK;
  if (!Y || !d || !loss_sums || (N && !prot_map)) return pk_ctx_fail(c, PK_ERR_ARG, "null pointer");
  if (B > 0x7fffffffLL) return pk_ctx_fail(c, PK_ERR_ARG, "batch too large for one launch");
  if (d->n_prot < 0 || d->n_rna < 0 || d->n_pho < 0) return pk_ctx_fail(c, PK_ERR_ARG, "bad loss-data sizes");
  auto bad_t = [&](int t) { return t < 0 || t >= T; };
  if (bad_t(d->prot_base_idx) || bad_t(d->rna_base_idx) || bad_t(d->pho_base_idx)) return pk_ctx_fail(c, PK_ERR_ARG, "baseline index outside the time grid");
  // every state index the kernel will form must lie inside a trajectory row
  auto blk_ok = [&](int i) {
    if (i < 0 || i >= N) return false;
    const int st = prot_map[2 * i], cnt = prot_map[2 * i + 1];
    if (st < 0 || cnt < 0 || (combinatorial && cnt > (1 << 20))) return false;
    return (long long)st + 1 + (combinatorial ? cnt : 1 + cnt) <= (long long)S;
  };
  for (int k = 0; k < d->n_prot; ++k) if (!blk_ok(d->p_prot[k]) || bad_t(d->t_prot[k])) return pk_ctx_fail(c, PK_ERR_ARG, "protein observation index out of range");
  for (int k = 0; k < d->n_rna; ++k) if (!blk_ok(d->p_rna[k]) || bad_t(d->t_rna[k])) return pk_ctx_fail(c, PK_ERR_ARG, "rna observation index out of range");
  for (int k = 0; k < d->n_pho; ++k) {
    if (!blk_ok(d->p_pho[k]) || bad_t(d->t_pho[k]) || d->s_pho[k] < 0) return pk_ctx_fail(c, PK_ERR_ARG, "phospho observation index out of range");
    const int cnt = prot_map[2 * d->p_pho[k] + 1];
    if (combinatorial ? (d->s_pho[k] >= 20 || (1 << d->s_pho[k]) >= (cnt > 1 ? cnt : 1)) : d->s_pho[k] >= cnt)
      return pk_ctx_fail(c, PK_ERR_ARG, "phospho site index out of range");
  }
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return pk_ctx_fail(c, PK_ERR_HIP, "hipSetDevice");
  pk_loss tmp;
  bool ok = true;
  pk::LossDev& v = tmp.d;
  v.n_prot = d->n_prot; v.n_rna = d->n_rna; v.n_pho = d->n_pho;
  v.base_prot = d->prot_base_idx; v.base_rna = d->rna_base_idx; v.base_pho = d->pho_base_idx;
  v.p_prot = up(&tmp, d->p_prot, d->n_prot, ok); v.t_prot = up(&tmp, d->t_prot, d->n_prot, ok);
  v.obs_prot = up(&tmp, d->obs_prot, d->n_prot, ok); v.w_prot = up(&tmp, d->w_prot, d->n_prot, ok);
  v.p_rna = up(&tmp, d->p_rna, d->n_rna, ok); v.t_rna = up(&tmp, d->t_rna, d->n_rna, ok);
  v.obs_rna = up(&tmp, d->obs_rna, d->n_rna, ok); v.w_rna = up(&tmp, d->w_rna, d->n_rna, ok);
  v.p_pho = up(&tmp, d->p_pho, d->n_pho, ok); v.s_pho = up(&tmp, d->s_pho, d->n_pho, ok); v.t_pho = up(&tmp, d->t_pho, d->n_pho, ok);
  v.obs_pho = up(&tmp, d->obs_pho, d->n_pho, ok); v.w_pho = up(&tmp, d->w_pho, d->n_pho, ok);
  v.norm_p = v.norm_r = v.norm_ph = 1.0;
  const int32_t* pm = up(&tmp, prot_map, (size_t)2 * N, ok);
  const double* Yd = up(&tmp, Y, (size_t)B * T * S, ok);
  double* out = nullptr;
  if (ok && hipMalloc((void**)&out, (size_t)B * 3 * sizeof(double)) == hipSuccess) tmp.allocs.push_back(out); else ok = false;
  int rc = PK_OK;
  if (!ok) rc = pk_ctx_fail(c, PK_ERR_NOMEM, "hipMalloc / hipMemcpy failed");
  else {
    hipStream_t st = (hipStream_t)pk_ctx_stream(c);
    hipLaunchKernelGGL(pk::loss_fn_kernel, dim3((unsigned)B), dim3(256), 0, st, v, pm, combinatorial ? 1 : 0, Yd, T, S, loss_mode, out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(loss_sums, out, (size_t)B * 3 * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = pk_ctx_fail(c, PK_ERR_HIP, hipGetErrorString(e));
  }
  for (void* p : tmp.allocs) (void)hipFree(p);
  return rc;
}

}  // extern "C"
