// pk_network.hip -- kernels and C ABI of the network (global_model) path: batched RHS, analytic Jacobian, softplus unpack.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>
#include "../../include/phoskin.h"
#include "pk_network.hpp"
#include "pk_network_solve.hpp"
#include "pk_network_solve_reg.hpp"
#include "pk_network_solve_reg2.hpp"
#include "pk_network_rk45.hpp"
namespace pk {
// order-4 additive integrator (pk_network_solve_ark.hpp), its own translation unit: returns the dynamic LDS it needs, or launches
size_t net_ark_lds_bytes(const NetDev& n, int nnzT, int max_sites, int threads);
hipError_t launch_net_ark(const NetDev& n, const NetSolveArgs& a, int max_sites, long long B, int threads, size_t lds, hipStream_t st);
// the same method in the dense two-lanes-per-protein layout (pk_network_solve_arkp.hpp; arrow topologies): n.lane_unit must be set
bool net_arkp_enabled();
bool net_arkp_fits(const NetDev& n, int max_sites);
bool net_arkp_fuses_loss();                      // the register-diet kernel (PK_ARK_PAIR=3, the default) scores observations at its output times
hipError_t launch_net_arkp(const NetDev& n, const NetSolveArgs& a, int nnzT, int max_sites, long long B, hipStream_t st);
}
#include <algorithm>
#include <cstdlib>

struct pk_ctx;                                   // defined in pk_capi.hip
extern "C" int pk_ctx_device(pk_ctx*);
extern "C" void* pk_ctx_stream(pk_ctx*);
extern "C" int pk_ctx_fail(pk_ctx*, int code, const char* msg);

namespace pk {

// dydt[b, :] = f(t_b, y_b; x_b)
__global__ __launch_bounds__(256) void net_rhs_kernel(const NetDev n, const double* __restrict__ x, const int x_is_raw,
                                                      const double* __restrict__ y, const int y_batched,
                                                      const double* __restrict__ t, const int t_batched, double* __restrict__ dydt) {
  extern __shared__ __align__(16) double lds[];
  const NetLds L(lds, n);
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const double* xb = x + b * n.n_var;
  for (int k = tid; k < n.n_var; k += nt) L.p[k] = x_is_raw ? softplus(xb[k]) : xb[k];
  const double* yb = y + (y_batched ? b * n.S : 0);
  for (int k = tid; k < n.S; k += nt) L.y[k] = yb[k];
  const int jb = net_bucket(t[t_batched ? b : 0], n.kin_grid, n.n_grid);
  __syncthreads();
  net_prepare<false>(n, L, jb);
  for (int k = tid; k < n.S; k += nt) dydt[b * n.S + k] = net_state_rhs(n, L, k);
}

// J[b, r, c] = d f_r / d y_c, row-major, analytic
__global__ __launch_bounds__(256) void net_jac_kernel(const NetDev n, const double* __restrict__ x, const int x_is_raw,
                                                      const double* __restrict__ y, const int y_batched,
                                                      const double* __restrict__ t, const int t_batched, double* __restrict__ J) {
  extern __shared__ __align__(16) double lds[];
  const NetLds L(lds, n);
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const double* xb = x + b * n.n_var;
  for (int k = tid; k < n.n_var; k += nt) L.p[k] = x_is_raw ? softplus(xb[k]) : xb[k];
  const double* yb = y + (y_batched ? b * n.S : 0);
  for (int k = tid; k < n.S; k += nt) L.y[k] = yb[k];
  const int jb = net_bucket(t[t_batched ? b : 0], n.kin_grid, n.n_grid);
  __syncthreads();
  net_prepare<true>(n, L, jb);
  double* Jb = J + b * (size_t)n.S * n.S;
  // block-diagonal part: one thread per row, its protein's columns; everything else zero
  for (int r = tid; r < n.S; r += nt) {
    double* row = Jb + (size_t)r * n.S;
    for (int c = 0; c < n.S; ++c) row[c] = 0.0;
    const int i = n.state_prot[r], lr = n.state_local[r];
    const int st = n.offset_y[i];
    const int cnt = (n.model == 2) ? 1 + (1 << n.n_sites[i]) : 2 + n.n_sites[i];
    for (int lc = 0; lc < cnt; ++lc) row[st + lc] = net_block_jac(n, L, i, lr, lc);
    if (lr == 0) {
      // TF coupling of the mRNA row: d synth_i / d y_c = dsyn_i * TF_ij * [c is a protein-form state of an undriven regulator j]
      for (int q = n.TF_indptr[i]; q < n.TF_indptr[i + 1]; ++q) {
        const int j = n.TF_indices[q];
        if (n.model != 2 && n.driver_map[j] >= 0) continue;
        const double w = L.dsyn[i] * n.TF_data[q];
        const int sj = n.offset_y[j];
        const int cj = (n.model == 2) ? (1 << n.n_sites[j]) : 1 + n.n_sites[j];
        for (int m = 0; m < cj; ++m) row[sj + 1 + m] += w;
      }
    }
  }
}

__global__ void net_unpack_kernel(const double* __restrict__ x, double* __restrict__ out, const long long total) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < total) out[k] = softplus(x[k]);
}

}  // namespace pk

struct pk_net {
  pk::NetDev d;
  std::vector<void*> allocs;
  size_t lds_bytes;
  size_t solve_lds_bytes;
  size_t solve_reg_lds_bytes;
  int max_sites;
  int nnzT = 0;
  std::vector<double> kin_grid_host;
  double* stops_dev = nullptr; int32_t* stop_out_dev = nullptr; size_t stops_cap = 0;
  std::mutex mu;       // a network handle may be shared by the threads of a process (each with its own pk_ctx): guards the stop buffers
};

namespace {
template <class T>
const T* upload(pk_net* n, const T* host, size_t count, bool& ok) {
  if (!ok) return nullptr;
  void* p = nullptr;
  if (hipMalloc(&p, (count ? count : 1) * sizeof(T)) != hipSuccess) { ok = false; return nullptr; }
  n->allocs.push_back(p);
  if (count && hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { ok = false; return nullptr; }
  return (const T*)p;
}
}  // namespace

extern "C" {

pk_net* pk_network_create(pk_ctx* c, const pk_network_desc* d) {
  if (!c || !d) return nullptr;
  if (!(d->model == 0 || d->model == 1 || d->model == 2 || d->model == 4)) { pk_ctx_fail(c, PK_ERR_ARG, "network model must be 0, 1, 2 or 4"); return nullptr; }
  if (d->N < 1 || d->n_K < 1 || d->total_sites < 0 || d->n_grid < 1) { pk_ctx_fail(c, PK_ERR_ARG, "bad network dimensions"); return nullptr; }
  if (!d->offset_y || !d->offset_s || !d->n_sites || !d->W_indptr || !d->TF_indptr || !d->tf_deg || !d->driver_map || !d->kin_grid || !d->kin_Kmat) {
    pk_ctx_fail(c, PK_ERR_ARG, "null array in pk_network_desc"); return nullptr;
  }
  // validate the layout on the host: a wrong offset would become an out-of-bounds access on the GPU
  int S = 0, sites = 0;
  std::vector<int32_t> sp, sl;
  for (int i = 0; i < d->N; ++i) {
    const int ns = d->n_sites[i];
    if (ns < 0 || (d->model == 2 && ns > 10)) { pk_ctx_fail(c, PK_ERR_ARG, "n_sites out of range (model 2: <= 10)"); return nullptr; }
    if (d->offset_y[i] != S || d->offset_s[i] != sites) { pk_ctx_fail(c, PK_ERR_ARG, "offset_y / offset_s are not the running sums of the block sizes"); return nullptr; }
    const int cnt = (d->model == 2) ? 1 + (1 << ns) : 2 + ns;
    for (int l = 0; l < cnt; ++l) { sp.push_back(i); sl.push_back(l); }
    S += cnt; sites += ns;
    if (d->driver_map[i] >= d->n_K) { pk_ctx_fail(c, PK_ERR_ARG, "driver_map entry >= n_K"); return nullptr; }
  }
  if (sites != d->total_sites) { pk_ctx_fail(c, PK_ERR_ARG, "sum(n_sites) != total_sites"); return nullptr; }
  if (d->W_indptr[0] != 0 || d->TF_indptr[0] != 0) { pk_ctx_fail(c, PK_ERR_ARG, "CSR indptr must start at 0"); return nullptr; }
  for (int r = 0; r < sites; ++r) if (d->W_indptr[r + 1] < d->W_indptr[r]) { pk_ctx_fail(c, PK_ERR_ARG, "W_indptr not monotone"); return nullptr; }
  for (int r = 0; r < d->N; ++r) if (d->TF_indptr[r + 1] < d->TF_indptr[r]) { pk_ctx_fail(c, PK_ERR_ARG, "TF_indptr not monotone"); return nullptr; }
  const int nnzW = d->W_indptr[sites], nnzT = d->TF_indptr[d->N];
  for (int q = 0; q < nnzW; ++q) if (d->W_indices[q] < 0 || d->W_indices[q] >= d->n_K) { pk_ctx_fail(c, PK_ERR_ARG, "W_indices out of range"); return nullptr; }
  for (int q = 0; q < nnzT; ++q) if (d->TF_indices[q] < 0 || d->TF_indices[q] >= d->N) { pk_ctx_fail(c, PK_ERR_ARG, "TF_indices out of range"); return nullptr; }
  for (int g = 1; g < d->n_grid; ++g) if (!(d->kin_grid[g] > d->kin_grid[g - 1])) { pk_ctx_fail(c, PK_ERR_ARG, "kin_grid must increase"); return nullptr; }

  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return nullptr;
  pk_net* n = new pk_net();
  bool ok = true;
  pk::NetDev& v = n->d;
  v.model = d->model; v.N = d->N; v.n_K = d->n_K; v.sites = sites; v.S = S; v.n_grid = d->n_grid; v.n_var = d->n_K + 5 * d->N + sites + 1;
  v.offset_y = upload(n, d->offset_y, d->N, ok); v.offset_s = upload(n, d->offset_s, d->N, ok); v.n_sites = upload(n, d->n_sites, d->N, ok);
  v.state_prot = upload(n, sp.data(), sp.size(), ok); v.state_local = upload(n, sl.data(), sl.size(), ok);
  v.W_indptr = upload(n, d->W_indptr, sites + 1, ok); v.W_indices = upload(n, d->W_indices, nnzW, ok); v.W_data = upload(n, d->W_data, nnzW, ok);
  v.TF_indptr = upload(n, d->TF_indptr, d->N + 1, ok); v.TF_indices = upload(n, d->TF_indices, nnzT, ok); v.TF_data = upload(n, d->TF_data, nnzT, ok);
  v.tf_deg = upload(n, d->tf_deg, d->N, ok); v.driver_map = upload(n, d->driver_map, d->N, ok);
  v.kin_grid = upload(n, d->kin_grid, d->n_grid, ok); v.kin_Kmat = upload(n, d->kin_Kmat, (size_t)d->n_K * d->n_grid, ok);
  n->lds_bytes = ((size_t)v.n_var + v.S + v.n_K + v.sites + 3 * (size_t)v.N) * sizeof(double);
  n->solve_lds_bytes = pk::net_solve_lds_bytes(v, nnzT);
  n->solve_reg_lds_bytes = pk::net_solve_reg_lds_bytes(v, nnzT);
  n->nnzT = nnzT;
  n->max_sites = 0;
  for (int i = 0; i < d->N; ++i) n->max_sites = std::max(n->max_sites, (int)d->n_sites[i]);
  n->kin_grid_host.assign(d->kin_grid, d->kin_grid + d->n_grid);
  v.lane_unit = nullptr; v.n_lanes = 0;
  if ((d->model == 0 || d->model == 1 || d->model == 4) && n->max_sites <= 8) {
    // dense lane table: a lane holds (2 + site class) / 2 rows; proteins with more sites than fit beside mRNA and protein take a second lane
    const int cls = n->max_sites <= 4 ? 4 : n->max_sites <= 6 ? 6 : 8, nrl = (2 + cls) / 2;
    std::vector<int32_t> lanes;
    // ... and so do proteins with more than 4 regulators: the lanes of a pair split the TF row, 4 register-resident entries each
    auto two = [&](int i) { return d->n_sites[i] > nrl - 2 || d->TF_indptr[i + 1] - d->TF_indptr[i] > 4; };
    for (int i = 0; i < d->N; ++i) if (two(i)) { lanes.push_back((i << 2) | 2); lanes.push_back((i << 2) | 3); }
    for (int i = 0; i < d->N; ++i) if (!two(i)) lanes.push_back(i << 2);
    v.n_lanes = (int)lanes.size();
    v.lane_unit = upload(n, lanes.data(), lanes.size(), ok);
  }
  if (!ok || n->lds_bytes > 160 * 1024) {
    pk_ctx_fail(c, ok ? PK_ERR_UNSUPPORTED : PK_ERR_NOMEM, ok ? "network too large for one workgroup's LDS (160 KiB)" : "hipMalloc / hipMemcpy failed");
    pk_network_destroy(n);
    return nullptr;
  }
  if (n->lds_bytes > 48 * 1024) {
    (void)hipFuncSetAttribute((const void*)pk::net_rhs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)n->lds_bytes);
    (void)hipFuncSetAttribute((const void*)pk::net_jac_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)n->lds_bytes);
  }
  if (n->solve_lds_bytes > 48 * 1024 && n->solve_lds_bytes <= 160 * 1024) {
    (void)hipFuncSetAttribute((const void*)pk::net_solve_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)n->solve_lds_bytes);
    (void)hipFuncSetAttribute((const void*)pk::net_solve_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)n->solve_lds_bytes);
    (void)hipFuncSetAttribute((const void*)pk::net_solve_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)n->solve_lds_bytes);
    (void)hipFuncSetAttribute((const void*)pk::net_solve_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)n->solve_lds_bytes);
  }
  return n;
}

void pk_network_destroy(pk_net* n) {
  if (!n) return;
  for (void* p : n->allocs) (void)hipFree(p);
  if (n->stops_dev) (void)hipFree(n->stops_dev);
  if (n->stop_out_dev) (void)hipFree(n->stop_out_dev);
  delete n;
}

const pk::NetDev* pk_net_dev(const pk_net* n) { return &n->d; }
int pk_network_n_states(const pk_net* n) { return n ? n->d.S : PK_ERR_ARG; }
int pk_network_n_var(const pk_net* n) { return n ? n->d.n_var : PK_ERR_ARG; }

static int net_args_ok(pk_ctx* c, pk_net* n, int64_t B, const void* x, const void* y, const void* t, const void* out) {
  if (!c || !n) return PK_ERR_ARG;
  if (B < 0) return pk_ctx_fail(c, PK_ERR_ARG, "B must be >= 0");
  if (B > 0 && (!x || !y || !t || !out)) return pk_ctx_fail(c, PK_ERR_ARG, "null pointer");
  if (B > 0x7fffffffLL) return pk_ctx_fail(c, PK_ERR_ARG, "batch too large for one launch");
  return PK_OK;
}

int pk_network_rhs_batch(pk_ctx* c, pk_net* n, int64_t B, const double* x, int x_is_raw, const double* y, int y_is_batched,
                         const double* t, int t_is_batched, double* dydt) {
  int rc = net_args_ok(c, n, B, x, y, t, dydt);
  if (rc || B == 0) return rc;
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return pk_ctx_fail(c, PK_ERR_HIP, "hipSetDevice");
  hipLaunchKernelGGL(pk::net_rhs_kernel, dim3((unsigned)B), dim3(256), n->lds_bytes, (hipStream_t)pk_ctx_stream(c), n->d, x, x_is_raw, y,
                     y_is_batched, t, t_is_batched, dydt);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PK_OK : pk_ctx_fail(c, PK_ERR_HIP, hipGetErrorString(e));
}

int pk_network_jacobian_batch(pk_ctx* c, pk_net* n, int64_t B, const double* x, int x_is_raw, const double* y, int y_is_batched,
                              const double* t, int t_is_batched, double* J) {
  int rc = net_args_ok(c, n, B, x, y, t, J);
  if (rc || B == 0) return rc;
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return pk_ctx_fail(c, PK_ERR_HIP, "hipSetDevice");
  hipLaunchKernelGGL(pk::net_jac_kernel, dim3((unsigned)B), dim3(256), n->lds_bytes, (hipStream_t)pk_ctx_stream(c), n->d, x, x_is_raw, y,
                     y_is_batched, t, t_is_batched, J);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PK_OK : pk_ctx_fail(c, PK_ERR_HIP, hipGetErrorString(e));
}

// Which integrator pk_network_simulate_batch runs for these options on this network -- the single source of truth (the Python host layer
// picks its default tolerances from it).  PK_METHOD_DP5 on request; PK_METHOD_ARK436 where its kernel fits and is the default (or was
// asked for); otherwise PK_METHOD_ROS34PW2.  PK_ERR_UNSUPPORTED when ARK436 was requested and cannot run.
int pk_network_resolve_method(const pk_net* n, const pk_solver_opts* opts) {
  if (!n) return PK_ERR_ARG;
  const int method = opts ? opts->method : PK_METHOD_LRP12, linsolve = opts ? opts->linsolve : PK_LINSOLVE_AUTO;
  if (method == PK_METHOD_DP5) return PK_METHOD_DP5;
  const int threads_a = ((n->d.N + 63) / 64) * 64;
  const bool ark_fits = n->d.N <= 256 && n->max_sites <= (n->d.model == 2 ? 3 : 8) && linsolve != PK_LINSOLVE_STRUCTURED;
  // one thread per protein (N <= 256), or [r3] the dense lane layout of topologies 0 / 1 / 4 (<= 512 lanes: up to 512 proteins)
  const bool ark_ok = (ark_fits && pk::net_ark_lds_bytes(n->d, n->nnzT, n->max_sites, threads_a) <= 160 * 1024) ||
                      (linsolve != PK_LINSOLVE_STRUCTURED && pk::net_arkp_fits(n->d, n->max_sites));
  if (method == PK_METHOD_ARK436) return ark_ok ? PK_METHOD_ARK436 : PK_ERR_UNSUPPORTED;
  // [r3] the combinatorial topology takes the additive method by default too: with the EXACT block solve (parity elimination of the
  // bit-pattern block, pk_network_solve_ark.hpp) it needs 4.6x fewer steps than the order-3 method and runs 1.8x faster at equal band error
  return (ark_ok && method != PK_METHOD_ROS34PW2) ? PK_METHOD_ARK436 : PK_METHOD_ROS34PW2;
}

// `fused`: null, or the loss fields of NetSolveArgs filled in (pk_network_simulate_objective_batch); Y may then be null
static int net_simulate_impl(pk_ctx* c, pk_net* n, int64_t B, const double* x, int x_is_raw, const double* y0, int y0_is_batched,
                             const double* t_host, int T, const pk_solver_opts* opts_in, double* Y, int32_t* status, int32_t* n_steps,
                             const pk::NetSolveArgs* fused) {
  if (!c || !n) return PK_ERR_ARG;
  if (B < 0) return pk_ctx_fail(c, PK_ERR_ARG, "B must be >= 0");
  if (T < 1 || !t_host) return pk_ctx_fail(c, PK_ERR_ARG, "t must hold >= 1 time points (host pointer)");
  if (B == 0) return PK_OK;
  if (!x || !y0 || (!Y && !fused)) return pk_ctx_fail(c, PK_ERR_ARG, "null pointer");
  if (B > 0x7fffffffLL) return pk_ctx_fail(c, PK_ERR_ARG, "batch too large for one launch");
  const bool dp5 = opts_in && opts_in->method == PK_METHOD_DP5;
  if (n->d.model == 2 && n->max_sites > 16) return pk_ctx_fail(c, PK_ERR_UNSUPPORTED, "combinatorial topology: <= 16 sites per protein");
  // combinatorial blocks: <= 3 sites per protein (and N <= 256) run one thread per protein out of registers (pk_network_solve_reg2.hpp);
  // larger blocks run in the general LDS kernel with the same approximate factorisation, swept serially by the protein's thread
  const bool comb_reg = n->d.model == 2 && n->max_sites <= 3 && n->d.N <= 256;
  if (!dp5 && !(n->d.model == 2 && comb_reg) && n->solve_lds_bytes > 160 * 1024) return pk_ctx_fail(c, PK_ERR_UNSUPPORTED, "network too large for one workgroup's LDS (160 KiB)");
  if (n->d.S > 1024 || n->d.N > 512) return pk_ctx_fail(c, PK_ERR_UNSUPPORTED, "simulate: S <= 1024 states and N <= 512 proteins per network");
  for (int k = 1; k < T; ++k) if (!(t_host[k] > t_host[k - 1])) return pk_ctx_fail(c, PK_ERR_ARG, "t must be strictly increasing");
  pk_solver_opts o;
  if (opts_in) o = *opts_in; else pk_default_opts(&o);
  if (!(o.rtol > 0.0 && o.atol >= 0.0)) return pk_ctx_fail(c, PK_ERR_ARG, "rtol must be > 0 and atol >= 0");
  if (o.max_steps <= 0) o.max_steps = dp5 ? 2000000 : 1000000;      // solvers.py:294 max_steps = 2_000_000
  if (dp5 && pk::net_rk45_lds_bytes(n->d) > 160 * 1024) return pk_ctx_fail(c, PK_ERR_UNSUPPORTED, "network too large for one workgroup's LDS (160 KiB)");
  // landing points: every output time after t[0] plus every bucket edge strictly inside (t[0], t[T-1])
  std::vector<std::pair<double, int>> st;
  for (int k = 1; k < T; ++k) st.push_back({t_host[k], k});
  for (double gk : n->kin_grid_host) {
    if (gk > t_host[0] && gk < t_host[T - 1]) {
      bool dup = false;
      for (int k = 1; k < T; ++k) if (t_host[k] == gk) { dup = true; break; }
      if (!dup) st.push_back({gk, -1});
    }
  }
  std::sort(st.begin(), st.end());
  if (dp5) {                                        // the explicit integrator interpolates its outputs: it only needs the T output times
    st.clear();
    for (int k = 0; k < T; ++k) st.push_back({t_host[k], k});
  }
  pk::NetSolveArgs a;
  std::memset(&a, 0, sizeof(a));
  a.x = x; a.x_is_raw = x_is_raw; a.y0 = y0; a.y0_batched = y0_is_batched ? 1 : 0; a.t0 = t_host[0]; a.T = T; a.Y = Y;
  a.status = status; a.n_steps = n_steps; a.rtol = o.rtol; a.atol = o.atol; a.h0 = o.h0; a.max_steps = o.max_steps;
  if (o.err_norm < PK_NORM_DEFAULT || o.err_norm > PK_NORM_RMS) return pk_ctx_fail(c, PK_ERR_ARG, "unknown opts->err_norm");
  a.err_rms = (o.err_norm == PK_NORM_RMS) ? 1 : 0;
  a.n_stops = (int)st.size();
  if (fused) {
    a.loss_obs = fused->loss_obs; a.loss_w = fused->loss_w; a.loss_defaults = fused->loss_defaults; a.loss_mode = fused->loss_mode;
    a.loss_fail = fused->loss_fail; a.loss_sums = fused->loss_sums; a.loss_F = fused->loss_F; a.loss_rna_base = fused->loss_rna_base;
    for (int k = 0; k < 4; ++k) a.loss_lam[k] = fused->loss_lam[k];
    for (int k = 0; k < 3; ++k) a.loss_norm[k] = fused->loss_norm[k];
    const bool can = !dp5 && pk_network_resolve_method(n, &o) == PK_METHOD_ARK436 && pk::net_arkp_fits(n->d, n->max_sites) && pk::net_arkp_fuses_loss();
    if (!can) return pk_ctx_fail(c, PK_ERR_UNSUPPORTED, "fused objective: topologies 0 / 1 / 4 on the default additive integrator only; "
                                                         "use pk_network_simulate_batch + pk_network_objective_batch");
  }
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return pk_ctx_fail(c, PK_ERR_HIP, "hipSetDevice");
  hipStream_t stream = (hipStream_t)pk_ctx_stream(c);
  if (st.size() <= 64) {
    for (size_t i = 0; i < st.size(); ++i) { a.stops_v[i] = st[i].first; a.stop_out_v[i] = st[i].second; }
  } else {
    // long output grids: stage through a per-network device buffer (the previous launch on it must have finished)
    std::lock_guard<std::mutex> g(n->mu);
    if (hipDeviceSynchronize() != hipSuccess) return pk_ctx_fail(c, PK_ERR_HIP, "hipDeviceSynchronize");      // any stream of any context
    if (n->stops_cap < st.size()) {
      if (n->stops_dev) (void)hipFree(n->stops_dev);
      if (n->stop_out_dev) (void)hipFree(n->stop_out_dev);
      n->stops_dev = nullptr; n->stop_out_dev = nullptr; n->stops_cap = 0;
      if (hipMalloc((void**)&n->stops_dev, st.size() * 8) != hipSuccess || hipMalloc((void**)&n->stop_out_dev, st.size() * 4) != hipSuccess)
        return pk_ctx_fail(c, PK_ERR_NOMEM, "hipMalloc");
      n->stops_cap = st.size();
    }
    std::vector<double> sv(st.size()); std::vector<int32_t> so(st.size());
    for (size_t i = 0; i < st.size(); ++i) { sv[i] = st[i].first; so[i] = st[i].second; }
    if (hipMemcpy(n->stops_dev, sv.data(), sv.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(n->stop_out_dev, so.data(), so.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
      return pk_ctx_fail(c, PK_ERR_HIP, "hipMemcpy");
    a.stops_p = n->stops_dev; a.stop_out_p = n->stop_out_dev;
  }
  if (dp5) {
    const size_t lb = pk::net_rk45_lds_bytes(n->d);
#define PK_RK_LAUNCH(M)                                                                                                              \
    do {                                                                                                                               \
      if (lb > 48 * 1024) (void)hipFuncSetAttribute((const void*)pk::net_rk45_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb); \
      hipLaunchKernelGGL(pk::net_rk45_kernel<M>, dim3((unsigned)B), dim3(256), lb, stream, n->d, a);                                   \
    } while (0)
    if (n->d.model == 0) PK_RK_LAUNCH(0); else if (n->d.model == 1) PK_RK_LAUNCH(1); else if (n->d.model == 2) PK_RK_LAUNCH(2); else PK_RK_LAUNCH(4);
#undef PK_RK_LAUNCH
    hipError_t er = hipGetLastError();
    return er == hipSuccess ? PK_OK : pk_ctx_fail(c, PK_ERR_HIP, hipGetErrorString(er));
  }
  // Register-resident kernel (one thread per protein) when every block fits its per-thread arrays; opts->linsolve ==
  // PK_LINSOLVE_STRUCTURED forces the LDS kernel (kept as the general fallback and as the A/B reference).
  // ---- default integrator: ARK436 (order 4) in the one-thread-per-protein layout; ROS34PW2 (order 3) everywhere else / on request
  {
    const int threads_a = ((n->d.N + 63) / 64) * 64;
    const int resolved = pk_network_resolve_method(n, &o);
    if (resolved == PK_ERR_UNSUPPORTED)
      return pk_ctx_fail(c, PK_ERR_UNSUPPORTED, "PK_METHOD_ARK436: <= 8 sites per protein and N <= 256 (topologies 0 / 1 / 4: <= 512 lanes of the dense layout); "
                                                "combinatorial topology: <= 3 sites, N <= 256; use PK_METHOD_ROS34PW2");
    const size_t lds_a = resolved == PK_METHOD_ARK436 ? pk::net_ark_lds_bytes(n->d, n->nnzT, n->max_sites, threads_a) : 0;
    if (resolved == PK_METHOD_ARK436) {
      // The order-4 method runs at 0.25 x the requested tolerances: at that factor its error equals the order-3 method's at the SAME nominal
      // tolerance.  Measured at rtol = atol = 1e-8, band widths from the converged solution (tools/gpu_ark_population*.py, bench.py):
      //   BASELINE config 5's population (8 192 candidates, log-normal 0.5 around the defaults), fixture candidate vs LSODA@1e-12:
      //     factor 1: 914 steps, 0.085;  0.5: 1 102, 0.048;  0.25: 1 329, 0.023;  ROS34PW2: 5 516 steps, 0.022  (reference's LSODA@1e-8: 0.054)
      //   2 048 candidates UNIFORM IN THE OPTIMISER'S RAW BOUNDS (extreme rates, strong TF coupling):
      //     factor 0.5: 4 217 steps, median 0.21, p99 0.85, 0.7 % beyond the band;  0.25: 5 114 steps, median 0.11, p99 0.41, 0.5 % beyond;
      //     ROS34PW2: 7 478 steps, median 0.16, p99 0.31, 0.05 % beyond  (both methods: worst candidates 6-7 band widths off)
      // (dev knob: PK_ARK_TOLFAC, read once)
      static const double tolfac = [] { const char* v = getenv("PK_ARK_TOLFAC"); const double f = v ? atof(v) : 0.0; return (f > 0.0 && f <= 1.0) ? f : 0.25; }();
      pk::NetSolveArgs aa = a;
      aa.rtol *= tolfac; aa.atol *= tolfac;
      static const double ctl_s = [] { const char* v = getenv("PK_ARK_SAFETY"); return v ? atof(v) : 0.0; }();      // dev knobs of the controller
      static const double ctl_g = [] { const char* v = getenv("PK_ARK_GROW"); return v ? atof(v) : 0.0; }();
      aa.ctl_safety = ctl_s; aa.ctl_grow = ctl_g;
      // [r3] arrow topologies: the dense two-lanes-per-protein layout, every stage vector in registers (PK_ARK_PAIR=0: round 2's kernel)
      const bool pair = pk::net_arkp_fits(n->d, n->max_sites);
      hipError_t ea = pair ? pk::launch_net_arkp(n->d, aa, n->nnzT, n->max_sites, (long long)B, stream)
                           : pk::launch_net_ark(n->d, aa, n->max_sites, (long long)B, threads_a, lds_a, stream);
      if (ea == hipSuccess) ea = hipGetLastError();
      return ea == hipSuccess ? PK_OK : pk_ctx_fail(c, PK_ERR_HIP, hipGetErrorString(ea));
    }
  }
  if (comb_reg && o.linsolve != PK_LINSOLVE_STRUCTURED) {
    const int threads2 = ((n->d.N + 63) / 64) * 64;
    if (n->max_sites <= 2) hipLaunchKernelGGL((pk::net_solve_reg2_kernel<2>), dim3((unsigned)B), dim3(threads2), n->solve_reg_lds_bytes, stream, n->d, a);
    else                   hipLaunchKernelGGL((pk::net_solve_reg2_kernel<3>), dim3((unsigned)B), dim3(threads2), n->solve_reg_lds_bytes, stream, n->d, a);
    hipError_t e2 = hipGetLastError();
    return e2 == hipSuccess ? PK_OK : pk_ctx_fail(c, PK_ERR_HIP, hipGetErrorString(e2));
  }
  const bool reg_ok = n->d.model != 2 && n->d.N <= 256 && n->max_sites <= 8 && n->solve_reg_lds_bytes <= 64 * 1024 && o.linsolve != PK_LINSOLVE_STRUCTURED;
  if (reg_ok) {
    const int threads = ((n->d.N + 63) / 64) * 64;
    const size_t lb = n->solve_reg_lds_bytes;
#define PK_REG_LAUNCH(M, MS) hipLaunchKernelGGL((pk::net_solve_reg_kernel<M, MS>), dim3((unsigned)B), dim3(threads), lb, stream, n->d, a)
    if (n->max_sites <= 4) {
      if (n->d.model == 0) PK_REG_LAUNCH(0, 4); else if (n->d.model == 1) PK_REG_LAUNCH(1, 4); else PK_REG_LAUNCH(4, 4);
    } else if (n->max_sites <= 6) {
      if (n->d.model == 0) PK_REG_LAUNCH(0, 6); else if (n->d.model == 1) PK_REG_LAUNCH(1, 6); else PK_REG_LAUNCH(4, 6);
    } else {
      if (n->d.model == 0) PK_REG_LAUNCH(0, 8); else if (n->d.model == 1) PK_REG_LAUNCH(1, 8); else PK_REG_LAUNCH(4, 8);
    }
#undef PK_REG_LAUNCH
  } else {
    // LDS kernel: every thread owns <= 4 states and <= 2 proteins (register-cached contexts in the kernel)
    int threads = (n->d.S <= 128 && n->d.N <= 64) ? 64 : 256;        // measured at S = 500: 256 threads beat 128 by 1.33x
    if (const char* e = getenv("PK_NET_THREADS")) { const int v = atoi(e); if ((v == 64 || v == 128 || v == 256) && n->d.S <= 4 * v && n->d.N <= 2 * v) threads = v; }
    if (n->d.model == 0)      hipLaunchKernelGGL(pk::net_solve_kernel<0>, dim3((unsigned)B), dim3(threads), n->solve_lds_bytes, stream, n->d, a);
    else if (n->d.model == 1) hipLaunchKernelGGL(pk::net_solve_kernel<1>, dim3((unsigned)B), dim3(threads), n->solve_lds_bytes, stream, n->d, a);
    else if (n->d.model == 2) hipLaunchKernelGGL(pk::net_solve_kernel<2>, dim3((unsigned)B), dim3(threads), n->solve_lds_bytes, stream, n->d, a);
    else                      hipLaunchKernelGGL(pk::net_solve_kernel<4>, dim3((unsigned)B), dim3(threads), n->solve_lds_bytes, stream, n->d, a);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PK_OK : pk_ctx_fail(c, PK_ERR_HIP, hipGetErrorString(e));
}

int pk_network_simulate_batch(pk_ctx* c, pk_net* n, int64_t B, const double* x, int x_is_raw, const double* y0, int y0_is_batched,
                              const double* t_host, int T, const pk_solver_opts* opts_in, double* Y, int32_t* status, int32_t* n_steps) {
  return net_simulate_impl(c, n, B, x, x_is_raw, y0, y0_is_batched, t_host, T, opts_in, Y, status, n_steps, nullptr);
}

// simulate + three-objective loss in ONE launch (SURVEY fused op (i), VERDICT r2 item 4 iii): the integrator scores the observations of the
// loss handle at its output times; the trajectory is written only when Y != null.  Same F / loss_sums as pk_network_simulate_batch followed
// by pk_network_objective_batch (up to the order of the sums).  PK_ERR_UNSUPPORTED when the network / options / loss data do not take the path.
int pk_loss_fused_tables(const pk_loss* l, const double** obs, const double** w, double* norms, int* T, int* rna_base);
int pk_network_simulate_objective_batch(pk_ctx* c, pk_net* n, pk_loss* l, int64_t B, const double* x, int x_is_raw, const double* y0,
                                        int y0_is_batched, const double* t_host, int T, const pk_solver_opts* opts, int loss_mode,
                                        const double* defaults, const double* lambdas, double fail_value, double* Y, int32_t* status,
                                        int32_t* n_steps, double* loss_sums, double* F) {
  if (!c || !n || !l) return PK_ERR_ARG;
  if (B == 0) return PK_OK;
  if (!loss_sums && !F) return pk_ctx_fail(c, PK_ERR_ARG, "null pointer");
  if (F && !lambdas) return pk_ctx_fail(c, PK_ERR_ARG, "lambdas (protein, rna, phospho, prior) are required for F");
  if (loss_mode < 0 || loss_mode > 7) return pk_ctx_fail(c, PK_ERR_ARG, "loss_mode must be 0..7");
  pk::NetSolveArgs f;
  std::memset(&f, 0, sizeof(f));
  int Tl = 0;
  if (!pk_loss_fused_tables(l, &f.loss_obs, &f.loss_w, f.loss_norm, &Tl, &f.loss_rna_base))
    return pk_ctx_fail(c, PK_ERR_UNSUPPORTED, "fused objective: protein / phospho baselines must be time index 0, rna observations not earlier than "
                                              "the rna baseline, and no (state, time) observed twice");
  if (Tl != T) return pk_ctx_fail(c, PK_ERR_ARG, "T differs from the grid the loss data were created for");
  f.loss_defaults = defaults; f.loss_mode = loss_mode; f.loss_fail = fail_value; f.loss_sums = loss_sums; f.loss_F = F;
  f.loss_lam[0] = lambdas ? lambdas[0] : 1.0; f.loss_lam[1] = lambdas ? lambdas[1] : 1.0; f.loss_lam[2] = lambdas ? lambdas[2] : 1.0;
  f.loss_lam[3] = lambdas ? lambdas[3] : 0.0;
  return net_simulate_impl(c, n, B, x, x_is_raw, y0, y0_is_batched, t_host, T, opts, Y, status, n_steps, &f);
}

int pk_network_unpack_batch(pk_ctx* c, pk_net* n, int64_t B, const double* x_raw, double* x_phys) {
  if (!c || !n) return PK_ERR_ARG;
  if (B < 0) return pk_ctx_fail(c, PK_ERR_ARG, "B must be >= 0");
  if (B == 0) return PK_OK;
  if (!x_raw || !x_phys) return pk_ctx_fail(c, PK_ERR_ARG, "null pointer");
  const long long total = (long long)B * n->d.n_var;
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return pk_ctx_fail(c, PK_ERR_HIP, "hipSetDevice");
  hipLaunchKernelGGL(pk::net_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)pk_ctx_stream(c), x_raw, x_phys, total);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PK_OK : pk_ctx_fail(c, PK_ERR_HIP, hipGetErrorString(e));
}

}  // extern "C"
