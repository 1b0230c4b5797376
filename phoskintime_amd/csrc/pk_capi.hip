// pk_capi.hip -- the C ABI of libphoskin_hip.so (include/phoskin.h): argument checking, launch geometry,
// context / stream management and the host-pointer convenience variants.  No kernels here (pk_solve_kernel.hpp).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <string>
#include "../../include/phoskin.h"
#include "pk_launch.hpp"
#include "pk_sens.hpp"

// A growable device buffer that outlives calls: the `_host` entry points stage through one, kernels that need per-replica HBM scratch
// use another (two, so that an inner device-pointer call can never move the staging area of the `_host` call around it).
struct pk_arena {
  void* p = nullptr; size_t bytes = 0; long long allocs = 0;
};
struct pk_ctx {
  int device;
  hipStream_t own_stream;
  hipStream_t stream;
  std::string err;
  hipEvent_t ev0, ev1;
  pk_arena stage, scratch;                       // device memory
  void* pin = nullptr; size_t pin_bytes = 0; long long pin_allocs = 0;      // page-locked host staging for small calls
  // The header asks for one context per thread, but a shared one must not corrupt memory: every `_host` entry point holds `mu` from
  // staging to the final synchronisation (they share `stage` / `pin`), and launches that use `scratch` are ordered by `scratch_ev`
  // across streams (pk_set_stream may change the stream between two calls).  Growing an arena drains the whole device first.
  std::recursive_mutex mu;
  hipEvent_t scratch_ev = nullptr; bool scratch_used = false;
};

namespace {

int fail(pk_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}
#define PK_HIP(ctx, call)                                                                     \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) return fail(ctx, PK_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

// Make `a` hold at least `bytes` (grow-only, 1.5x + 64 KB slack so that a slowly growing batch does not reallocate every call).
// Growing frees the old buffer, so the DEVICE is drained first: work queued by earlier calls -- on this stream or on one a caller set
// before (pk_set_stream) -- may still read it.  Growth is rare (1.5x), so the device-wide wait costs nothing in steady state.
int arena_reserve(pk_ctx* c, pk_arena& a, size_t bytes) {
  if (bytes <= a.bytes) return PK_OK;
  const size_t want = bytes + bytes / 2 + (64u << 10);
  if (a.p) { PK_HIP(c, hipDeviceSynchronize()); PK_HIP(c, hipFree(a.p)); a.p = nullptr; a.bytes = 0; }
  hipError_t e = hipMalloc(&a.p, want);
  if (e != hipSuccess) { a.p = nullptr; return fail(c, PK_ERR_NOMEM, std::string("hipMalloc of ") + std::to_string(want) + " bytes: " + hipGetErrorString(e)); }
  a.bytes = want; ++a.allocs;
  return PK_OK;
}
int pin_reserve(pk_ctx* c, size_t bytes) {
  if (bytes <= c->pin_bytes) return PK_OK;
  const size_t want = bytes + bytes / 2 + (64u << 10);
  if (c->pin) { PK_HIP(c, hipDeviceSynchronize()); PK_HIP(c, hipHostFree(c->pin)); c->pin = nullptr; c->pin_bytes = 0; }
  hipError_t e = hipHostMalloc(&c->pin, want, hipHostMallocDefault);
  if (e != hipSuccess) { c->pin = nullptr; return fail(c, PK_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
  c->pin_bytes = want; ++c->pin_allocs;
  return PK_OK;
}
constexpr size_t kAlign = 256;
size_t aligned(size_t b) { return (b + kAlign - 1) / kAlign * kAlign; }
// calls whose host arrays total at most this many bytes travel through ONE packed page-locked buffer (one copy each way);
// larger ones are copied array by array straight from / to the caller's (pageable) memory
constexpr size_t kPackedLimit = 4u << 20;

bool resolvent_method(int m) { return m == PK_METHOD_RODAS4 || m == PK_METHOD_LRP8 || m == PK_METHOD_LRP12; }
int group_width(int S) { return S <= 8 ? 8 : S <= 16 ? 16 : S <= 32 ? 32 : S <= 64 ? 64 : 0; }

int check_model(pk_ctx* c, int model, int n_sites) {
  if (model < 0 || model > 2) return fail(c, PK_ERR_ARG, "model must be 0 (dist), 1 (succ) or 2 (rand)");
  if (n_sites < 1) return fail(c, PK_ERR_ARG, "n_sites must be >= 1");
  if (model == PK_MODEL_RAND && n_sites > 20) return fail(c, PK_ERR_UNSUPPORTED, "randmod: n_sites <= 20 (2^n states, 2^n + n + 3 parameters per replica)");
  if (model != PK_MODEL_RAND && pk::n_states(model, n_sites) > 64 && !pk::wide_chain_fits(pk::n_states(model, n_sites), n_sites))
    return fail(c, PK_ERR_UNSUPPORTED, "distmod / succmod: n_sites <= 1276 (sixteen LDS vectors of n_sites + 2 doubles per workgroup)");
  return PK_OK;
}
// systems beyond one wavefront's lane groups (pk_wide.hpp): distmod / succmod with more than 64 states, randmod with n_sites >= 7
bool is_wide(int model, int n_sites) { return model == PK_MODEL_RAND ? n_sites >= 7 : pk::n_states(model, n_sites) > 64; }


int gidx(int G) { return G == 8 ? 0 : G == 16 ? 1 : G == 32 ? 2 : 3; }
#define PK_ROW(base, M) {pk::base##M##_g8, pk::base##M##_g16, pk::base##M##_g32, pk::base##M##_g64}
const pk::SolveLauncher kSolve[3][4] = {PK_ROW(launch_solve_m, 0), PK_ROW(launch_solve_m, 1), PK_ROW(launch_solve_m, 2)};
const pk::RhsLauncher kRhs[3][4] = {PK_ROW(launch_rhs_m, 0), PK_ROW(launch_rhs_m, 1), PK_ROW(launch_rhs_m, 2)};
const pk::JacLauncher kJac[3][4] = {PK_ROW(launch_jac_m, 0), PK_ROW(launch_jac_m, 1), PK_ROW(launch_jac_m, 2)};
const pk::SteadyLauncher kSteady[3][4] = {PK_ROW(launch_steady_m, 0), PK_ROW(launch_steady_m, 1), PK_ROW(launch_steady_m, 2)};
#undef PK_ROW

}  // namespace

extern "C" {

int pk_version(void) { return PK_VERSION; }

void pk_default_opts(pk_solver_opts* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->method = PK_METHOD_LRP12;
  o->linsolve = PK_LINSOLVE_AUTO;
  o->rtol = 1e-6;         // with LRP12: worst trajectory error over all golden fixtures = 0.05 of the rtol 1e-6 / atol 1e-8 parity band
  o->atol = 1e-8;         // (tools/gpu_band_scan.py); the lower-order methods need 1e-7 / 1e-9 for the same margin
  o->h0 = 0.0;
  o->rk4_h = 1e-3;
  o->max_steps = 100000;
  o->clip_nonneg = 1;
  o->normalize = 0;
  o->kernel = PK_KERNEL_AUTO;
  o->err_norm = PK_NORM_DEFAULT;
}

int pk_protein_n_states(int model, int n_sites) {
  if (model < 0 || model > 2 || n_sites < 1 || (model == PK_MODEL_RAND && n_sites > 20)) return PK_ERR_ARG;
  return pk::n_states(model, n_sites);
}
int pk_protein_n_params(int model, int n_sites) {
  if (model < 0 || model > 2 || n_sites < 1 || (model == PK_MODEL_RAND && n_sites > 20)) return PK_ERR_ARG;
  return pk::n_params(model, n_sites);
}
int pk_protein_flat_len(int model, int n_sites, int T) {
  if (model < 0 || model > 2 || n_sites < 1 || T < 1) return PK_ERR_ARG;
  return (T > 5 ? T - 5 : 0) + T + n_sites * T;
}

static thread_local std::string g_create_err;
const char* pk_create_error(void) { return g_create_err.c_str(); }

pk_ctx* pk_create(int device_id) {
  g_create_err.clear();
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess) { g_create_err = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); return nullptr; }
  if (ndev <= 0 || device_id < 0 || device_id >= ndev) { g_create_err = "device " + std::to_string(device_id) + " not in [0, " + std::to_string(ndev) + ")"; return nullptr; }
  if ((e = hipSetDevice(device_id)) != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e); return nullptr; }
  pk_ctx* c = new pk_ctx();
  c->device = device_id;
  if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) { g_create_err = std::string("hipStreamCreate: ") + hipGetErrorString(e); delete c; return nullptr; }
  c->stream = c->own_stream;
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
      hipEventCreateWithFlags(&c->scratch_ev, hipEventDisableTiming) != hipSuccess) { g_create_err = "hipEventCreate failed"; delete c; return nullptr; }
  return c;
}

void pk_destroy(pk_ctx* c) {
  if (!c) return;
  (void)pk_comm_destroy(c);                      // the communicator of the C-ABI collective (pk_comm.hip), if this context owns one
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->stage.p) (void)hipFree(c->stage.p);
  if (c->scratch.p) (void)hipFree(c->scratch.p);
  if (c->pin) (void)hipHostFree(c->pin);
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  if (c->scratch_ev) (void)hipEventDestroy(c->scratch_ev);
  (void)hipStreamDestroy(c->own_stream);
  delete c;
}

const char* pk_last_error(pk_ctx* c) { return c ? c->err.c_str() : "null context"; }

// accessors for the other translation units (pk_network.hip)
int pk_ctx_device(pk_ctx* c) { return c->device; }
void* pk_ctx_stream(pk_ctx* c) { return (void*)c->stream; }
int pk_ctx_fail(pk_ctx* c, int code, const char* msg) { return fail(c, code, msg ? msg : ""); }

int pk_workspace_stats(pk_ctx* c, int64_t out[6]) {
  if (!c || !out) return PK_ERR_ARG;
  out[0] = c->stage.allocs; out[1] = (int64_t)c->stage.bytes; out[2] = c->scratch.allocs; out[3] = (int64_t)c->scratch.bytes;
  out[4] = c->pin_allocs; out[5] = (int64_t)c->pin_bytes;
  return PK_OK;
}

int pk_set_stream(pk_ctx* c, void* s) {
  if (!c) return PK_ERR_ARG;
  c->stream = (hipStream_t)s;
  return PK_OK;
}

int pk_use_own_stream(pk_ctx* c) {
  if (!c) return PK_ERR_ARG;
  c->stream = c->own_stream;
  return PK_OK;
}

int pk_synchronize(pk_ctx* c) {
  if (!c) return PK_ERR_ARG;
  PK_HIP(c, hipStreamSynchronize(c->stream));
  return PK_OK;
}

int pk_solve_protein_batch(pk_ctx* c, int model, int n_sites, int64_t B, const double* theta, const double* y0,
                           int y0_is_batched, const double* t, int T, const pk_solver_opts* opts_in, double* sol,
                           double* flat, double* metric, int metric_id, int32_t* status, int32_t* n_steps) {
  if (!c) return PK_ERR_ARG;
  int rc = check_model(c, model, n_sites);
  if (rc) return rc;
  if (B < 0) return fail(c, PK_ERR_ARG, "B must be >= 0");
  if (T < 1) return fail(c, PK_ERR_ARG, "T must be >= 1");
  if (B == 0) return PK_OK;
  if (!theta || !y0 || !t) return fail(c, PK_ERR_ARG, "theta, y0 and t must be non-null");
  if (metric && (metric_id < 0 || metric_id > 4)) return fail(c, PK_ERR_ARG, "unknown metric_id");
  pk_solver_opts o;
  if (opts_in) o = *opts_in; else pk_default_opts(&o);
  if (o.method < 0 || o.method > 5 || o.method == PK_METHOD_DP5) return fail(c, PK_ERR_ARG, "unknown method (PK_METHOD_DP5 is a network integrator)");
  if (o.method != PK_METHOD_RK4 && !(o.rtol > 0.0 && o.atol >= 0.0)) return fail(c, PK_ERR_ARG, "rtol must be > 0 and atol >= 0");
  if (o.method == PK_METHOD_RK4 && !(o.rk4_h > 0.0)) return fail(c, PK_ERR_ARG, "rk4_h must be > 0");
  if (o.max_steps <= 0) o.max_steps = 100000;

  pk::SolveArgs a;
  a.theta = theta; a.y0 = y0; a.t = t; a.sol = sol; a.flat = flat; a.metric = metric; a.status = status; a.n_steps = n_steps;
  a.B = B; a.n_sites = n_sites; a.S = pk::n_states(model, n_sites); a.P = pk::n_params(model, n_sites); a.T = T;
  a.F = pk_protein_flat_len(model, n_sites, T); a.n_obs = n_sites; a.y0_batched = y0_is_batched ? 1 : 0; a.metric_id = metric_id;
  a.rtol = o.rtol; a.atol = o.atol; a.h0 = o.h0; a.rk4_h = o.rk4_h; a.max_steps = o.max_steps; a.clip = o.clip_nonneg; a.normalize = o.normalize; a.stage_form = o.stage_form;

  if (o.kernel < PK_KERNEL_AUTO || o.kernel > PK_KERNEL_TPR) return fail(c, PK_ERR_ARG, "unknown opts->kernel");
  if (is_wide(model, n_sites)) {
    // one workgroup per replica (pk_wide.hpp).  distmod / succmod: LRP12 with exact structured solves; randmod: ROS34PW2-W on the n-cube
    // (selected by any of the implicit one-step methods: there is no exact sparse resolvent for the LRP / RODAS family at this size)
    if (B > 0x7fffffffLL) return fail(c, PK_ERR_ARG, "batch too large for one launch");
    PK_HIP(c, hipSetDevice(c->device));
    if (model == PK_MODEL_RAND) {
      if (!resolvent_method(o.method) || o.stage_form)
        return fail(c, PK_ERR_UNSUPPORTED, "randmod n_sites >= 7: method must be LRP12 / LRP8 / RODAS4 in resolvent form (n = 7: LRP12 with the dense inverse; beyond: additive Runge-Kutta on the n-cube)");
      // PK_WIDE_RAND_EXACT (read once) picks among the exact kernels and the approximate one -- the tests hold them to agreement:
      //   1 (default)  parity elimination: the odd-popcount block of M is diagonal, the even Schur complement (64 x 64 at n = 7, 128 x 128
      //                at n = 8) is inverted in registers (pk_rand_parity.hpp)
      //   2            n = 7: the full 128 x 128 inverse in registers (pk_rand_dense.hpp, round 2's kernel); n = 8: block elimination over
      //                the popcount levels with the Schur complements in LDS (pk_rand_level.hpp)
      //   0            the n-cube kernel below (approximate factorisation), which stays the path for n >= 9
      static const int exact_env = [] { const char* v = getenv("PK_WIDE_RAND_EXACT"); return v ? atoi(v) : 1; }();
      if (n_sites == 7 && exact_env == 1 && pk::rand_dense_available(7)) { PK_HIP(c, pk::launch_rand_parity(a, c->stream, o.kernel != PK_KERNEL_AUTO)); return PK_OK; }
      if (pk::rand_dense_available(n_sites) && exact_env != 0) {   // n = 7, PK_WIDE_RAND_EXACT=2 (PK_WIDE_RAND_DENSE=0 also selects the n-cube kernel)
        PK_HIP(c, pk::launch_rand_dense(a, c->stream));
        return PK_OK;
      }
      if (n_sites == 8 && exact_env != 0) {
        PK_HIP(c, exact_env == 2 ? pk::launch_rand_level(a, c->stream) : pk::launch_rand_parity(a, c->stream, o.kernel != PK_KERNEL_AUTO));
        return PK_OK;
      }
      double* scr = nullptr;
      if (!pk::wide_rand_in_lds(n_sites)) {
        // the scratch rows are shared by every launch of this context: reserve + launch under the lock, and order the launch after the
        // previous scratch user whatever stream that one ran on
        std::lock_guard<std::recursive_mutex> g(c->mu);
        rc = arena_reserve(c, c->scratch, pk::wide_rand_scratch_bytes(n_sites, B));
        if (rc) return rc;
        scr = (double*)c->scratch.p;
        if (c->scratch_used) PK_HIP(c, hipStreamWaitEvent(c->stream, c->scratch_ev, 0));
        PK_HIP(c, pk::launch_wide_rand(a, scr, c->stream));
        PK_HIP(c, hipEventRecord(c->scratch_ev, c->stream));
        c->scratch_used = true;
      } else
        PK_HIP(c, pk::launch_wide_rand(a, scr, c->stream));
    } else {
      if (o.method != PK_METHOD_LRP12 || o.stage_form)
        return fail(c, PK_ERR_UNSUPPORTED, "distmod / succmod with more than 64 states integrate with method LRP12 (the default) only");
      PK_HIP(c, pk::launch_wide_chain(a, model, c->stream));
    }
    PK_HIP(c, hipGetLastError());
    return PK_OK;
  }
  // A/B switch of the tests: PK_RAND_LEVEL6=1 (read once) runs randmod n = 6 on the level-block kernel instead of the one-wave kernel
  static const int level6_env = [] { const char* v = getenv("PK_RAND_LEVEL6"); return v ? atoi(v) : 0; }();
  if (model == PK_MODEL_RAND && n_sites == 6 && level6_env == 1 && o.method == PK_METHOD_LRP12 && !o.stage_form) {
    if (B > 0x7fffffffLL) return fail(c, PK_ERR_ARG, "batch too large for one launch");
    PK_HIP(c, hipSetDevice(c->device));
    PK_HIP(c, pk::launch_rand_level(a, c->stream));
    return PK_OK;
  }
  // [r3] randmod n = 6 with the default method: parity elimination in ONE wave per replica (pk_rand_parity.hpp, 8 x 8 lanes over the 32 x 32
  // even Schur complement): 2.1-2.2 M replicas/s against 0.59 M of the 64 x 64 in-register inverse below (same box, B = 16 384 ... 65 536).
  // PK_RAND_PARITY56 (read once): 0 = the old kernel, 1 = also n = 5 (dev; 4 x 4 lanes per replica, four replicas per wave: 9.1-10.2 M against 8.2-9.4 M of the 32-lane kernel at B >= 16 384,
  // but 0.63 against 0.44 ms per launch at B = 7 -- not worth a second default; one wave per replica: 5.2-5.8 M)
  static const int parity56_env = [] { const char* v = getenv("PK_RAND_PARITY56"); return v ? atoi(v) : -1; }();
  if (model == PK_MODEL_RAND && ((n_sites == 6 && parity56_env != 0) || (n_sites == 5 && parity56_env == 1)) && o.method == PK_METHOD_LRP12 &&
      o.linsolve == PK_LINSOLVE_AUTO && !o.stage_form) {
    if (B > 0x7fffffffLL) return fail(c, PK_ERR_ARG, "batch too large for one launch");
    PK_HIP(c, hipSetDevice(c->device));
    PK_HIP(c, pk::launch_rand_parity(a, c->stream));
    return PK_OK;
  }
  const bool rand_fast = model == PK_MODEL_RAND && resolvent_method(o.method) && (o.linsolve == PK_LINSOLVE_AUTO || a.S > 64) && !o.stage_form;
  if (a.S > 64 && !rand_fast)       // n = 6: the in-register inverse of pk_rand_fast.hpp is the only solver (every `linsolve` value selects it)
    return fail(c, PK_ERR_UNSUPPORTED, "randmod n_sites = 6 (S = 65): only method RODAS4 / LRP8 in resolvent form (the generic kernels hold one state per lane)");
  const int G = a.S > 64 ? 64 : group_width(a.S);
  const long long rpb = 256 / G;
  const long long nblk = (B + rpb - 1) / rpb;
  if (nblk > 0x7fffffffLL) return fail(c, PK_ERR_ARG, "batch too large for one launch");
  const bool structured = (o.linsolve != PK_LINSOLVE_DENSE) && (model != PK_MODEL_RAND);
  PK_HIP(c, hipSetDevice(c->device));
  dim3 grid((unsigned)nblk);
  // small systems, large batches: one lane per replica (64 replicas per wave: the batch must be large enough to occupy the SIMDs).
  // Thresholds from tools/gpu_bench_dev.py tprB (crossover against the lane-group kernels).  opts->kernel pins the family (sharded runs
  // that must reproduce single-GPU bits); the PK_TPR=0 / 1 environment variable (read once per process) does the same for dev A/B runs.
  static const int tpr_env_once = [] { const char* v = getenv("PK_TPR"); return v ? atoi(v) : -1; }();
  const int tpr_env = o.kernel == PK_KERNEL_GROUP ? 0 : o.kernel == PK_KERNEL_TPR ? 1 : tpr_env_once;
  const long long tpr_min = (model == PK_MODEL_SUCC) ? (n_sites <= 8 ? 16384 : 32768) : (model == PK_MODEL_RAND) ? 32768 : (n_sites <= 8 ? 32768 : 49152);
  const bool tpr = o.method == PK_METHOD_LRP12 && o.linsolve == PK_LINSOLVE_AUTO && !o.stage_form && pk::tpr_available(model, n_sites) &&
                   (tpr_env == 1 || (tpr_env != 0 && B >= tpr_min));
  if (tpr)
    PK_HIP(c, pk::launch_tpr(a, model, c->stream));
  else if (model == PK_MODEL_DIST && resolvent_method(o.method) && o.linsolve == PK_LINSOLVE_AUTO && !o.stage_form)
    pk::launch_dist_fast(a, o.method, c->stream);                      // throughput layout: 4-16 lanes per replica, shadowed R / P rows
  else if (rand_fast)
    pk::launch_rand_fast(a, o.method, c->stream);                  // 2^n lanes per replica, shadowed mRNA row
  else
    kSolve[model][gidx(G)](a, o.method, structured, grid, c->stream);
  PK_HIP(c, hipGetLastError());
  return PK_OK;
}

int pk_rhs_protein_batch(pk_ctx* c, int model, int n_sites, int64_t B, const double* theta, const double* y, double* dydt) {
  if (!c) return PK_ERR_ARG;
  int rc = check_model(c, model, n_sites);
  if (rc) return rc;
  if (B < 0) return fail(c, PK_ERR_ARG, "B must be >= 0");
  if (B == 0) return PK_OK;
  if (!theta || !y || !dydt) return fail(c, PK_ERR_ARG, "null pointer");
  const int S = pk::n_states(model, n_sites), P = pk::n_params(model, n_sites), G = S > 64 ? 1 : group_width(S);
  const long long rpb = 256 / G, nblk = (S > 64 ? (B * S + 255) / 256 : (B + rpb - 1) / rpb);
  if (nblk > 0x7fffffffLL) return fail(c, PK_ERR_ARG, "batch too large for one launch");
  PK_HIP(c, hipSetDevice(c->device));
  dim3 grid((unsigned)nblk);
  if (S > 64 && model != PK_MODEL_RAND) pk::launch_chain_rhs_wide(model, theta, y, dydt, (long long)B, n_sites, S, P, c->stream);
  else if (S > 64) pk::launch_rand_rhs_wide(theta, y, dydt, (long long)B, n_sites, S, P, c->stream);
  else kRhs[model][gidx(G)](theta, y, dydt, (long long)B, n_sites, S, P, grid, c->stream);
  PK_HIP(c, hipGetLastError());
  return PK_OK;
}

int pk_jacobian_protein_batch(pk_ctx* c, int model, int n_sites, int64_t B, const double* theta, double* J) {
  if (!c) return PK_ERR_ARG;
  int rc = check_model(c, model, n_sites);
  if (rc) return rc;
  if (B < 0) return fail(c, PK_ERR_ARG, "B must be >= 0");
  if (B == 0) return PK_OK;
  if (!theta || !J) return fail(c, PK_ERR_ARG, "null pointer");
  const int S = pk::n_states(model, n_sites), P = pk::n_params(model, n_sites), G = S > 64 ? 1 : group_width(S);
  const long long rpb = 256 / G, nblk = (S > 64 ? (B * S + 255) / 256 : (B + rpb - 1) / rpb);
  if (nblk > 0x7fffffffLL) return fail(c, PK_ERR_ARG, "batch too large for one launch");
  PK_HIP(c, hipSetDevice(c->device));
  dim3 grid((unsigned)nblk);
  if (S > 64 && model != PK_MODEL_RAND) pk::launch_chain_jac_wide(model, theta, J, (long long)B, n_sites, S, P, c->stream);
  else if (S > 64) pk::launch_rand_jac_wide(theta, J, (long long)B, n_sites, S, P, c->stream);
  else kJac[model][gidx(G)](theta, J, (long long)B, n_sites, S, P, grid, c->stream);
  PK_HIP(c, hipGetLastError());
  return PK_OK;
}

int pk_steady_state_protein_batch(pk_ctx* c, int model, int n_sites, int64_t B, const double* theta, double* y_ss, int32_t* status) {
  if (!c) return PK_ERR_ARG;
  int rc = check_model(c, model, n_sites);
  if (rc) return rc;
  if (B < 0) return fail(c, PK_ERR_ARG, "B must be >= 0");
  if (B == 0) return PK_OK;
  if (!theta || !y_ss) return fail(c, PK_ERR_ARG, "null pointer");
  const int S = pk::n_states(model, n_sites), P = pk::n_params(model, n_sites);
  if (S > 64) {                                  // one workgroup per replica (pk_wide.hpp): closed form / cyclic reduction / Gauss-Seidel on the n-cube
    if (B > 0x7fffffffLL) return fail(c, PK_ERR_ARG, "batch too large for one launch");
    PK_HIP(c, hipSetDevice(c->device));
    hipError_t e = pk::launch_wide_steady(model, theta, y_ss, status, (long long)B, n_sites, S, P, c->stream);
    if (e == hipErrorInvalidValue) return fail(c, PK_ERR_UNSUPPORTED, "steady state: randmod n_sites <= 12 (two LDS vectors of 2^n doubles + the level table)");
    PK_HIP(c, e);
    PK_HIP(c, hipGetLastError());
    return PK_OK;
  }
  const int G = group_width(S);
  const long long rpb = 256 / G, nblk = (B + rpb - 1) / rpb;
  if (nblk > 0x7fffffffLL) return fail(c, PK_ERR_ARG, "batch too large for one launch");
  PK_HIP(c, hipSetDevice(c->device));
  kSteady[model][gidx(G)](theta, y_ss, status, (long long)B, n_sites, S, P, dim3((unsigned)nblk), c->stream);
  PK_HIP(c, hipGetLastError());
  return PK_OK;
}

// ------------------------------------------------------------------------------- host-pointer variants
// No hipMalloc / hipFree per call: device staging comes out of the context's grow-only arena.  Small calls (the reference's own call
// shape: ONE parameter vector per solve_ode call) additionally pack all inputs into one page-locked buffer and all outputs into
// another region of it, so a call is H2D copy -> kernel -> D2H copy -> one stream synchronisation.
}  // extern "C"
namespace {
struct Seg { const void* src; void* dst; size_t bytes; size_t off; };     // one host array and its offset in the staging block

struct HostCall {
  pk_ctx* c; Seg in[4]; int n_in = 0; Seg out[6]; int n_out = 0; size_t in_bytes = 0, total = 0;
  explicit HostCall(pk_ctx* ctx) : c(ctx) {}
  size_t add_in(const void* src, size_t bytes) { in[n_in++] = {src, nullptr, bytes, total}; const size_t o = total; total += aligned(bytes); in_bytes = total; return o; }
  size_t add_out(void* dst, size_t bytes) { if (!dst) return (size_t)-1; out[n_out++] = {nullptr, dst, bytes, total}; const size_t o = total; total += aligned(bytes); return o; }
  template <class T> T* dev(size_t off) const { return off == (size_t)-1 ? nullptr : reinterpret_cast<T*>((char*)c->stage.p + off); }
  bool packed() const { return total <= kPackedLimit; }
  int upload() {
    int rc = arena_reserve(c, c->stage, total);
    if (rc) return rc;
    if (packed()) {
      if ((rc = pin_reserve(c, total))) return rc;
      for (int i = 0; i < n_in; ++i) std::memcpy((char*)c->pin + in[i].off, in[i].src, in[i].bytes);
      PK_HIP(c, hipMemcpyAsync(c->stage.p, c->pin, in_bytes, hipMemcpyHostToDevice, c->stream));
    } else {
      for (int i = 0; i < n_in; ++i) PK_HIP(c, hipMemcpyAsync((char*)c->stage.p + in[i].off, in[i].src, in[i].bytes, hipMemcpyHostToDevice, c->stream));
    }
    return PK_OK;
  }
  int download() {
    if (n_out == 0) { PK_HIP(c, hipStreamSynchronize(c->stream)); return PK_OK; }
    if (packed()) {
      const size_t lo = out[0].off, hi = out[n_out - 1].off + out[n_out - 1].bytes;
      PK_HIP(c, hipMemcpyAsync((char*)c->pin + lo, (char*)c->stage.p + lo, hi - lo, hipMemcpyDeviceToHost, c->stream));
      PK_HIP(c, hipStreamSynchronize(c->stream));
      for (int i = 0; i < n_out; ++i) std::memcpy(out[i].dst, (char*)c->pin + out[i].off, out[i].bytes);
    } else {
      for (int i = 0; i < n_out; ++i) PK_HIP(c, hipMemcpyAsync(out[i].dst, (char*)c->stage.p + out[i].off, out[i].bytes, hipMemcpyDeviceToHost, c->stream));
      PK_HIP(c, hipStreamSynchronize(c->stream));
    }
    return PK_OK;
  }
};
}  // namespace
extern "C" {

int pk_protein_sens_available(int model, int n_sites) {
  if (model < 0 || model > 2 || n_sites < 1) return 0;
  return pk::sens_available(model, n_sites) ? 1 : 0;
}

int pk_solve_protein_sens_batch(pk_ctx* c, int model, int n_sites, int64_t B, const double* theta, const double* y0, int y0_is_batched,
                                const double* t, int T, const pk_solver_opts* opts_in, double* flat, double* dflat, int32_t* status,
                                int32_t* n_steps) {
  if (!c) return PK_ERR_ARG;
  int rc = check_model(c, model, n_sites);
  if (rc) return rc;
  if (B < 0) return fail(c, PK_ERR_ARG, "B must be >= 0");
  if (T < 1) return fail(c, PK_ERR_ARG, "T must be >= 1");
  if (!pk::sens_available(model, n_sites))
    return fail(c, PK_ERR_UNSUPPORTED, "forward sensitivities: distmod / succmod n_sites <= 62, randmod n_sites <= 7 (difference the batched solve beyond)");
  if (B == 0) return PK_OK;
  if (!theta || !y0 || !t || !flat || !dflat) return fail(c, PK_ERR_ARG, "theta, y0, t, flat and dflat must be non-null");
  pk_solver_opts o;
  if (opts_in) o = *opts_in; else pk_default_opts(&o);
  if (o.method != PK_METHOD_LRP12 || o.stage_form) return fail(c, PK_ERR_UNSUPPORTED, "forward sensitivities integrate with method LRP12 (the default) only");
  if (!(o.rtol > 0.0 && o.atol >= 0.0)) return fail(c, PK_ERR_ARG, "rtol must be > 0 and atol >= 0");
  if (o.max_steps <= 0) o.max_steps = 100000;
  pk::SensArgs sa;
  pk::SolveArgs& a = sa.s;
  a.theta = theta; a.y0 = y0; a.t = t; a.sol = nullptr; a.flat = flat; a.metric = nullptr; a.status = status; a.n_steps = n_steps;
  a.B = B; a.n_sites = n_sites; a.S = pk::n_states(model, n_sites); a.P = pk::n_params(model, n_sites); a.T = T;
  a.F = pk_protein_flat_len(model, n_sites, T); a.n_obs = n_sites; a.y0_batched = y0_is_batched ? 1 : 0; a.metric_id = 0;
  a.rtol = o.rtol; a.atol = o.atol; a.h0 = o.h0; a.rk4_h = o.rk4_h; a.max_steps = o.max_steps; a.clip = o.clip_nonneg; a.normalize = o.normalize; a.stage_form = 0;
  sa.dflat = dflat;
  if (B > 0x7fffffffLL) return fail(c, PK_ERR_ARG, "batch too large for one launch");
  PK_HIP(c, hipSetDevice(c->device));
  PK_HIP(c, pk::launch_sens(sa, model, c->stream));
  return PK_OK;
}

int pk_solve_protein_batch_host(pk_ctx* c, int model, int n_sites, int64_t B, const double* theta, const double* y0,
                                int y0_is_batched, const double* t, int T, const pk_solver_opts* opts, double* sol,
                                double* flat, double* metric, int metric_id, int32_t* status, int32_t* n_steps) {
  if (!c) return PK_ERR_ARG;
  std::lock_guard<std::recursive_mutex> host_guard(c->mu);          // `stage` / `pin` are shared by every `_host` call of this context
  int rc = check_model(c, model, n_sites);
  if (rc) return rc;
  if (B < 0 || T < 1) return fail(c, PK_ERR_ARG, "B must be >= 0 and T >= 1");
  if (B == 0) return PK_OK;
  if (!theta || !y0 || !t) return fail(c, PK_ERR_ARG, "theta, y0 and t must be non-null");
  const size_t S = pk::n_states(model, n_sites), P = pk::n_params(model, n_sites), F = pk_protein_flat_len(model, n_sites, T);
  PK_HIP(c, hipSetDevice(c->device));
  HostCall h(c);
  const size_t ny0 = (y0_is_batched ? (size_t)B : 1) * S;
  const size_t o_th = h.add_in(theta, (size_t)B * P * 8), o_y0 = h.add_in(y0, ny0 * 8), o_t = h.add_in(t, (size_t)T * 8);
  const size_t o_sol = h.add_out(sol, (size_t)B * T * S * 8), o_flat = h.add_out(flat, (size_t)B * F * 8), o_met = h.add_out(metric, (size_t)B * 8),
               o_st = h.add_out(status, (size_t)B * 4), o_ns = h.add_out(n_steps, (size_t)B * 8);
  if ((rc = h.upload())) return rc;
  rc = pk_solve_protein_batch(c, model, n_sites, B, h.dev<const double>(o_th), h.dev<const double>(o_y0), y0_is_batched, h.dev<const double>(o_t), T, opts,
                              h.dev<double>(o_sol), h.dev<double>(o_flat), h.dev<double>(o_met), metric_id, h.dev<int32_t>(o_st), h.dev<int32_t>(o_ns));
  if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
  return h.download();
}

int pk_solve_protein_sens_batch_host(pk_ctx* c, int model, int n_sites, int64_t B, const double* theta, const double* y0, int y0_is_batched,
                                     const double* t, int T, const pk_solver_opts* opts, double* flat, double* dflat, int32_t* status,
                                     int32_t* n_steps) {
  if (!c) return PK_ERR_ARG;
  std::lock_guard<std::recursive_mutex> host_guard(c->mu);          // `stage` / `pin` are shared by every `_host` call of this context
  int rc = check_model(c, model, n_sites);
  if (rc) return rc;
  if (B < 0 || T < 1) return fail(c, PK_ERR_ARG, "B must be >= 0 and T >= 1");
  if (!pk::sens_available(model, n_sites))
    return fail(c, PK_ERR_UNSUPPORTED, "forward sensitivities: distmod / succmod n_sites <= 62, randmod n_sites <= 7 (difference the batched solve beyond)");
  if (B == 0) return PK_OK;
  if (!theta || !y0 || !t || !flat || !dflat) return fail(c, PK_ERR_ARG, "theta, y0, t, flat and dflat must be non-null");
  const size_t S = pk::n_states(model, n_sites), P = pk::n_params(model, n_sites), F = pk_protein_flat_len(model, n_sites, T);
  PK_HIP(c, hipSetDevice(c->device));
  HostCall h(c);
  const size_t ny0 = (y0_is_batched ? (size_t)B : 1) * S;
  const size_t o_th = h.add_in(theta, (size_t)B * P * 8), o_y0 = h.add_in(y0, ny0 * 8), o_t = h.add_in(t, (size_t)T * 8);
  const size_t o_flat = h.add_out(flat, (size_t)B * F * 8), o_df = h.add_out(dflat, (size_t)B * F * P * 8), o_st = h.add_out(status, (size_t)B * 4),
               o_ns = h.add_out(n_steps, (size_t)B * 8);
  if ((rc = h.upload())) return rc;
  rc = pk_solve_protein_sens_batch(c, model, n_sites, B, h.dev<const double>(o_th), h.dev<const double>(o_y0), y0_is_batched, h.dev<const double>(o_t), T, opts,
                                   h.dev<double>(o_flat), h.dev<double>(o_df), h.dev<int32_t>(o_st), h.dev<int32_t>(o_ns));
  if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
  return h.download();
}

int pk_rhs_protein_batch_host(pk_ctx* c, int model, int n_sites, int64_t B, const double* theta, const double* y, double* dydt) {
  if (!c) return PK_ERR_ARG;
  std::lock_guard<std::recursive_mutex> host_guard(c->mu);          // `stage` / `pin` are shared by every `_host` call of this context
  int rc = check_model(c, model, n_sites);
  if (rc) return rc;
  if (B < 0) return fail(c, PK_ERR_ARG, "B must be >= 0");
  if (B == 0) return PK_OK;
  if (!theta || !y || !dydt) return fail(c, PK_ERR_ARG, "null pointer");
  const size_t S = pk::n_states(model, n_sites), P = pk::n_params(model, n_sites);
  PK_HIP(c, hipSetDevice(c->device));
  HostCall h(c);
  const size_t o_th = h.add_in(theta, (size_t)B * P * 8), o_y = h.add_in(y, (size_t)B * S * 8), o_f = h.add_out(dydt, (size_t)B * S * 8);
  if ((rc = h.upload())) return rc;
  rc = pk_rhs_protein_batch(c, model, n_sites, B, h.dev<const double>(o_th), h.dev<const double>(o_y), h.dev<double>(o_f));
  if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
  return h.download();
}

int pk_jacobian_protein_batch_host(pk_ctx* c, int model, int n_sites, int64_t B, const double* theta, double* J) {
  if (!c) return PK_ERR_ARG;
  std::lock_guard<std::recursive_mutex> host_guard(c->mu);          // `stage` / `pin` are shared by every `_host` call of this context
  int rc = check_model(c, model, n_sites);
  if (rc) return rc;
  if (B < 0) return fail(c, PK_ERR_ARG, "B must be >= 0");
  if (B == 0) return PK_OK;
  if (!theta || !J) return fail(c, PK_ERR_ARG, "null pointer");
  const size_t S = pk::n_states(model, n_sites), P = pk::n_params(model, n_sites);
  PK_HIP(c, hipSetDevice(c->device));
  HostCall h(c);
  const size_t o_th = h.add_in(theta, (size_t)B * P * 8), o_J = h.add_out(J, (size_t)B * S * S * 8);
  if ((rc = h.upload())) return rc;
  rc = pk_jacobian_protein_batch(c, model, n_sites, B, h.dev<const double>(o_th), h.dev<double>(o_J));
  if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
  return h.download();
}

double pk_time_solve_protein_batch(pk_ctx* c, int iters, int model, int n_sites, int64_t B, const double* theta,
                                   const double* y0, int y0_is_batched, const double* t, int T,
                                   const pk_solver_opts* opts, double* sol, double* flat, double* metric, int metric_id,
                                   int32_t* status, int32_t* n_steps) {
  if (!c || iters < 1) return -1.0;
  if (hipSetDevice(c->device) != hipSuccess) return -1.0;
  if (hipEventRecord(c->ev0, c->stream) != hipSuccess) return -1.0;
  for (int i = 0; i < iters; ++i) {
    int rc = pk_solve_protein_batch(c, model, n_sites, B, theta, y0, y0_is_batched, t, T, opts, sol, flat, metric, metric_id, status, n_steps);
    if (rc) return (double)rc;
  }
  if (hipEventRecord(c->ev1, c->stream) != hipSuccess) return -1.0;
  if (hipEventSynchronize(c->ev1) != hipSuccess) { c->err = "hipEventSynchronize failed"; return -1.0; }
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.0;
  return (double)ms / iters;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ score_fit (a6)
namespace pk {
// One wave per candidate: residual r = |target - pred| / N ;  score = delta sum r^2 + alpha sqrt(mean r^2) + beta mean r
//   + gamma var(r) + mu ||theta||_2 / len(theta)            (config/config.py:176-226)
__global__ __launch_bounds__(256) void score_fit_kernel(const double* __restrict__ theta, const int P, const double* __restrict__ target,
                                                        const double* __restrict__ pred, const int N, const long long B,
                                                        const double alpha, const double beta, const double gamma, const double delta,
                                                        const double mu, double* __restrict__ out) {
  const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const double* pb = pred + b * N;
  double s1 = 0.0, s2 = 0.0;
  for (int k = lane; k < N; k += 64) { const double r = fabs(target[k] - pb[k]) / (double)N; s1 += r; s2 = __builtin_fma(r, r, s2); }
  double t2 = 0.0;
  const double* tb = theta + b * P;
  for (int k = lane; k < P; k += 64) t2 = __builtin_fma(tb[k], tb[k], t2);
  for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); t2 += __shfl_xor(t2, off); }
  const double mean = s1 / N;
  // two-pass variance like np.var
  double v = 0.0;
  for (int k = lane; k < N; k += 64) { const double d = fabs(target[k] - pb[k]) / (double)N - mean; v = __builtin_fma(d, d, v); }
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if (lane == 0) out[b] = delta * s2 + alpha * sqrt(s2 / N) + beta * mean + gamma * (v / N) + mu * (sqrt(t2) / P);
}
}  // namespace pk

extern "C" int pk_score_fit_batch(pk_ctx* c, int64_t B, const double* theta, int P, const double* target, const double* pred, int N,
                                  const double* weights, double* out) {
  if (!c) return PK_ERR_ARG;
  if (B < 0 || P < 1 || N < 1) return fail(c, PK_ERR_ARG, "B >= 0, P >= 1, N >= 1 required");
  if (B == 0) return PK_OK;
  if (!theta || !target || !pred || !out) return fail(c, PK_ERR_ARG, "null pointer");
  const double a = weights ? weights[0] : 1.0, b = weights ? weights[1] : 1.0, g = weights ? weights[2] : 1.0, d = weights ? weights[3] : 1.0,
               m = weights ? weights[4] : 1.0;
  PK_HIP(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(pk::score_fit_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, c->stream, theta, P, target, pred, N, (long long)B, a, b, g, d, m, out);
  PK_HIP(c, hipGetLastError());
  return PK_OK;
}
