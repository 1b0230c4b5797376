// pk_frechet.hip -- batched discrete Frechet distance between observed and predicted time courses.
//
// Reference: frechet/distance.py:9-56 (frechet_distance: pairwise Euclidean distances of the 2-D points (time, value), then the
// dynamic programme  c[i][j] = max(min(c[i-1][j], c[i][j-1], c[i-1][j-1]), d[i][j])) as used by the Pareto pick loop of
// global_model/runner.py:780-841, which rebuilds DataFrames and calls it once per (solution, protein / site).  Here one thread
// owns one (candidate, series) pair and keeps a rolling DP row in registers; the predicted values come straight from the
// fold-change array of pk_network_observables_batch.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/phoskin.h"

struct pk_ctx;
extern "C" int pk_ctx_device(pk_ctx*);
extern "C" void* pk_ctx_stream(pk_ctx*);
extern "C" int pk_ctx_fail(pk_ctx*, int code, const char* msg);

namespace pk {

constexpr int FRECHET_MAX_PTS = 32;

__global__ __launch_bounds__(256) void frechet_kernel(const long long total, const int n_series, const int32_t* __restrict__ obs_ptr,
                                                      const double* __restrict__ obs_t, const double* __restrict__ obs_v,
                                                      const int32_t* __restrict__ pred_ptr, const double* __restrict__ pred_t,
                                                      const int32_t* __restrict__ pred_idx, const double* __restrict__ pred, const int n_obs,
                                                      double* __restrict__ out) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const int s = (int)(g % n_series);
  const long long b = g / n_series;
  const int o0 = obs_ptr[s], n = obs_ptr[s + 1] - o0;        // true curve: n points
  const int p0 = pred_ptr[s], m = pred_ptr[s + 1] - p0;      // predicted curve: m points
  if (n < 1 || m < 1) { out[g] = 0.0; return; }
  const double* pr = pred + b * n_obs;
  double row[FRECHET_MAX_PTS];                               // c[i][*] of the current i
  double py[FRECHET_MAX_PTS], px[FRECHET_MAX_PTS];
#pragma unroll
  for (int j = 0; j < FRECHET_MAX_PTS; ++j) {
    if (j < m) { px[j] = pred_t[p0 + j]; py[j] = pr[pred_idx[p0 + j]]; } else { px[j] = 0.0; py[j] = 0.0; }
    row[j] = 0.0;
  }
  for (int i = 0; i < n; ++i) {
    const double tx = obs_t[o0 + i], ty = obs_v[o0 + i];
    double diag = 0.0, left = 0.0;                           // c[i-1][j-1], c[i][j-1]
#pragma unroll
    for (int j = 0; j < FRECHET_MAX_PTS; ++j) {
      if (j < m) {
        const double dx = tx - px[j], dy = ty - py[j];
        const double d = sqrt(dx * dx + dy * dy);
        const double up = row[j];                            // c[i-1][j]
        double c;
        if (i == 0) c = (j == 0) ? d : fmax(left, d);
        else if (j == 0) c = fmax(up, d);
        else c = fmax(fmin(fmin(up, left), diag), d);
        diag = up; left = c; row[j] = c;
      }
    }
  }
  double r = 0.0;
#pragma unroll
  for (int j = 0; j < FRECHET_MAX_PTS; ++j) if (j == m - 1) r = row[j];
  out[g] = r;
}

}  // namespace pk

extern "C" int pk_frechet_batch(pk_ctx* c, int64_t B, int n_series, const int32_t* obs_ptr, const double* obs_t, const double* obs_v,
                                const int32_t* pred_ptr, const double* pred_t, const int32_t* pred_idx, const double* pred, int n_obs,
                                int max_points, double* out) {
  if (!c) return PK_ERR_ARG;
  if (B < 0 || n_series < 0 || n_obs < 1) return pk_ctx_fail(c, PK_ERR_ARG, "B, n_series must be >= 0 and n_obs >= 1");
  if (max_points > pk::FRECHET_MAX_PTS) return pk_ctx_fail(c, PK_ERR_UNSUPPORTED, "curves of at most 32 points");
  if (B == 0 || n_series == 0) return PK_OK;
  if (!obs_ptr || !obs_t || !obs_v || !pred_ptr || !pred_t || !pred_idx || !pred || !out) return pk_ctx_fail(c, PK_ERR_ARG, "null pointer");
  const long long total = (long long)B * n_series;
  if ((total + 255) / 256 > 0x7fffffffLL) return pk_ctx_fail(c, PK_ERR_ARG, "batch too large for one launch");
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return pk_ctx_fail(c, PK_ERR_HIP, "hipSetDevice");
  hipLaunchKernelGGL(pk::frechet_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)pk_ctx_stream(c), total, n_series, obs_ptr,
                     obs_t, obs_v, pred_ptr, pred_t, pred_idx, pred, n_obs, out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PK_OK : pk_ctx_fail(c, PK_ERR_HIP, hipGetErrorString(e));
}
