// pk_morris.hip -- Morris screening around the solve: the sample matrix is BUILT in HBM from the (small) random draws and the
// elementary effects are taken from the per-replica outputs without leaving the GPU.
//
// Reference: sensitivity/analysis.py:221-265 and global_model/sensitivity.py:210-277 call SALib's morris.sample -> N*(D+1) x D host
// matrix -> process pool -> morris.analyze.  SALib is a third-party dependency absent here; the construction is the standard
// Morris (1991) trajectory design: trajectory r starts at base[r, :] on the level grid and moves coordinate i by sign[r, i] * delta at
// step rank[r, i] + 1, so row s of the trajectory is  x_i = base_i + sign_i * delta * [rank_i < s].
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/phoskin.h"

struct pk_ctx;
extern "C" int pk_ctx_device(pk_ctx*);
extern "C" void* pk_ctx_stream(pk_ctx*);
extern "C" int pk_ctx_fail(pk_ctx*, int code, const char* msg);

namespace pk {

// X[(r * (D + 1) + s) * D + i], one thread per element; unit-cube value clipped to [0, 1], then scaled to [lb_i, ub_i]
__global__ void morris_build_kernel(const long long total, const int D, const double delta, const double* __restrict__ base,
                                    const double* __restrict__ sign, const int32_t* __restrict__ rank, const double* __restrict__ lb,
                                    const double* __restrict__ ub, double* __restrict__ X) {
#pragma clang fp contract(off)      // separately rounded mul / add: the matrix is bit-identical to the host builder (sensitivity/morris.py build())
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const int i = (int)(g % D);
  const long long row = g / D;
  const int s = (int)(row % (D + 1));
  const long long r = row / (D + 1);
  const long long k = r * D + i;
  double u = base[k];
  if (rank[k] < s) u = u + sign[k] * delta;
  u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
  X[g] = lb[i] + u * (ub[i] - lb[i]);
}

// EE[r, i] = (Y[r, rank_i + 1] - Y[r, rank_i]) / (sign_i * delta)   (inputs on the unit cube, as SALib scales them)
__global__ void morris_effects_kernel(const long long total, const int D, const double delta, const double* __restrict__ sign,
                                      const int32_t* __restrict__ rank, const double* __restrict__ Y, double* __restrict__ EE) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= total) return;
  const long long r = k / D;
  const int s = rank[k];
  const double* y = Y + r * (D + 1);
  EE[k] = (y[s + 1] - y[s]) / (sign[k] * delta);
}

}  // namespace pk

extern "C" {

int pk_morris_build_batch(pk_ctx* c, int64_t N, int D, double delta, const double* base, const double* sign, const int32_t* rank,
                          const double* lb, const double* ub, double* X) {
  if (!c) return PK_ERR_ARG;
  if (N < 0 || D < 1) return pk_ctx_fail(c, PK_ERR_ARG, "N must be >= 0 and D >= 1");
  if (N == 0) return PK_OK;
  if (!base || !sign || !rank || !lb || !ub || !X) return pk_ctx_fail(c, PK_ERR_ARG, "null pointer");
  const long long total = (long long)N * (D + 1) * D;
  if ((total + 255) / 256 > 0x7fffffffLL) return pk_ctx_fail(c, PK_ERR_ARG, "sample too large for one launch");
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return pk_ctx_fail(c, PK_ERR_HIP, "hipSetDevice");
  hipLaunchKernelGGL(pk::morris_build_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)pk_ctx_stream(c), total, D, delta, base,
                     sign, rank, lb, ub, X);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PK_OK : pk_ctx_fail(c, PK_ERR_HIP, hipGetErrorString(e));
}

int pk_morris_effects_batch(pk_ctx* c, int64_t N, int D, double delta, const double* sign, const int32_t* rank, const double* Y, double* EE) {
  if (!c) return PK_ERR_ARG;
  if (N < 0 || D < 1 || !(delta > 0.0)) return pk_ctx_fail(c, PK_ERR_ARG, "N must be >= 0, D >= 1 and delta > 0");
  if (N == 0) return PK_OK;
  if (!sign || !rank || !Y || !EE) return pk_ctx_fail(c, PK_ERR_ARG, "null pointer");
  const long long total = (long long)N * D;
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return pk_ctx_fail(c, PK_ERR_HIP, "hipSetDevice");
  hipLaunchKernelGGL(pk::morris_effects_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)pk_ctx_stream(c), total, D, delta, sign,
                     rank, Y, EE);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PK_OK : pk_ctx_fail(c, PK_ERR_HIP, hipGetErrorString(e));
}

}  // extern "C"
