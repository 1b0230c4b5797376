// pk_peaks.hip -- the two machine peaks bench.py divides by, MEASURED on the box it runs on (BASELINE.md section 3 promised that; round 1
// hard-coded the spec-sheet values): a STREAM-style copy for HBM bandwidth and a dependent-chain-free FP64 FMA loop for the vector peak.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/phoskin.h"

extern "C" int pk_ctx_device(pk_ctx*);
extern "C" void* pk_ctx_stream(pk_ctx*);
extern "C" int pk_ctx_fail(pk_ctx*, int, const char*);

namespace pk {

// 16 bytes per lane per access, grid-stride: the access shape the microarchitecture guide calibrates FETCH_SIZE / WRITE_SIZE on
__global__ __launch_bounds__(256) void stream_copy_kernel(const double2* __restrict__ src, double2* __restrict__ dst, const size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {                      // four independent 16-byte loads in flight per lane
    const double2 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

// 16 independent FMA chains per lane (v_fma_f64: one wave64 instruction per 4 cycles per SIMD), 4 waves per SIMD resident
__global__ __launch_bounds__(256) void fma_f64_kernel(double* out, const int iters, const double a, const double b) {
  double x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = threadIdx.x * 1e-9 + k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = __builtin_fma(x[k], a, b);
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += x[k];
  if (s == 12345.678) out[0] = s;                   // never true: keeps the chains alive
}

}  // namespace pk

extern "C" {

// Sustained HBM copy rate in GB/s (bytes read + bytes written per second) over `iters` back-to-back copies of `bytes` bytes.
double pk_measure_hbm_gbs(pk_ctx* c, int64_t bytes, int iters) {
  if (!c || bytes < (1 << 20) || iters < 1) return -1.0;
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return -1.0;
  hipStream_t st = (hipStream_t)pk_ctx_stream(c);
  void *a = nullptr, *b = nullptr;
  if (hipMalloc(&a, (size_t)bytes) != hipSuccess) return -1.0;
  if (hipMalloc(&b, (size_t)bytes) != hipSuccess) { (void)hipFree(a); return -1.0; }
  (void)hipMemsetAsync(a, 0, (size_t)bytes, st);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t n = (size_t)bytes / sizeof(double2);
  const unsigned grid = 256 * 8;                     // 8 workgroups of 4 waves per CU
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(pk::stream_copy_kernel, dim3(grid), dim3(256), 0, st, (const double2*)a, (double2*)b, n);
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(pk::stream_copy_kernel, dim3(grid), dim3(256), 0, st, (const double2*)a, (double2*)b, n);
  (void)hipEventRecord(e1, st);
  double out = -1.0;
  if (hipEventSynchronize(e1) == hipSuccess) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f) out = 2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9;
  }
  // the runtime's own device-to-device copy as a second opinion: report the better of the two
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) (void)hipMemcpyAsync(b, a, (size_t)bytes, hipMemcpyDeviceToDevice, st);
  (void)hipEventRecord(e1, st);
  if (hipEventSynchronize(e1) == hipSuccess) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f) { const double r2 = 2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9; if (r2 > out) out = r2; }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b);
  return out;
}

// Sustained FP64 vector FMA rate in TFLOP/s (2 flops per FMA) with every SIMD holding 4 waves.
double pk_measure_fp64_fma_tflops(pk_ctx* c, int iters) {
  if (!c || iters < 1) return -1.0;
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return -1.0;
  hipStream_t st = (hipStream_t)pk_ctx_stream(c);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, pk_ctx_device(c)) != hipSuccess) return -1.0;
  const unsigned grid = (unsigned)prop.multiProcessorCount * 4;          // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  double* out = nullptr;
  if (hipMalloc((void**)&out, 8) != hipSuccess) return -1.0;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(pk::fma_f64_kernel, dim3(grid), dim3(256), 0, st, out, iters, 0.999999, 1e-9);
  (void)hipEventRecord(e0, st);
  hipLaunchKernelGGL(pk::fma_f64_kernel, dim3(grid), dim3(256), 0, st, out, iters, 0.999999, 1e-9);
  (void)hipEventRecord(e1, st);
  double r = -1.0;
  if (hipEventSynchronize(e1) == hipSuccess) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f) r = 2.0 * 16.0 * (double)iters * 256.0 * grid / (ms * 1e-3) / 1e12;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(out);
  return r;
}

}  // extern "C"
