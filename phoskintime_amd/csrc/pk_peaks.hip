// pk_peaks.hip -- the two machine peaks bench.py divides by, MEASURED on the box it runs on (BASELINE.md section 3 promised that; round 1
// hard-coded the spec-sheet values): a STREAM-style copy for HBM bandwidth and a dependent-chain-free FP64 FMA loop for the vector peak.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/phoskin.h"

extern "C" int pk_ctx_device(pk_ctx*);
extern "C" void* pk_ctx_stream(pk_ctx*);
extern "C" int pk_ctx_fail(pk_ctx*, int, const char*);

namespace pk {

// 16 bytes per lane per access (the shape the microarchitecture guide calibrates FETCH_SIZE / WRITE_SIZE on), ONE tile of 4 x 256 x 16 B per
// workgroup, non-temporal loads and stores.  Measured on MI355X against the alternatives (build/dev/hbm.hip, round 3): grid-stride loops
// over a persistent grid 4.4-5.0 TB/s, a contiguous slab per persistent workgroup 5.3-5.6, one tile per workgroup 5.8-5.9 (copy);
// read-only 7.05-7.07 TB/s, write-only 5.6 TB/s -- a copy pays the read/write bus turnarounds, which is why it stays below the read rate.
typedef double pk_v2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void stream_copy_kernel(const pk_v2* __restrict__ src, pk_v2* __restrict__ dst, const size_t n) {
  const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
  pk_v2 r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = (i + k * 256 < n) ? __builtin_nontemporal_load(src + i + k * 256) : pk_v2{0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 4; ++k) if (i + k * 256 < n) __builtin_nontemporal_store(r[k], dst + i + k * 256);
}
__global__ __launch_bounds__(256) void stream_read_kernel(const pk_v2* __restrict__ src, pk_v2* __restrict__ dst, const size_t n) {
  const size_t i = (size_t)blockIdx.x * 2048 + threadIdx.x;
  pk_v2 acc{0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 8; ++k) if (i + k * 256 < n) acc += __builtin_nontemporal_load(src + i + k * 256);
  if (acc.x == 1.2345e300) dst[0] = acc;                     // never true: keeps the loads alive
}
__global__ __launch_bounds__(256) void stream_write_kernel(pk_v2* __restrict__ dst, const size_t n) {
  const size_t i = (size_t)blockIdx.x * 2048 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; ++k) if (i + k * 256 < n) __builtin_nontemporal_store(pk_v2{1.0, 2.0}, dst + i + k * 256);
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
  const size_t n = (size_t)bytes / sizeof(pk::pk_v2);
  const unsigned grid = (unsigned)((n + 1023) / 1024);      // one 16 KiB tile per workgroup
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(pk::stream_copy_kernel, dim3(grid), dim3(256), 0, st, (const pk::pk_v2*)a, (pk::pk_v2*)b, n);
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(pk::stream_copy_kernel, dim3(grid), dim3(256), 0, st, (const pk::pk_v2*)a, (pk::pk_v2*)b, n);
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

// Sustained one-directional HBM rate in GB/s: mode 0 = read-only (bytes read per second), 1 = write-only.
double pk_measure_hbm_stream_gbs(pk_ctx* c, int64_t bytes, int iters, int mode) {
  if (!c || bytes < (1 << 20) || iters < 1 || mode < 0 || mode > 1) return -1.0;
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return -1.0;
  hipStream_t st = (hipStream_t)pk_ctx_stream(c);
  void *a = nullptr, *b = nullptr;
  if (hipMalloc(&a, (size_t)bytes) != hipSuccess) return -1.0;
  if (hipMalloc(&b, 64) != hipSuccess) { (void)hipFree(a); return -1.0; }
  (void)hipMemsetAsync(a, 0, (size_t)bytes, st);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t n = (size_t)bytes / sizeof(pk::pk_v2);
  const unsigned grid = (unsigned)((n + 2047) / 2048);
  auto launch = [&] {
    if (mode == 0) hipLaunchKernelGGL(pk::stream_read_kernel, dim3(grid), dim3(256), 0, st, (const pk::pk_v2*)a, (pk::pk_v2*)b, n);
    else hipLaunchKernelGGL(pk::stream_write_kernel, dim3(grid), dim3(256), 0, st, (pk::pk_v2*)a, n);
  };
  for (int i = 0; i < 3; ++i) launch();
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) launch();
  (void)hipEventRecord(e1, st);
  double out = -1.0;
  if (hipEventSynchronize(e1) == hipSuccess) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f) out = (double)bytes * iters / (ms * 1e-3) / 1e9;
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
