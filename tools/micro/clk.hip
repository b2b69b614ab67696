// Micro-benchmark (diagnostic only): shader clock and fp64 / LDS chain speed of a small latency-bound kernel
// while a full-chip streaming clear runs on another stream.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float vfloat4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_clear(vfloat4 *p, size_t n) {
  const vfloat4 z = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(z, p + i);
}
__global__ __launch_bounds__(64) void k_lat(long long *out, double *sink, int iters) {
  __shared__ double s[512];
  for (int i = threadIdx.x; i < 512; i += 64) s[i] = 1.0 + i;
  __syncthreads();
  double    a  = threadIdx.x * 1e-3;
  long long w0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a = __builtin_fma(a, 1.0000001, s[(i + r) & 511]);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2 + 0] = c1 - c0;
    out[blockIdx.x * 2 + 1] = w1 - w0;
  }
  sink[blockIdx.x * 64 + threadIdx.x] = a;
}
int main() {
  const size_t bytes = 16ull << 30;
  vfloat4     *g;
  long long   *d, h[2048];
  double      *s;
  (void)hipMalloc(&g, bytes);
  (void)hipMalloc(&d, sizeof(h));
  (void)hipMalloc(&s, 1024 * 64 * 8);
  hipStream_t sa, sb;
  (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  const int iters = 40000, nwg = 896;
  for (int mode = 0; mode < 3; ++mode) {
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    if (mode >= 1)
      for (int r = 0; r < 6; ++r) hipLaunchKernelGGL(k_clear, dim3(mode == 1 ? 2048 : 256), dim3(256), 0, sa, g, bytes / 16);
    (void)hipEventRecord(e0, sb);
    hipLaunchKernelGGL(k_lat, dim3(nwg), dim3(64), 0, sb, d, s, iters);
    (void)hipEventRecord(e1, sb);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, d, sizeof(long long) * 2 * nwg, hipMemcpyDeviceToHost);
    double ghz = 0, us = 0;
    for (int b = 0; b < nwg; ++b) {
      ghz += (double)h[2 * b] / ((double)h[2 * b + 1] * 10.0);  // wall clock 100 MHz -> ns
      us += h[2 * b + 1] / 100.0;
    }
    printf("mode %d (%s): k_lat %.2f ms, mean in-kernel %.1f us, clock64 rate %.3f GHz, ticks per fma+lds step %.2f\n", mode,
           mode == 0 ? "alone" : (mode == 1 ? "with 2048-WG clear" : "with 256-WG clear"), ms, us / nwg, ghz / nwg,
           (double)h[0] / (iters * 16.0));
  }
  return 0;
}
