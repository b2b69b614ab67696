// Micro-benchmark (diagnostic only): fp64 VALU dependent-chain latency vs independent-issue throughput for a
// lone wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__device__ inline long long chain(double &seed, int iters) {
  double a[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) a[c] = seed + c;
  long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) a[c] = __builtin_fma(a[c], 1.0000001, 0.5);
  }
  long long c1 = clock64();
#pragma unroll
  for (int c = 0; c < CH; ++c) seed += a[c];
  return c1 - c0;
}
template <int CH>
__device__ inline long long chain_ma(double &seed, int iters) {
  double a[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) a[c] = seed + c;
  long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        double m = a[c] * 1.0000001;
        asm volatile("" : "+v"(m));
        a[c] = m + 0.5;
      }
  }
  long long c1 = clock64();
#pragma unroll
  for (int c = 0; c < CH; ++c) seed += a[c];
  return c1 - c0;
}
__global__ __launch_bounds__(256) void k(long long *out, double *sink, int iters) {
  double seed = threadIdx.x * 1e-3;
  long long t1 = chain<1>(seed, iters), t2 = chain<2>(seed, iters), t4 = chain<4>(seed, iters), t8 = chain<8>(seed, iters);
  long long m1 = chain_ma<1>(seed, iters), m4 = chain_ma<4>(seed, iters);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = t1; out[1] = t2; out[2] = t4; out[3] = t8; out[4] = m1; out[5] = m4;
  }
  sink[blockIdx.x * 256 + threadIdx.x] = seed;
}
int main() {
  long long *d, h[8];
  double *s;
  (void)hipMalloc(&d, 64);
  (void)hipMalloc(&s, 128 * 256 * 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(128), dim3(256), 0, 0, d, s, iters);
    (void)hipDeviceSynchronize();
  }
  (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  const double n = iters * 16.0;
  printf("fma chains=1: %.2f ticks per fma-step (1 fma)\n", h[0] / n);
  printf("fma chains=2: %.2f ticks per step (2 fma)\n", h[1] / n);
  printf("fma chains=4: %.2f ticks per step (4 fma)\n", h[2] / n);
  printf("fma chains=8: %.2f ticks per step (8 fma)\n", h[3] / n);
  printf("mul+add chains=1: %.2f ticks per mul+add pair\n", h[4] / (iters * 8.0));
  printf("mul+add chains=4: %.2f ticks per 4 pairs\n", h[5] / (iters * 8.0));
  return 0;
}
