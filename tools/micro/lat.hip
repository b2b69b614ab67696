// Micro-benchmark (diagnostic only): shader clock, barrier cost, dependent LDS / fp64 FMA latency for a lone
// wave per SIMD (the k_qp situation: 256-thread workgroup, one workgroup per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(long long *out, int iters) {
  __shared__ double s[512];
  __shared__ int    idx[512];
  const int tid = threadIdx.x;
  for (int i = tid; i < 512; i += 256) {
    s[i]   = 1.0 + i;
    idx[i] = (i * 7 + 3) & 511;
  }
  __syncthreads();
  long long w0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < iters; ++i) __syncthreads();
  long long c1 = clock64();
  // dependent LDS chain
  int p = tid;
  for (int i = 0; i < iters; ++i) p = idx[p];
  long long c2 = clock64();
  // dependent fp64 fma chain
  double a = s[tid];
  for (int i = 0; i < iters; ++i) a = __builtin_fma(a, 1.0000001, 0.5);
  long long c3 = clock64();
  // 8 independent LDS reads + wait
  double acc = 0;
  for (int i = 0; i < iters; ++i) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = s[(p + 37 * u + i) & 511];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  long long c4 = clock64(), w1 = wall_clock64();
  if (tid == 0 && blockIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = c2 - c1;
    out[2] = c3 - c2;
    out[3] = c4 - c3;
    out[4] = w1 - w0;
    out[5] = c4 - c0;
    out[6] = p + (long long)a + (long long)acc;
  }
}
int main() {
  long long *d, h[8];
  hipMalloc(&d, 64);
  const int iters = 20000;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k, dim3(128), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  int wc = 0;
  hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
  printf("wall clock rate kHz %d\n", wc);
  double ghz = (double)h[5] / ((double)h[4] / (wc * 1e3)) / 1e9;
  printf("clock64 ticks per s (GHz) %.3f\n", ghz);
  printf("barrier            %.1f ticks\n", (double)h[0] / iters);
  printf("dependent LDS read %.1f ticks\n", (double)h[1] / iters);
  printf("dependent fp64 fma %.1f ticks\n", (double)h[2] / iters);
  printf("8 indep LDS + 8 add %.1f ticks\n", (double)h[3] / iters);
  return 0;
}
