// Micro-check (gfx950): sum of the four 16-lane rows with v_permlane16_swap / v_permlane32_swap against two __shfl_xor butterflies.
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/micro/permlane_rows.hip -o tools/micro/permlane_rows && tools/micro/permlane_rows
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__device__ inline double rows_sum(double v) {
  // sum of the four 16-lane rows, ((r0 + r1) + (r2 + r3)), in every lane
  unsigned lo = __double2loint(v), hi = __double2hiint(v);
  u2 a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  u2 b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  double x = __hiloint2double(b.x, a.x), y = __hiloint2double(b.y, a.y);
  double s = x + y;
  lo = __double2loint(s); hi = __double2hiint(s);
  a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  x = __hiloint2double(b.x, a.x); y = __hiloint2double(b.y, a.y);
  return x + y;
}
__global__ void k(double *out) {
  const int l = threadIdx.x;
  double v = (l >> 4) == 0 ? 1.0 + l : (l >> 4) == 1 ? 1e-17 * (l & 15) : (l >> 4) == 2 ? 100.0 + l : 0.25 * (l & 15);
  out[l] = rows_sum(v);
  double a = v; a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
  out[64 + l] = a;
}
int main() {
  double *d, h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) if (h[l] != h[64 + l]) ++bad;
  printf("permlane rows_sum vs shfl butterflies: %s (lane 0: %.17g %.17g, lane 37: %.17g %.17g)\n", bad ? "MISMATCH" : "ok", h[0], h[64], h[37], h[101]);
  return bad != 0;
}
