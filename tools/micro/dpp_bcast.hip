// Micro-check (gfx950): v_fmac_f64_dpp with row_newbcast:k reads lane k of each 16-lane row for src0.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/dpp_bcast.hip -o tools/micro/dpp_bcast && tools/micro/dpp_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out) {
  const int l = threadIdx.x;
  double    r = 100.0 * (l >> 4) + (l & 15);  // lane (row q, i) holds 100 q + i
  double    one = 1.0, acc5 = 0.0, acc11 = 0.0;
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc5) : "v"(r), "v"(one));
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xf" : "+v"(acc11) : "v"(r), "v"(one));
  out[l]      = acc5;
  out[64 + l] = acc11;
}
int main() {
  double *d, h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    if (h[l] != 100.0 * (l >> 4) + 5) ++bad;
    if (h[64 + l] != 100.0 * (l >> 4) + 11) ++bad;
  }
  printf("row_newbcast check: %s (lane 0: %g %g, lane 37: %g %g)\n", bad ? "MISMATCH" : "ok", h[0], h[64], h[37], h[64 + 37]);
  return bad != 0;
}
