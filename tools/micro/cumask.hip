// cumask.hip — what hipExtStreamCreateWithCUMask does on this device: which (XCC, SE, CU) the workgroups of a masked stream
// land on, for the 16-CU "units" sogm_flight_run builds its partitions from (unit(i) = ((i / 8) % 4) * 4 + ((i / 32 + i % 8) % 4):
// two CUs per XCD whether mask bit i means XCD i / 32 or XCD i % 8).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/cumask.hip -o tools/micro/cumask && tools/micro/cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_where(unsigned *out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xF) << 16 | (hw & 0xFFFF);
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
}
static int unit_of(int i) { return ((i / 8) % 4) * 4 + ((i / 32 + i % 8) % 4); }
int main() {
  int n_cu = 0;
  CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
  printf("CUs %d\n", n_cu);
  unsigned *d;
  const int N = 4096;
  CK(hipMalloc(&d, sizeof(unsigned) * N));
  std::vector<unsigned> h(N);
  auto run = [&](const char *name, const std::vector<uint32_t> &mask, int lds) -> int {
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    CK(hipMemsetAsync(d, 0xFF, sizeof(unsigned) * N, st));
    hipLaunchKernelGGL(k_where, dim3(N), dim3(64), lds, st, d, 2000);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d, sizeof(unsigned) * N, hipMemcpyDeviceToHost));
    std::set<unsigned> cus;
    int per_xcc[16] = {0};
    for (unsigned v : h) {
      const unsigned xcc = v >> 16, se = (v >> 13) & 7, sh = (v >> 12) & 1, cu = (v >> 8) & 15;
      const unsigned id = xcc << 12 | se << 8 | sh << 4 | cu;
      if (cus.insert(id).second) per_xcc[xcc & 15]++;
    }
    int bits = 0;
    for (uint32_t w : mask) bits += __builtin_popcount(w);
    printf("%-28s mask bits %3d -> distinct CUs %3zu, per XCC:", name, bits, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
    CK(hipStreamDestroy(st));
    return 0;
  };
  const int words = (n_cu + 31) / 32;
  std::vector<uint32_t> all(words, 0xFFFFFFFFu), lo64(words, 0), units4(words, 0), unit0(words, 0), odd(words, 0);
  for (int i = 0; i < 64; ++i) lo64[i / 32] |= 1u << (i % 32);
  for (int i = 0; i < n_cu; ++i) {
    if (unit_of(i) < 4) units4[i / 32] |= 1u << (i % 32);
    if (unit_of(i) == 0) unit0[i / 32] |= 1u << (i % 32);
    if (unit_of(i) >= 10) odd[i / 32] |= 1u << (i % 32);
  }
  if (run("all", all, 0)) return 1;
  if (run("bits 0..63", lo64, 0)) return 1;
  if (run("units 0..3 (64 CUs)", units4, 0)) return 1;
  if (run("unit 0 (16 CUs)", unit0, 0)) return 1;
  if (run("units 10..15 (96 CUs)", odd, 0)) return 1;
  if (run("unit 0, 120 KB LDS", unit0, 120 * 1024)) return 1;
  return 0;
}
