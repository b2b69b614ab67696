// Fills the LDS of every compute unit with a 32-bit pattern and exits: LDS is not cleared between kernels or processes, so a
// kernel that reads a word of its LDS before writing it sees whatever the previous workgroup on that CU left there —
// usually a previous launch of the same kernel (benign values), after this program the pattern.
// hipcc --offload-arch=gfx950 -O2 tools/micro/lds_poison.hip -o /tmp/lds_poison ; /tmp/lds_poison 0x7ff80000
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_poison(unsigned pattern, int words, unsigned *sink) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = pattern + (pattern == 0xdeadbeefu ? (unsigned)i : 0u);
  __syncthreads();
  // keep the stores alive and the workgroup resident for a moment so that the launch spreads over all CUs
  unsigned acc = 0;
  for (int i = threadIdx.x; i < words; i += blockDim.x) acc ^= lds[i];
  for (int k = 0; k < 200; ++k) __builtin_amdgcn_s_sleep(127);
  if (acc == 0x12345678u) sink[0] = acc;
}
int main(int argc, char **argv) {
  const unsigned pattern = argc > 1 ? (unsigned)strtoul(argv[1], nullptr, 0) : 0xffffffffu;
  const int      bytes   = 160 * 1024;
  unsigned      *sink    = nullptr;
  if (hipMalloc((void **)&sink, 4) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void *)k_poison, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 2;
  hipLaunchKernelGGL(k_poison, dim3(2048), dim3(256), bytes, 0, pattern, bytes / 4, sink);
  if (hipDeviceSynchronize() != hipSuccess) return 3;
  std::printf("LDS poisoned with 0x%08x\n", pattern);
  return 0;
}
