// How many long-running kernels on streams of their own run AT ONCE, and does a packet queued behind a running kernel (an
// event record = a barrier packet) hold the queue's pipe?  N one-workgroup kernels on N masked streams (every stream a hardware
// queue of its own) stamp their start and spin until the host sets a flag 100 ms later; with argv[2] = 1 an event is recorded
// on every stream right behind its kernel (what sogm_flight_run does for its four kernels).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/concurrent_kernels.hip -o /tmp/ck && /tmp/ck 12 1
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void k_spin(long long *start, volatile int *flag, int i) {
  if (threadIdx.x == 0) {
    start[i] = wall_clock64();
    const long long t0 = wall_clock64();
    while (!*flag && wall_clock64() - t0 < 400000000LL) __builtin_amdgcn_s_sleep(64);
  }
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 12, with_events = argc > 2 ? atoi(argv[2]) : 1, masked = argc > 3 ? atoi(argv[3]) : 1;
  long long *start;
  int       *flag;
  hipHostMalloc((void **)&start, sizeof(long long) * n, hipHostMallocDefault);
  hipHostMalloc((void **)&flag, sizeof(int), hipHostMallocDefault);
  for (int i = 0; i < n; ++i) start[i] = 0;
  *flag = 0;
  std::vector<hipStream_t> st(n);
  std::vector<hipEvent_t>  ev(n);
  uint32_t mask[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  for (int i = 0; i < n; ++i) {
    if (masked) hipExtStreamCreateWithCUMask(&st[i], 8, mask); else hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
  }
  hipDeviceSynchronize();
  for (int i = 0; i < n; ++i) {
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], start, flag, i);
    if (with_events) hipEventRecord(ev[i], st[i]);
  }
  std::this_thread::sleep_for(std::chrono::milliseconds(100));
  int running = 0;
  for (int i = 0; i < n; ++i) running += start[i] != 0;
  printf("n %d events %d masked %d: %d kernels running after 100 ms\n", n, with_events, masked, running);
  *flag = 1;
  hipDeviceSynchronize();
  long long t0 = 0;
  for (int i = 0; i < n; ++i) if (start[i] && (!t0 || start[i] < t0)) t0 = start[i];
  printf("start ms:");
  for (int i = 0; i < n; ++i) printf(" %.2f", (start[i] - t0) / 1e5);
  printf("\n");
  return 0;
}
