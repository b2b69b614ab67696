// fake_rccl.cpp — an in-process stand-in for librccl (TEST INFRASTRUCTURE; loaded through SOGM_RCCL_LIB by
// tests/test_exchange_gpu.py): N "ranks" are N host threads of ONE process sharing one GPU, a communicator group is
// found by its unique id, and ncclAllGather really moves the bytes between the ranks' buffers, on each caller's
// stream, behind a kernel that spins for a millisecond — so a consumer of sogm_traj_allgather that is not ordered
// behind the collective's completion event, or a collective that does not wait for its producer, reads stale data.
// Only the seven entry points csrc/sogm_exchange.hip resolves are provided.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Group {
  std::mutex              mu;
  std::condition_variable cv;
  int                     nranks = 0;
  long long               arrived = 0, finished = 0;  // cumulative over collectives
  std::vector<const void *> send;
  std::vector<hipEvent_t>   posted, done;
};
struct Comm {
  Group    *g;
  int       rank;
  long long seq = 0;
};
std::mutex                     g_mu;
std::map<std::string, Group *> g_groups;

__global__ void k_spin(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  static int n = 0;
  std::memset(id, 0, sizeof(*id));
  std::snprintf(id->internal, sizeof(id->internal), "fake-rccl-%d", ++n);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
  std::lock_guard<std::mutex> lk(g_mu);
  Group *&g = g_groups[std::string(id.internal)];
  if (!g) {
    g         = new Group();
    g->nranks = nranks;
    g->send.resize(nranks);
    g->posted.resize(nranks);
    g->done.resize(nranks);
    for (int r = 0; r < nranks; ++r) {
      (void)hipEventCreateWithFlags(&g->posted[r], hipEventDisableTiming);
      (void)hipEventCreateWithFlags(&g->done[r], hipEventDisableTiming);
    }
  }
  if (g->nranks != nranks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm *c = new Comm();
  c->g    = g;
  c->rank = rank;
  *out    = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  delete reinterpret_cast<Comm *>(comm);
  return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t comm, int *n) {
  *n = reinterpret_cast<const Comm *>(comm)->g->nranks;
  return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *r) {
  *r = reinterpret_cast<const Comm *>(comm)->rank;
  return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake rccl error"; }

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t, ncclComm_t comm,
                           hipStream_t stream) {
  Comm  *c = reinterpret_cast<Comm *>(comm);
  Group *g = c->g;
  const int       n   = g->nranks;
  const long long seq = ++c->seq;
  {  // rendezvous 1: every rank has posted its send buffer (ready when its stream reaches `posted`)
    std::unique_lock<std::mutex> lk(g->mu);
    g->send[c->rank] = sendbuff;
    if (hipEventRecord(g->posted[c->rank], stream) != hipSuccess) return ncclUnhandledCudaError;
    ++g->arrived;
    g->cv.notify_all();
    g->cv.wait(lk, [&] { return g->arrived >= seq * n; });
  }
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, stream, 100000LL);  // 1 ms of the 100 MHz clock: "the wire"
  for (int p = 0; p < n; ++p) {
    if (hipStreamWaitEvent(stream, g->posted[p], 0) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(static_cast<char *>(recvbuff) + (size_t)p * count, g->send[p], count, hipMemcpyDeviceToDevice,
                       stream) != hipSuccess)
      return ncclUnhandledCudaError;
  }
  {  // rendezvous 2: nobody's send buffer is reused (next collective, next producer) before every rank has read it
    std::unique_lock<std::mutex> lk(g->mu);
    if (hipEventRecord(g->done[c->rank], stream) != hipSuccess) return ncclUnhandledCudaError;
    ++g->finished;
    g->cv.notify_all();
    g->cv.wait(lk, [&] { return g->finished >= seq * n; });
  }
  for (int p = 0; p < n; ++p)
    if (hipStreamWaitEvent(stream, g->done[p], 0) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

}  // extern "C"
