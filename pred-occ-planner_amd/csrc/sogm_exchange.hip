// sogm_exchange.hip — the per-tick trajectory exchange behind the C ABI: ONE RCCL all-gather over xGMI.
//
// Reference: every drone publishes its BezierTraj on /broadcast_traj after a successful replan
// (FiniteStateMachine::publishTrajectory, plan_manager/src/plan_manager.cpp:364-399) and stores what the others
// publish (ParticleATC::trajectoryCallback, traj_coordinator/src/particles.cpp:131-191; latest wins per drone_id).
// Here agents are sharded over the GPUs of a node (SURVEY §8 e): every rank contributes the fixed-size records of
// its agents and receives everybody's — ncclAllGather of n_local * sizeof(SogmTrajRecord) bytes per rank
// (~264 KB at 128 agents per GPU), latency-bound; no other collective exists on this path.
//
// RCCL is resolved at run time (dlopen of librccl.so.1): a process that already carries RCCL (PyTorch-ROCm, or a
// C++ host linked against it) shares that one instance, so an ncclComm_t created by the host can be passed in;
// hosts without RCCL headers create the communicator through sogm_comm_* below.  The library itself has no
// link-time dependency on RCCL and single-GPU users never load it.
//
// Ordering: the all-gather runs on the context's exchange stream.  It starts when the caller's stream has
// produced the local records (an event recorded at call time: after sogm_replan's fan-in and the caller's
// latest-wins merge) and it is consumed by whatever reads the swarm's records next — sogm_project_neighbours,
// sogm_replan's deconfliction, sogm_safe_after_opt — which wait for its completion event on THEIR stream.  The next
// tick's clear / stamp / trajectory sampling therefore overlap with the collective.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "sogm_device.hpp"

namespace {

struct RcclApi {
  void *handle;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*CommCount)(const ncclComm_t, int *);
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  const char *(*GetErrorString)(ncclResult_t);
};

RcclApi *rccl() {
  static RcclApi        api;
  static int            state = -1;  // 1 ok, -1 failed
  static std::once_flag once;        // several contexts / host threads may reach the first collective together
  std::call_once(once, [] {
    // SOGM_RCCL_LIB names the library to load instead (tests substitute an in-process stand-in for RCCL whose
    // collectives take a visible amount of time: tests/fake_rccl.cpp)
    const char *names[] = {getenv("SOGM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      if (!n || !*n) continue;
      api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.handle) break;
    }
    if (api.handle) {
      api.GetUniqueId    = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
      api.CommInitRank   = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
      api.CommDestroy    = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
      api.CommCount      = (decltype(api.CommCount))dlsym(api.handle, "ncclCommCount");
      api.CommUserRank   = (decltype(api.CommUserRank))dlsym(api.handle, "ncclCommUserRank");
      api.AllGather      = (decltype(api.AllGather))dlsym(api.handle, "ncclAllGather");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
      if (api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.CommCount && api.CommUserRank &&
          api.AllGather && api.GetErrorString)
        state = 1;
    }
  });
  return state == 1 ? &api : nullptr;
}

int rccl_fail(const char *what, ncclResult_t r) {
  RcclApi *a = rccl();
  char     buf[384];
  std::snprintf(buf, sizeof(buf), "%s: %s", what, a ? a->GetErrorString(r) : "RCCL not loaded");
  sogm::set_error_text(buf);
  return SOGM_ERR_COMM;
}

}  // namespace

namespace sogm {
int exchange_stream(sogm_ctx *ctx, hipStream_t *out) {
  if (!ctx->xstream) {
    // A stream with a compute-unit mask — here: every unit — gets a hardware queue of its OWN (plain streams share a pool of
    // GPU_MAX_HW_QUEUES queues).  The exchange stream carries kernels that WAIT (k_flight_xwait of a multi-rank flight: until a
    // tick is complete) and collectives that wait for their peers: whatever shared their queue would wait with them — with two
    // ranks in one process (tests) that was the other rank's control-block reset, and the flights stalled.
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    uint32_t mask[16];
    for (int i = 0; i < 16; ++i) mask[i] = 0xFFFFFFFFu;
    if (hipExtStreamCreateWithCUMask(&ctx->xstream, (uint32_t)((n_cu + 31) / 32 < 16 ? (n_cu + 31) / 32 : 16), mask) != hipSuccess) {
      (void)hipGetLastError();
      SOGM_HIP_CHECK(hipStreamCreateWithFlags(&ctx->xstream, hipStreamNonBlocking));
    }
    SOGM_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_xin, hipEventDisableTiming));
    SOGM_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_xdone, hipEventDisableTiming));
  }
  if (out) *out = ctx->xstream;
  return SOGM_OK;
}
int exchange_allgather_raw(sogm_ctx *ctx, void *nccl_comm, const void *send, void *recv, size_t bytes_per_rank) {
  RcclApi *a = rccl();
  if (!a) {
    sogm::set_error_text("all-gather: librccl.so.1 could not be loaded");
    return SOGM_ERR_COMM;
  }
  if (int rc = exchange_stream(ctx, nullptr)) return rc;
  const ncclResult_t r = a->AllGather(send, recv, bytes_per_rank, ncclUint8, (ncclComm_t)nccl_comm, ctx->xstream);
  if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
  return SOGM_OK;
}
int exchange_mark_pending(sogm_ctx *ctx) {
  SOGM_HIP_CHECK(hipEventRecord(ctx->ev_xdone, ctx->xstream));
  ctx->exchange_pending = 1;
  return SOGM_OK;
}
}  // namespace sogm

struct sogm_comm {
  ncclComm_t comm;
  int        rank, world, device;
};

extern "C" {

int sogm_comm_unique_id(char *out_id_host) {
  if (!out_id_host) return SOGM_ERR_INVALID_ARG;
  RcclApi *a = rccl();
  if (!a) {
    sogm::set_error_text("sogm_comm_unique_id: librccl.so.1 could not be loaded");
    return SOGM_ERR_COMM;
  }
  ncclUniqueId id;
  ncclResult_t r = a->GetUniqueId(&id);
  if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
  static_assert(sizeof(id) == SOGM_COMM_ID_BYTES, "ncclUniqueId size");
  std::memcpy(out_id_host, &id, sizeof(id));
  return SOGM_OK;
}

int sogm_comm_create(const char *id_host, int rank, int world, int device, sogm_comm **out) {
  if (!id_host || !out || world < 1 || rank < 0 || rank >= world) return SOGM_ERR_INVALID_ARG;
  *out       = nullptr;
  RcclApi *a = rccl();
  if (!a) {
    sogm::set_error_text("sogm_comm_create: librccl.so.1 could not be loaded");
    return SOGM_ERR_COMM;
  }
  if (hipSetDevice(device) != hipSuccess) return SOGM_ERR_NO_DEVICE;
  sogm_comm *c = new (std::nothrow) sogm_comm();
  if (!c) return SOGM_ERR_INVALID_ARG;
  c->rank   = rank;
  c->world  = world;
  c->device = device;
  ncclUniqueId id;
  std::memcpy(&id, id_host, sizeof(id));
  ncclResult_t r = a->CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return rccl_fail("ncclCommInitRank", r);
  }
  *out = c;
  return SOGM_OK;
}

void sogm_comm_destroy(sogm_comm *c) {
  if (!c) return;
  RcclApi *a = rccl();
  if (a && c->comm) {
    (void)hipSetDevice(c->device);
    (void)a->CommDestroy(c->comm);
  }
  delete c;
}

void *sogm_comm_handle(sogm_comm *c) { return c ? (void *)c->comm : nullptr; }

int sogm_comm_info(sogm_comm *c, int32_t *out) {
  if (!c || !out) return SOGM_ERR_INVALID_ARG;
  RcclApi *a = rccl();
  if (!a) return SOGM_ERR_COMM;
  int          n = -1, r = -1;
  ncclResult_t e = a->CommCount(c->comm, &n);
  if (e == ncclSuccess) e = a->CommUserRank(c->comm, &r);
  if (e != ncclSuccess) return rccl_fail("ncclCommCount / ncclCommUserRank", e);
  out[0] = n;
  out[1] = r;
  return SOGM_OK;
}

int sogm_traj_allgather(sogm_ctx *ctx, void *nccl_comm, const SogmTrajRecord *local_records, int n_local,
                        SogmTrajRecord *all_records, void *stream) {
  if (!ctx || !nccl_comm || !local_records || !all_records || n_local <= 0) return SOGM_ERR_INVALID_ARG;
  RcclApi *a = rccl();
  if (!a) {
    sogm::set_error_text("sogm_traj_allgather: librccl.so.1 could not be loaded");
    return SOGM_ERR_COMM;
  }
  SOGM_HIP_CHECK(hipSetDevice(ctx->device));
  if (int rc = sogm::exchange_stream(ctx, nullptr)) return rc;
  // the local records are final once everything queued on the caller's stream so far has run; every earlier
  // reader of all_records was queued on (or joined to) that stream as well
  if (ctx->records_final_valid && ctx->records_final_ptr == local_records) {
    // the records a publishing sogm_replan has just written: final — and the swarm table free of that replan's readers
    // — when its finishing kernel ends, which is before the caller's stream gets there (the pre-stamp's tail, the
    // report): the collective runs under those
    SOGM_HIP_CHECK(hipStreamWaitEvent(ctx->xstream, ctx->ev_records_final, 0));
  } else {
    SOGM_HIP_CHECK(hipEventRecord(ctx->ev_xin, (hipStream_t)stream));
    SOGM_HIP_CHECK(hipStreamWaitEvent(ctx->xstream, ctx->ev_xin, 0));
  }
  ctx->records_final_valid = 0;
  sogm::prof_begin(ctx, SOGM_PROF_EXCHANGE, ctx->xstream);
  ncclResult_t r = a->AllGather(local_records, all_records, (size_t)n_local * sizeof(SogmTrajRecord), ncclUint8,
                                (ncclComm_t)nccl_comm, ctx->xstream);
  sogm::prof_end(ctx, SOGM_PROF_EXCHANGE, ctx->xstream);
  // (recorded even when the collective failed: consumers stay ordered behind the producers on the exchange stream
  //  instead of behind the previous collective's event)
  SOGM_HIP_CHECK(hipEventRecord(ctx->ev_xdone, ctx->xstream));
  ctx->exchange_pending = 1;
  if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
  return SOGM_OK;
}

int sogm_exchange_wait(sogm_ctx *ctx, void *stream) {
  if (!ctx) return SOGM_ERR_INVALID_ARG;
  return sogm::join_exchange(ctx, (hipStream_t)stream);
}

}  // extern "C"
