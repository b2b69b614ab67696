// sogm_planner.hip — planner context and host side of the planner entry points: the per-stage calls (search,
// corridors, QP, deconfliction) and sogm_replan in its two forms — the dataflow replan (replan_flow: five launches,
// persistent corridor / QP / finish kernels chained per agent through ready lists) and the grouped-stream chain
// (replan_impl).  SOGM_ERR_STATE from an entry point means a call-order error (no map update before planning).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "sogm_planner.hpp"

using namespace sogm;

// Per-piece min-jerk cost  QM = p2j^T [[I/3, I/6],[I/6, I/3]] p2j  with p2j = a2j v2a p2v
// (traj_opt/src/bezier_optimizer.cpp:64-111; N = 4, DIM = 3).  Host-side, once per planner.
static void min_jerk_block(double *QM /*15x15*/) {
  // derivative-of-control-points operators act per dimension: rows (i,d), cols (j,d)
  auto diff = [](int rows, double scale, double *out, int cols) {
    for (int i = 0; i < rows * cols; ++i) out[i] = 0.0;
    for (int i = 0; i < rows / 3; ++i)
      for (int d = 0; d < 3; ++d) {
        out[(i * 3 + d) * cols + i * 3 + d]       = -scale;
        out[(i * 3 + d) * cols + (i + 1) * 3 + d] = scale;
      }
  };
  double p2v[12 * 15], v2a[9 * 12], a2j[6 * 9], p2a[9 * 15], p2j[6 * 15], W[6 * 6], T[6 * 15];
  diff(12, 4, p2v, 15);
  diff(9, 3, v2a, 12);
  diff(6, 2, a2j, 9);
  auto mul = [](const double *A, const double *B, double *C, int r, int k, int c) {
    for (int i = 0; i < r; ++i)
      for (int j = 0; j < c; ++j) {
        double s = 0;
        for (int q = 0; q < k; ++q) s += A[i * k + q] * B[q * c + j];
        C[i * c + j] = s;
      }
  };
  mul(v2a, p2v, p2a, 9, 12, 15);
  mul(a2j, p2a, p2j, 6, 9, 15);
  for (int i = 0; i < 36; ++i) W[i] = 0.0;
  for (int d = 0; d < 3; ++d) {
    W[d * 6 + d]           = 1.0 / 3;
    W[d * 6 + 3 + d]       = 1.0 / 6;
    W[(3 + d) * 6 + d]     = 1.0 / 6;
    W[(3 + d) * 6 + 3 + d] = 1.0 / 3;
  }
  mul(W, p2j, T, 6, 6, 15);
  for (int i = 0; i < 15; ++i)
    for (int j = 0; j < 15; ++j) {
      double s = 0;
      for (int q = 0; q < 6; ++q) s += p2j[q * 15 + i] * T[q * 15 + j];
      QM[i * 15 + j] = s;
    }
}

// BezierTraj record of a successful replan (plan_manager/src/plan_manager.cpp:364-399 publishes
// duration[] and cpts[]); n_pieces = 0 marks "replan() returned false".
__global__ void k_pack_records(int A, double corridor_tau, const int32_t *ret, const int32_t *npoly,
                               const int32_t *status, const int32_t *safe, const double *cpts,
                               const double *t_start, const int32_t *drone_ids, SogmTrajRecord *out,
                               int32_t *ok, int agent0, unsigned long long *counters, SogmTrajRecord *pub_own,
                               SogmTrajRecord *pub_table) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x + agent0;
  if (a >= A) return;
  if (counters) {  // where this replan ended (baseline_fake.cpp: :292 no path, :405-419 corridors, :447 QP, :455 unsafe)
    int k = SOGM_CNT_REPLAN_OK;
    if (ret[a] == 0) k = SOGM_CNT_FAIL_SEARCH;
    else if (npoly[a] <= 0) k = SOGM_CNT_FAIL_CORRIDOR;
    else if (!(status[a] == 1 || status[a] == 2)) k = SOGM_CNT_FAIL_QP;
    else if (safe != nullptr && safe[a] == 0) k = SOGM_CNT_FAIL_UNSAFE;
    atomicAdd(&counters[k], 1ull);
  }
  SogmTrajRecord &r = out[a];
  // isSafeAfterOpt false -> replan() returns false (baseline_fake.cpp:455-460)
  const bool good = ret[a] != 0 && npoly[a] > 0 && (status[a] == 1 || status[a] == 2) &&
                    (safe == nullptr || safe[a] != 0);
  r.drone_id        = drone_ids[a];
  r.time_start      = t_start[a];
  r.n_pieces        = good ? npoly[a] : 0;
  for (int i = 0; i < SOGM_MAX_PIECES; ++i) r.duration[i] = (good && i < npoly[a]) ? corridor_tau : 0.0;
  for (int i = 0; i < SOGM_MAX_PIECES * 15; ++i)
    r.cpts[i] = (good && i < npoly[a] * 15) ? cpts[(size_t)a * SOGM_MAX_PIECES * 15 + i] : 0.0;
  ok[a] = good ? 1 : 0;
  if (pub_own) {  // publication, as k_finish_flow does it (sogm_planner_set_publish)
    if (good) pub_own[a] = r;
    if (pub_table) pub_table[a] = pub_own[a];
  }
}

// End of a dataflow replan (one lane, on the caller's stream after the fan-in): a wait of this tick timed out ->
// remember the code and count the tick in pinned host memory, so the host can see failed ticks without a device
// synchronisation (sogm_planner_flow_failures).
__global__ void k_flow_report(const int *__restrict__ hdr, int *__restrict__ host_words, int *__restrict__ epoch_word,
                              long long *__restrict__ tick_clock) {
  if (tick_clock) tick_clock[1] = wall_clock64();  // the tick's last kernel (sogm_tick_clock)
  // the replan is over: the wide launch of the side-stream clear retires (sogm_ctx::clear_epoch_word), so that the
  // one-wave glue kernels between two replans (latest-wins merge, tick inputs, the stamp) find the memory pipeline
  // responsive; the narrow launch goes on
  if (epoch_word) *epoch_word = 0;
  const int e = hdr[FLOW_ERR];
  if (e != 0) {
    host_words[0] = e;
    host_words[1] = host_words[1] + 1;
  }
}

extern "C" {

int sogm_planner_create(sogm_ctx *map, const SogmAstarParams *astar, const SogmPlannerParams *pp,
                        const SogmQpSettings *qp, sogm_planner **out) {
  if (!map || !astar || !pp || !qp || !out) return SOGM_ERR_INVALID_ARG;
  sogm_planner *p = new (std::nothrow) sogm_planner();
  if (!p) return SOGM_ERR_INVALID_ARG;
  std::memset(p, 0, sizeof(*p));
  p->map = map;
  p->ap  = *astar;
  p->pp  = *pp;
  // BaselinePlanner searches with RiskHybridAstar, FakeBaselinePlanner with FakeRiskHybridAstar: the classes
  // differ in the shot check only (risk_hybrid_a_star.cpp:514 vs fake_risk_hybrid_a_star.cpp:521)
  p->ap.shot_ignores_time = pp->fake_planner ? 0 : 1;
  p->qs  = *qp;
  // The QP's register-resident path assumes every velocity / acceleration row is two-sided (rho = rho_cur or 1000 rho_cur):
  // "unbounded" limits (OSQP_INFTY-like values) would give those rows RHO_MIN and a K that does not match
  if (!(pp->opt_max_vel > 0 && pp->opt_max_vel < 1e20) || !(pp->opt_max_acc > 0 && pp->opt_max_acc < 1e20)) {
    delete p;
    return SOGM_ERR_INVALID_ARG;
  }
  if (astar->allocate_num < 2 || astar->allocate_num > astar_pool_max() || astar->check_num < 1 || !(astar->resolution > 0) ||
      !(astar->time_resolution > 0)) {
    delete p;
    return SOGM_ERR_INVALID_ARG;
  }
  const int A = map->n_agents;
  int       hc = 1;
  while (hc < 2 * astar->allocate_num) hc <<= 1;
  p->aw.hash_cap    = hc;
  p->aw.pool_stride = astar_node_bytes() * (size_t)astar->allocate_num;
  p->route_cap      = 64;
  // two pools / hash tables per agent: the replan's second search attempt runs speculatively beside the first
  hipError_t e      = hipMalloc((void **)&p->aw.pool, p->aw.pool_stride * A * 2);
  if (e == hipSuccess) e = hipMalloc(&p->aw.hkeys, 8 * (size_t)hc * A * 2);
  if (e == hipSuccess) e = hipMalloc((void **)&p->aw.verdict, sizeof(int) * A);
  if (e == hipSuccess) e = hipMalloc((void **)&p->aw.dbg, sizeof(long long) * 8 * A);
  if (e == hipSuccess) e = hipMemset(p->aw.dbg, 0, sizeof(long long) * 8 * A);
  if (e == hipSuccess) e = hipMalloc((void **)&p->d_ret, sizeof(int32_t) * A);
  if (e == hipSuccess) e = hipMalloc((void **)&p->d_route_len, sizeof(int32_t) * A);
  if (e == hipSuccess) e = hipMalloc((void **)&p->d_stats, sizeof(int32_t) * 4 * A);
  if (e == hipSuccess) e = hipMalloc((void **)&p->d_route, sizeof(double) * 6 * p->route_cap * A);
  {
    if (pp->max_faces < 6 || pp->max_faces > 64 || pp->pc_capacity < 1 || pp->pc_capacity > 16384 ||
        pp->firi_iterations < 1) {
      sogm_planner_destroy(p);
      return SOGM_ERR_INVALID_ARG;
    }
    const size_t slots = (size_t)A * SOGM_MAX_PIECES, cap = (size_t)pp->pc_capacity;
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.pc, sizeof(double) * slots * cap * 3);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.fpc, sizeof(double) * slots * cap * 3);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.tang, sizeof(double) * slots * cap * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.distr, sizeof(double) * slots * cap);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.polys, sizeof(double) * slots * pp->max_faces * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.seg_nfaces, sizeof(int32_t) * slots);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.seg_state, sizeof(int32_t) * slots);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.seg_npts, sizeof(int32_t) * slots);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.seg_dbg, sizeof(long long) * 16 * slots);
    // (slots no segment ever ran in are read by sogm_debug_corridor_stats as well: zero, not whatever the allocation held)
    if (e == hipSuccess) e = hipMemset(p->cw.seg_dbg, 0, sizeof(long long) * 16 * slots);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.counters, sizeof(unsigned long long) * SOGM_CNT_N);
    if (e == hipSuccess) e = hipMemset(p->cw.counters, 0, sizeof(unsigned long long) * SOGM_CNT_N);
    // QP row storage fallback (rows normally live in LDS)
    p->qw.scratch_stride = qp_scratch_bytes_per_agent(pp->max_faces);
    p->qw.dyn_lds_bytes  = qp_dynamic_lds_bytes();  // 160 KiB/CU minus k_qp's static LDS
    if (e == hipSuccess) e = hipMalloc((void **)&p->qw.scratch, p->qw.scratch_stride * (size_t)A);
    if (e == hipSuccess) e = hipMalloc((void **)&p->qw.k1_scratch, qp_k1_scratch_bytes_per_agent() * (size_t)A);
    if (e == hipSuccess) e = hipMalloc((void **)&p->qw.dbg, sizeof(long long) * 16 * (size_t)A);
    if (e == hipSuccess) e = hipMemset(p->qw.dbg, 0, sizeof(long long) * 16 * (size_t)A);
    min_jerk_block(p->qc.QM);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_polys, sizeof(double) * slots * pp->max_faces * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_goal, sizeof(double) * 6 * A);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_cpts, sizeof(double) * slots * 15);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_nfaces, sizeof(int32_t) * slots);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_npoly, sizeof(int32_t) * A);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_status, sizeof(int32_t) * A);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_iters, sizeof(int32_t) * A);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_safe, sizeof(int32_t) * A);
  }
  // Streams beyond the number of hardware queues (ROCm default GPU_MAX_HW_QUEUES = 4) share a queue
  // and serialise, so the default is 2 groups; the Python driver raises both (GPU_MAX_HW_QUEUES = 32 before HIP
  // initialises, sogm_set_tuning "groups" = 8 before it creates the planner).
  {
    int ng = map->tune_i(SOGM_TUNE_GROUPS);
    if (ng < 1) ng = 1;
    if (ng > SOGM_MAX_GROUPS) ng = SOGM_MAX_GROUPS;
    p->n_groups = A < ng ? A : ng;
  }
  p->sel_first   = 0;
  p->sel_count   = A;
  p->search_mode = 0;
  {
    p->spec_astar = map->tune_i(SOGM_TUNE_SPEC_ASTAR) != 0;
    // the second attempts wait on the first ones: only when every workgroup of both is resident at once
    if (p->spec_astar && 2 * A > sogm::astar_resident_workgroups(map->device)) p->spec_astar = 0;
  }
  // (the group streams themselves are created by the grouped path's first replan: the dataflow path never uses them,
  // and every stream is a hardware queue the process holds — INTEGRATION.md section 2 on sharing a GPU)
  for (int g = 0; g < p->n_groups && e == hipSuccess; ++g) {
    e = hipEventCreateWithFlags(&p->ev_corr[g], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_pts[g], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_done[g], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_in, hipEventDisableTiming);
  // dataflow replan: control block + four streams (needs >= 5 hardware queues to overlap with the clear)
  {
    const char *ef = getenv("SOGM_FLOW");  // (one of the library's three environment switches)
    p->flow        = ef ? atoi(ef) != 0 : 1;
    // layout: header, seg_done[A], stage[A] (zeroed per replan), then the four ready lists (-1 per replan)
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_flow, sizeof(int) * (FLOW_HDR + 6 * (size_t)A + 1));  // (+ the reset's generation word)
    if (e == hipSuccess) e = hipMemset(p->d_flow, 0, sizeof(int) * (FLOW_HDR + 6 * (size_t)A + 1));
    p->fc.hdr      = p->d_flow;
    p->fc.seg_done = p->d_flow + FLOW_HDR;
    p->fc.stage    = p->fc.seg_done + A;
    p->fc.a_ready  = p->fc.stage + A;
    p->fc.q_ready  = p->fc.a_ready + A;
    p->fc.f_ready  = p->fc.q_ready + A;
    p->fc.p_ready  = p->fc.f_ready + A;
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_pdone, hipEventDisableTiming);
    // [A][8] chain stamps, then [A][4] pre-stamp stamps (record seen, cull done, bits done, marks done)
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_flow_ts, sizeof(long long) * 12 * (size_t)A);
    if (e == hipSuccess) e = hipMemset(p->d_flow_ts, 0, sizeof(long long) * 12 * (size_t)A);
    p->fc.ts = p->d_flow_ts;

    for (int k = 0; k < 4 && e == hipSuccess; ++k) {
      e = sogm::create_stream_partitioned(&p->fstream[k], k == 2 ? 2 : 1);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_fdone[k], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_gate, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_epoch, sizeof(int));
    if (e == hipSuccess) e = hipMemset(p->d_epoch, 0, sizeof(int));
    if (e == hipSuccess) e = hipHostMalloc((void **)&p->h_flow_fail, sizeof(int) * 2, hipHostMallocMapped);
    if (e == hipSuccess) p->h_flow_fail[0] = p->h_flow_fail[1] = 0;
  }
  // (the memsets above are null-stream operations: complete before any kernel on the planner's non-blocking streams)
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  if (e != hipSuccess) {
    sogm::set_error("sogm_planner_create", e);
    sogm_planner_destroy(p);
    return SOGM_ERR_HIP;
  }
  if (p->flow) {  // the map's pooled clear follows this planner's "corridors final" counter (sogm_device.hpp)
    map->clear_gate        = p->fc.hdr + FLOW_Q_READY_N;
    map->clear_gate_err    = p->fc.hdr + FLOW_ERR;
    map->clear_gate_target = A;
    {  // tuning aid: open the wide clear at a fraction of A
      const double f = map->tune[SOGM_TUNE_CLEAR_GATE_FRAC];
      if (f > 0.0 && f < 1.0) map->clear_gate_target = (int)(f * A + 0.5) < 1 ? 1 : (int)(f * A + 0.5);
    }
    map->clear_epoch_word  = p->d_epoch;
  }
  *out = p;
  return SOGM_OK;
}
void sogm_planner_destroy(sogm_planner *p) {
  if (!p) return;
  // (the planner's map must still exist: sogm_abi.h "destroy planners before their sogm_ctx")
  if (p->map) (void)hipSetDevice(p->map->device);
  if (p->map && p->map->ps_fail_host == p->h_flow_fail) p->map->ps_fail_host = nullptr;
  if (p->map && p->map->pdone_pending && p->map->ev_pdone == p->ev_pdone) {
    (void)hipDeviceSynchronize();  // a pre-stamp nobody joined: its event and progress words go away with this planner
    p->map->pdone_pending = 0;
    p->map->ps_stage = p->map->ps_err = nullptr;
  }
  if (p->map && p->d_epoch && p->map->clear_epoch_word == p->d_epoch) {
    (void)hipDeviceSynchronize();  // a gate kernel may still be polling this planner's words
    p->map->clear_gate = p->map->clear_gate_err = p->map->clear_epoch_word = nullptr;
  }
  void *ptrs[] = {p->aw.pool, p->aw.hkeys, p->aw.dbg, p->aw.verdict,
                  p->d_ret,   p->d_route_len, p->d_stats, p->d_route,
                  p->cw.pc,   p->cw.fpc,  p->cw.tang, p->cw.distr, p->cw.polys,
                  p->cw.seg_nfaces, p->cw.seg_state, p->cw.seg_npts, p->cw.seg_dbg, p->cw.counters,
                  p->qw.scratch, p->qw.dbg, p->qw.k1_scratch,
                  p->d_polys, p->d_goal, p->d_cpts, p->d_nfaces, p->d_npoly, p->d_status, p->d_iters,
                  p->d_safe,  p->d_flow, p->d_flow_ts, p->d_epoch};
  for (void *q : ptrs)
    if (q) (void)hipFree(q);
  for (int g = 0; g < SOGM_MAX_GROUPS; ++g) {
    if (p->gstream[g]) {
      (void)hipStreamSynchronize(p->gstream[g]);
      (void)hipStreamDestroy(p->gstream[g]);
    }
    if (p->ev_corr[g]) (void)hipEventDestroy(p->ev_corr[g]);
    if (p->ev_pts[g]) (void)hipEventDestroy(p->ev_pts[g]);
    if (p->ev_done[g]) (void)hipEventDestroy(p->ev_done[g]);
  }
  if (p->ev_in) (void)hipEventDestroy(p->ev_in);
  for (int k = 0; k < 4; ++k) {
    if (p->fstream[k]) {
      (void)hipStreamSynchronize(p->fstream[k]);
      (void)hipStreamDestroy(p->fstream[k]);
    }
    if (p->ev_fdone[k]) (void)hipEventDestroy(p->ev_fdone[k]);
  }
  if (p->peek) (void)hipStreamDestroy(p->peek);
  for (int k = 0; k < 4; ++k) {
    if (p->fl_stream[k]) {
      (void)hipStreamSynchronize(p->fl_stream[k]);
      (void)hipStreamDestroy(p->fl_stream[k]);
    }
    if (p->fl_ev_done[k]) (void)hipEventDestroy(p->fl_ev_done[k]);
  }
  if (p->fl_ev_in) (void)hipEventDestroy(p->fl_ev_in);
  if (p->h_fl_worlds) (void)hipHostFree(p->h_fl_worlds);
  {
    void *fp[] = {p->d_fl, p->d_fl_worlds, p->d_fl_pva, p->d_fl_tstart, p->d_fl_now, p->fl.ts, p->fl.acc, p->fl.prof, p->fl.ts_log};
    for (void *q : fp)
      if (q) (void)hipFree(q);
  }
  if (p->ev_gate) (void)hipEventDestroy(p->ev_gate);
  if (p->ev_pdone) (void)hipEventDestroy(p->ev_pdone);
  if (p->h_flow_fail) (void)hipHostFree(p->h_flow_fail);
  delete p;
}

// The search's per-phase clock statistics (sogm_debug_astar_stats, tools/diag_astar.py) are collected while the map's
// profiling is on: five s_memrealtime reads per expansion are not free.
static sogm::AstarWorkspace astar_ws(const sogm_planner *p) {
  sogm::AstarWorkspace w = p->aw;
  if (!((p->map->profiling >> SOGM_PROF_ASTAR) & 1)) w.dbg = nullptr;
  return w;
}

int sogm_astar_search(sogm_planner *p, const double *start_pva, const double *goal,
                      const double *t_start, int32_t *out_ret, double *out_route,
                      int32_t *out_route_len, int route_cap, int32_t *out_stats,
                      int32_t *out_trace, int trace_cap, void *stream) {
  if (!p || !start_pva || !goal || !t_start || !out_ret || !out_route || !out_route_len ||
      !out_stats || route_cap < 2)
    return SOGM_ERR_INVALID_ARG;
  if (!p->map->updated) return SOGM_ERR_STATE;
  SOGM_HIP_CHECK(hipSetDevice(p->map->device));
  hipStream_t st = (hipStream_t)stream;
  if (int rc0 = sogm::join_update(p->map, st)) return rc0;
  prof_begin(p->map, SOGM_PROF_ASTAR, st);
  int rc = launch_astar(view_of(p->map), p->ap, p->pp.corridor_tau, astar_ws(p), p->sel_count,
                        start_pva, goal, t_start, out_ret, out_route, out_route_len, route_cap,
                        out_stats, out_trace, out_trace ? trace_cap : 0, st, p->sel_first, nullptr, p->search_mode);
  prof_end(p->map, SOGM_PROF_ASTAR, st);
  if (rc) {
    sogm::set_error("k_astar", hipGetLastError());
    return SOGM_ERR_HIP;
  }
  return SOGM_OK;
}
int sogm_corridor_generate(sogm_planner *p, const double *start_pva, const double *t_start,
                           const double *route, const int32_t *route_len, int route_cap,
                           double *out_polys, int32_t *out_nfaces, int32_t *out_npoly,
                           double *out_goal, void *stream) {
  if (!p || !start_pva || !t_start || !route || !route_len || !out_polys || !out_nfaces ||
      !out_npoly || !out_goal || route_cap < 2)
    return SOGM_ERR_INVALID_ARG;
  if (!p->map->updated) return SOGM_ERR_STATE;
  SOGM_HIP_CHECK(hipSetDevice(p->map->device));
  hipStream_t st = (hipStream_t)stream;
  if (int rc0 = sogm::join_update(p->map, st)) return rc0;
  prof_begin(p->map, SOGM_PROF_CORRIDOR, st);
  int rc = launch_corridor(view_of(p->map), p->pp, p->cw, p->sel_count, start_pva, t_start,
                           route, route_len, route_cap, out_polys, out_nfaces, out_npoly, out_goal,
                           st, p->sel_first);
  prof_end(p->map, SOGM_PROF_CORRIDOR, st);
  if (rc) {
    sogm::set_error("k_corridor", hipGetLastError());
    return SOGM_ERR_HIP;
  }
  return SOGM_OK;
}
// diagnostics (tools/ only): per-agent A* phase ticks
int sogm_debug_astar_stats(sogm_planner *p, long long *out_host) {
  if (!p || !out_host) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  SOGM_HIP_CHECK(hipMemcpy(out_host, p->aw.dbg, sizeof(long long) * 8 * (size_t)p->map->n_agents,
                           hipMemcpyDeviceToHost));
  return SOGM_OK;
}

// diagnostics (tools/ only): raw 128-byte node records of one agent's last search through sogm_astar_search (the
// replan's speculative second attempt keeps its nodes in the second half of the pool)
int sogm_debug_astar_nodes(sogm_planner *p, int agent, void *out_host, int n) {
  if (!p || !out_host || agent < 0 || agent >= p->map->n_agents || n < 0) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  SOGM_HIP_CHECK(hipMemcpy(out_host, p->aw.pool + (size_t)agent * p->aw.pool_stride, (size_t)n * sogm::astar_node_bytes(),
                           hipMemcpyDeviceToHost));
  return SOGM_OK;
}

// diagnostics (tools/ only): per-agent clock split of the last QP solve (QpWorkspace::dbg), [A][16]
int sogm_debug_qp_stats(sogm_planner *p, long long *out_host) {
  if (!p || !out_host) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  SOGM_HIP_CHECK(hipMemcpy(out_host, p->qw.dbg, sizeof(long long) * 16 * (size_t)p->map->n_agents, hipMemcpyDeviceToHost));
  return SOGM_OK;
}

// diagnostics (tools/ only): copy the per-segment corridor counters to the host
int sogm_debug_corridor_stats(sogm_planner *p, long long *out_host) {
  if (!p || !out_host) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  SOGM_HIP_CHECK(hipMemcpy(out_host, p->cw.seg_dbg,
                           sizeof(long long) * 16 * (size_t)p->map->n_agents * SOGM_MAX_PIECES,
                           hipMemcpyDeviceToHost));
  return SOGM_OK;
}

int sogm_bezier_qp_solve(sogm_planner *p, const double *start_pva, const double *goal_pv,
                         const double *polys, const int32_t *nfaces, const int32_t *npoly,
                         double *out_cpts, int32_t *out_status, int32_t *out_iters, void *stream) {
  if (!p || !start_pva || !goal_pv || !polys || !nfaces || !npoly || !out_cpts || !out_status ||
      !out_iters)
    return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(p->map->device));
  hipStream_t st = (hipStream_t)stream;
  prof_begin(p->map, SOGM_PROF_QP, st);
  sogm::QpWorkspace qw0 = p->qw;
  qw0.ablate            = p->map->tune_i(SOGM_TUNE_QP_ABLATE);
  int rc = launch_qp(p->pp, p->qs, qw0, p->qc, p->sel_count, start_pva, goal_pv, polys,
                     nfaces, npoly, out_cpts, out_status, out_iters, st, p->sel_first);
  prof_end(p->map, SOGM_PROF_QP, st);
  if (rc) {
    sogm::set_error("k_qp", hipGetLastError());
    return SOGM_ERR_HIP;
  }
  return SOGM_OK;
}
int sogm_bezier_qp_solve_timed(sogm_planner *p, const double *start_pva, const double *end_pva,
                               const double *time_alloc, double max_vel, double max_acc, const double *polys,
                               const int32_t *nfaces, const int32_t *npoly, double *out_cpts, int32_t *out_status,
                               int32_t *out_iters, void *stream) {
  if (!p || !start_pva || !end_pva || !time_alloc || !polys || !nfaces || !npoly || !out_cpts || !out_status ||
      !out_iters || !(max_vel > 0 && max_vel < 1e20) || !(max_acc > 0 && max_acc < 1e20))
    return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(p->map->device));
  hipStream_t       st = (hipStream_t)stream;
  SogmPlannerParams pp = p->pp;
  pp.opt_max_vel       = max_vel;
  pp.opt_max_acc       = max_acc;
  sogm::QpWorkspace qw = p->qw;
  qw.t_alloc           = time_alloc;
  qw.goal_stride       = 9;
  prof_begin(p->map, SOGM_PROF_QP, st);
  int rc = launch_qp(pp, p->qs, qw, p->qc, p->sel_count, start_pva, end_pva, polys, nfaces, npoly, out_cpts,
                     out_status, out_iters, st, p->sel_first);
  prof_end(p->map, SOGM_PROF_QP, st);
  if (rc) {
    sogm::set_error("k_qp", hipGetLastError());
    return SOGM_ERR_HIP;
  }
  return SOGM_OK;
}
int sogm_safe_after_opt(sogm_planner *p, const double *cpts, const int32_t *npoly,
                        const SogmTrajRecord *records, int n_records, const int32_t *ego_ids,
                        const double *t_now, int32_t *out_safe, void *stream) {
  if (!p || !cpts || !npoly || !ego_ids || !t_now || !out_safe || n_records < 0 || (n_records > 0 && !records))
    return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(p->map->device));
  if (int rc = sogm::join_exchange(p->map, (hipStream_t)stream)) return rc;
  if (sogm::launch_deconflict(p->sel_count, cpts, npoly, records, n_records, ego_ids, t_now, out_safe,
                              (hipStream_t)stream, p->sel_first) != 0) {
    sogm::set_error("sogm_safe_after_opt", hipGetLastError());
    return SOGM_ERR_HIP;
  }
  return SOGM_OK;
}

int sogm_planner_select_agents(sogm_planner *p, int first, int count) {
  if (!p || first < 0 || count < 1 || first + count > p->map->n_agents) return SOGM_ERR_INVALID_ARG;
  p->sel_first = first;
  p->sel_count = count;
  return SOGM_OK;
}
int sogm_planner_set_search_mode(sogm_planner *p, int mode) {
  if (!p || mode < 0 || (mode & ~(3 | 4 | 16)) != 0 || (mode & 3) == 3) return SOGM_ERR_INVALID_ARG;
  p->search_mode = mode;
  return SOGM_OK;
}

// diagnostics (tools/ only): the dataflow control block, copied on a private stream while the tick's kernels run
int sogm_debug_flow_peek(sogm_planner *p, int *out_host, int n) {
  if (!p || !out_host || n < 0) return SOGM_ERR_INVALID_ARG;
  if (!p->peek) SOGM_HIP_CHECK(hipStreamCreateWithFlags(&p->peek, hipStreamNonBlocking));
  hipStream_t peek = p->peek;
  const int nf = FLOW_HDR + 6 * p->map->n_agents;
  SOGM_HIP_CHECK(hipMemcpyAsync(out_host, p->d_flow, sizeof(int) * (size_t)(n < nf ? n : nf), hipMemcpyDeviceToHost, peek));
  if (n > nf)  // followed by d_safe (progress markers in debug builds)
    SOGM_HIP_CHECK(hipMemcpyAsync(out_host + nf, p->d_safe, sizeof(int) * (size_t)(n - nf), hipMemcpyDeviceToHost, peek));
  SOGM_HIP_CHECK(hipStreamSynchronize(peek));
  return SOGM_OK;
}

// diagnostics (tools/ only): per-agent stage timestamps of the last dataflow replan, [A][8] ticks of 10 ns
int sogm_debug_flow_times(sogm_planner *p, long long *out_host) {
  if (!p || !out_host) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  SOGM_HIP_CHECK(hipMemcpy(out_host, p->d_flow_ts, sizeof(long long) * 8 * (size_t)p->map->n_agents, hipMemcpyDeviceToHost));
  return SOGM_OK;
}

// diagnostics (tools/ only): the internal stage outputs of the last sogm_replan, copied to the host.  which: 0 polytopes
// [A][16][max_faces][4] f64, 1 faces per polytope [A][16] i32, 2 polytopes per agent [A] i32, 3 local goal [A][6] f64,
// 4 route [A][route_cap][6] f64, 5 route length [A] i32, 6 control points [A][16 * 15] f64, 7 QP iterations [A] i32,
// 8 QP status [A] i32, 9 search return codes [A] i32, 10 search stats [A][4] i32
int sogm_debug_planner_buffer(sogm_planner *p, int which, void *out_host, size_t bytes) {
  if (!p || !out_host) return SOGM_ERR_INVALID_ARG;
  const size_t A = (size_t)p->map->n_agents, slots = A * SOGM_MAX_PIECES;
  const void  *src[11] = {p->d_polys, p->d_nfaces, p->d_npoly, p->d_goal, p->d_route, p->d_route_len, p->d_cpts, p->d_iters,
                          p->d_status, p->d_ret, p->d_stats};
  const size_t len[11] = {sizeof(double) * slots * p->pp.max_faces * 4, sizeof(int32_t) * slots, sizeof(int32_t) * A,
                          sizeof(double) * 6 * A, sizeof(double) * 6 * p->route_cap * A, sizeof(int32_t) * A,
                          sizeof(double) * slots * 15, sizeof(int32_t) * A, sizeof(int32_t) * A, sizeof(int32_t) * A,
                          sizeof(int32_t) * 4 * A};
  if (which < 0 || which > 10 || bytes > len[which]) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  SOGM_HIP_CHECK(hipMemcpy(out_host, src[which], bytes, hipMemcpyDeviceToHost));
  return SOGM_OK;
}

// diagnostics (tools/ only): per-agent stamps of the last pre-stamp, [A][4] ticks of 10 ns: the agent's record seen by its
// first ticket, cylinders culled, last bits ticket done, last marks ticket done
int sogm_debug_prestamp_times(sogm_planner *p, long long *out_host) {
  if (!p || !out_host) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  SOGM_HIP_CHECK(hipMemcpy(out_host, p->d_flow_ts + 8 * (size_t)p->map->n_agents, sizeof(long long) * 4 * (size_t)p->map->n_agents,
                           hipMemcpyDeviceToHost));
  return SOGM_OK;
}

int sogm_planner_flow_error(sogm_planner *p) {
  if (!p) return SOGM_ERR_INVALID_ARG;
  int hdr[FLOW_HDR];
  SOGM_HIP_CHECK(hipSetDevice(p->map->device));
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  SOGM_HIP_CHECK(hipMemcpy(hdr, p->d_flow, sizeof(hdr), hipMemcpyDeviceToHost));
  return hdr[FLOW_ERR];
}

int sogm_planner_flow_failures(sogm_planner *p, int32_t out[2]) {
  if (!p || !out) return SOGM_ERR_INVALID_ARG;
  volatile int *h = p->h_flow_fail;
  out[0]          = h ? h[0] : 0;
  out[1]          = h ? h[1] : 0;
  return SOGM_OK;
}

int sogm_planner_counters(sogm_planner *p, int64_t *out_host, int reset) {
  if (!p || !out_host) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(p->map->device));
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  SOGM_HIP_CHECK(hipMemcpy(out_host, p->cw.counters, sizeof(int64_t) * SOGM_CNT_N, hipMemcpyDeviceToHost));
  if (reset) SOGM_HIP_CHECK(hipMemset(p->cw.counters, 0, sizeof(int64_t) * SOGM_CNT_N));
  return SOGM_OK;
}

int sogm_planner_set_publish(sogm_planner *p, SogmTrajRecord *own_records, SogmTrajRecord *next_table) {
  if (!p || (next_table && !own_records)) return SOGM_ERR_INVALID_ARG;
  p->pub_own   = own_records;
  p->pub_table = next_table;
  return SOGM_OK;
}

int sogm_planner_set_prestamp(sogm_planner *p, const SogmPrestamp *ps) {
  if (!p) return SOGM_ERR_INVALID_ARG;
  if (!ps) {
    p->ps_on = 0;
    return SOGM_OK;
  }
  if (!ps->hover_inout || !ps->out_now || !ps->out_t_start || !ps->out_pva) return SOGM_ERR_INVALID_ARG;
  const SogmWorld *w = ps->world;
  if (w) {
    if (!w->cloud_xyz || !w->block_bounds || w->n_points < 0 || w->block_points < 64 || w->block_points > 4096 ||
        w->n_blocks != (w->n_points + w->block_points - 1) / w->block_points || w->n_cyl < 0 || (w->n_cyl > 0 && !w->cylinders))
      return SOGM_ERR_INVALID_ARG;
  } else if (!ps->cloud_xyz || !ps->cloud_range || ps->n_cyl < 0 || (ps->n_cyl > 0 && !ps->cylinders)) {
    return SOGM_ERR_INVALID_ARG;
  }
  std::memset(&p->ps, 0, sizeof(p->ps));
  p->ps.cloud        = w ? w->cloud_xyz : ps->cloud_xyz;
  p->ps.cloud_range  = w ? nullptr : ps->cloud_range;
  p->ps.cyl          = w ? w->cylinders : ps->cylinders;
  p->ps.n_cyl        = w ? w->n_cyl : ps->n_cyl;
  p->ps_world_on     = w ? 1 : 0;
  if (w) {
    p->ps_world = *w;
    // the crop lists grow HERE if this frame holds more blocks than any before it (world_blocks then drains the device and
    // re-allocates): between two calls of the tick, never inside sogm_replan with the tick's persistent kernels queued
    sogm::CloudBlocks sized{};
    if (int rc = sogm::world_blocks(p->map, w, &sized)) return rc;
  }
  p->ps.stamp        = ps->next_stamp;
  p->ps.start_offset = ps->replan_start_offset;
  p->ps.hover        = ps->hover_inout;
  p->ps.now          = ps->out_now;
  p->ps.t_start      = ps->out_t_start;
  p->ps.pva          = ps->out_pva;
  p->ps.poses_host   = ps->out_poses;
  p->ps_on           = 1;
  return SOGM_OK;
}

int sogm_planner_set_swarm(sogm_planner *p, const SogmTrajRecord *records, int n_records,
                           const int32_t *ego_ids, const double *t_now) {
  if (!p || n_records < 0 || (records && (!ego_ids || !t_now))) return SOGM_ERR_INVALID_ARG;
  p->swarm     = records;
  p->n_swarm   = records ? n_records : 0;
  p->swarm_ego = ego_ids;
  p->swarm_now = t_now;
  return SOGM_OK;
}

// The dataflow replan: five launches for the whole tick.  k_astar (one workgroup per agent) publishes agents as their
// searches finish; k_corridor_flow, k_qp_flow and k_finish_flow are persistent and chain per agent through ready
// lists in HBM, so a slow search / corridor / QP only delays its own agent's chain.  k_flow_gate makes the waiting
// kernels dispatch only after every search is resident (they could otherwise fill the CUs and starve it).
// The dataflow replan's control block and outputs back to their start values: counters / seg_done 0, ready lists -1,
// verdicts 0, ok 0, records empty.
// epoch_word (dense clear's gate, sogm_device.hpp): this replan's epoch, written by the block that reset the counters
// and after them.
// ONE workgroup, on the corridor stream, launched before the searches and run under the map update (it only touches the
// planner's own words): when everything is back to its start value the generation word gets this replan's number — the
// search workgroups wait for it in their prologue (no event between the update's last kernel and k_astar on the caller's
// stream: the record's marker cost 55 us there) and zero their agent's ok / record themselves.
__global__ __launch_bounds__(1024) void k_flow_reset(int *hdr, int n_hdr, int *ready, int n_ready, int *verdict,
                                                     int n_verdict, int *epoch_word, int epoch, int *gen_word, int gen) {
  const int i0 = (int)threadIdx.x, step = (int)blockDim.x;
  for (int i = i0; i < n_hdr; i += step) hdr[i] = 0;
  for (int i = i0; i < n_ready; i += step) ready[i] = -1;
  if (verdict)
    for (int i = i0; i < n_verdict; i += step) verdict[i] = 0;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    if (epoch_word) __hip_atomic_store(epoch_word, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(gen_word, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// order[rank] = agent, rank = the number of agents whose chain of this tick (search start -> finished) was longer (ties:
// lower index first); by == 0: identity
__global__ void k_update_rank(const long long *__restrict__ ts, int *__restrict__ order, int n, int by) {
  for (int a = threadIdx.x; a < n; a += blockDim.x) {
    if (!by) {
      order[a] = a;
      continue;
    }
    const long long da = ts[a * 8 + 6] - ts[a * 8 + 0];
    int             r  = 0;
    for (int b = 0; b < n; ++b) {
      const long long db = ts[b * 8 + 6] - ts[b * 8 + 0];
      r += (db > da || (db == da && b < a)) ? 1 : 0;
    }
    order[r] = a;
  }
}

static int replan_flow(sogm_planner *p, const double *start_pva, const double *goal, const double *t_start,
                       const int32_t *drone_ids, SogmTrajRecord *out_records, int32_t *out_ok, void *stream) {
  sogm_ctx     *c    = p->map;
  hipStream_t   main = (hipStream_t)stream;
  const int     A    = c->n_agents;
  const MapView mv   = view_of(c);
  if (p->swarm)
    if (int rc = sogm::join_exchange(c, main)) return rc;
  // (the searches run on the caller's stream: they are the first kernel of the chain, and a hop to another stream
  //  costs an event round trip on the critical path)
  hipStream_t sA = main, sC = p->fstream[1], sQ = p->fstream[2], sF = p->fstream[3];
  // reset the control block in stream order: counters and seg_done to 0, ready lists to -1
  // (one launch instead of five memset nodes: each cost a dispatch gap on the tick's critical path)
  // k_finish_flow writes ok / the record of every agent whose chain completes; an agent whose chain does NOT (a wait
  // timed out, FLOW_ERR) must report ok = 0 and an empty record, not the previous tick's
  const bool spec = p->spec_astar != 0;
  // The reset on the corridor stream: behind the previous replan's report (that stream's last kernel: below) and, in the
  // pre-stamping variant, behind the pre-stamp's end; the waiting kernels of THIS replan follow it on the same stream or
  // behind ev_gate.  The readers of the swarm table on those streams wait for an all-gather in flight themselves.
  SOGM_HIP_CHECK(hipStreamWaitEvent(sC, p->ev_pdone, 0));  // (never recorded, or long complete, without a pre-stamp: no wait)
  if (c->cur_prestamped) {
    // a grid adopted by sogm_update_prestamped: its overlay (on the caller's stream) waits per agent on the control
    // block's stage words — the reset must not run under it.  An event on the caller's stream, as before (this variant pays
    // the marker in front of its searches)
    SOGM_HIP_CHECK(hipEventRecord(p->ev_gate, main));
    SOGM_HIP_CHECK(hipStreamWaitEvent(sC, p->ev_gate, 0));
  }
  int *gen_word = p->d_flow + FLOW_HDR + 6 * (size_t)A;
  ++p->reset_epoch;
  hipLaunchKernelGGL(k_flow_reset, dim3(1), dim3(1024), 0, sC, p->d_flow, FLOW_HDR + 2 * A, p->fc.a_ready, 4 * A,
                     spec ? p->aw.verdict : nullptr, A,
                     c->overlap >= 2 && c->clear_gate ? c->clear_epoch_word : nullptr,
                     c->overlap >= 2 && c->clear_gate ? sogm::next_clear_epoch(c) : 0, gen_word, p->reset_epoch);
  SOGM_HIP_CHECK(hipGetLastError());
  if (c->exchange_pending)
    for (int k = 1; k < 4; ++k) SOGM_HIP_CHECK(hipStreamWaitEvent(p->fstream[k], c->ev_xdone, 0));
  prof_begin(c, SOGM_PROF_ASTAR, sA);
  // an update flow still building this tick's maps (sogm_update_world with update_flow): the searches are launched beside
  // it and every search workgroup waits for ITS agent's map; the caller's stream joins the flow's end at the fan-in
  sogm::FlowCtl fca = p->fc;
  fca.map_ready = c->update_pending ? c->d_map_ready : nullptr;
  fca.map_epoch = c->map_epoch;
  fca.reset_gen   = gen_word;
  fca.reset_epoch = p->reset_epoch;
  fca.out_ok      = out_ok;
  fca.out_records = reinterpret_cast<int *>(out_records);
  fca.rec_words   = (int)(sizeof(SogmTrajRecord) / sizeof(int));
  if (launch_astar(mv, p->ap, p->pp.corridor_tau, astar_ws(p), A, start_pva, goal, t_start, p->d_ret, p->d_route,
                   p->d_route_len, p->route_cap, p->d_stats, nullptr, 0, sA, 0, &fca, spec ? 8 : 0)) {
    sogm::set_error("sogm_replan: k_astar", hipGetLastError());
    return SOGM_ERR_HIP;
  }
  prof_end(c, SOGM_PROF_ASTAR, sA);
  // "the searches are launched" for the side-stream work that starts from here (spare grids' clears, the pre-stamp):
  // recorded BEHIND k_astar — in front of it the record's marker held the searches back by 55 us
  SOGM_HIP_CHECK(hipEventRecord(p->ev_in, main));
  if (sogm::launch_flow_gate(p->fc, spec ? 2 * A : A, sC)) return SOGM_ERR_HIP;
  SOGM_HIP_CHECK(hipEventRecord(p->ev_gate, sC));
  SOGM_HIP_CHECK(hipStreamWaitEvent(sQ, p->ev_gate, 0));
  SOGM_HIP_CHECK(hipStreamWaitEvent(sF, p->ev_gate, 0));
  // persistent workgroups: one wave each for the corridor items (4 per CU fit), half of the CUs' worth of QP
  // workgroups (each takes a whole CU's LDS while it runs; with fewer, agents queue for a solver — measured: 64 / 96
  // / 128 / 160 workgroups -> 6.3 / 6.9 / 7.5 / 7.4 k replans/s), a handful of finishing waves
  int n_cu = 256;
  (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
  int wg_c = 4 * n_cu;
  if (wg_c > A * SOGM_MAX_PIECES) wg_c = A * SOGM_MAX_PIECES;
  int wg_q = n_cu / 2;
  if (c->tune_i(SOGM_TUNE_QP_WGS) > 0) wg_q = c->tune_i(SOGM_TUNE_QP_WGS);  // tuning aid
  if (wg_q > A) wg_q = A;
  if (wg_q < 1) wg_q = 1;
  int wg_f = A < 64 ? A : 64;
  prof_begin(c, SOGM_PROF_CORRIDOR, sC);
  if (sogm::launch_corridor_flow(mv, p->pp, p->cw, p->fc, A, wg_c, start_pva, t_start, p->d_route, p->d_route_len,
                                 p->route_cap, p->d_polys, p->d_nfaces, p->d_npoly, p->d_goal, sC)) {
    sogm::set_error("sogm_replan: k_corridor_flow", hipGetLastError());
    return SOGM_ERR_HIP;
  }
  prof_end(c, SOGM_PROF_CORRIDOR, sC);
  // (a QP workgroup takes a whole CU — registers and LDS — from the moment it is resident, and the first corridors are
  //  final long after an update flow has ended: the QP kernel is dispatched behind the flow's end, not beside it)
  if (c->update_pending) SOGM_HIP_CHECK(hipStreamWaitEvent(sQ, c->ev_udone, 0));
  prof_begin(c, SOGM_PROF_QP, sQ);
  if (sogm::launch_qp_flow(p->pp, p->qs, p->qw, p->qc, p->fc, A, wg_q, start_pva, p->d_goal, p->d_polys, p->d_nfaces,
                           p->d_npoly, p->d_cpts, p->d_status, p->d_iters, sQ)) {
    sogm::set_error("sogm_replan: k_qp_flow", hipGetLastError());
    return SOGM_ERR_HIP;
  }
  prof_end(c, SOGM_PROF_QP, sQ);
  // (queued AFTER the searches / corridor / QP launches: its gate waits for the corridor stage, and when streams share a
  //  hardware queue a launch behind a waiting gate waits with it — a QP kernel started 5 ms late costs the tick 5 ms)
  c->gate_open_valid = 0;
  c->gate_frac_valid = 0;
  {
    const double f      = c->tune[SOGM_TUNE_PRESTAMP_GATE_FRAC];
    c->gate_frac_agents = f > 0.0 && f < 1.0 ? ((int)(f * A + 0.5) < 1 ? 1 : (int)(f * A + 0.5)) : 0;
  }
  if (c->overlap >= 2) {
    // the side-stream clear of the grid this tick's update swapped out: narrow, with a wide second launch that joins
    // once every agent's corridors are final (FLOW_Q_READY_N == A, registered as the gate at planner creation) —
    // announce this replan's epoch now that the counters are reset
    int rc = sogm::queue_spare_clears(c, p->ev_in);  // grids still dirty (first ticks, pool changes)
    if (rc) return rc;
  }
  sogm::FlowCtl fcf = p->fc;
  // (only with the sparse reset: the pre-stamp runs behind the target grid's reset on the side stream, and with
  //  SOGM_SPARSE_RESET=0 a dense clear fills that stream for the whole tick.  On the first ticks of a flight — and
  //  after a dense writer — queue_spare_clears above has queued dense clears even in sparse mode: the pre-stamp, the
  //  report and ev_pdone then sit behind them, the next sogm_update_prestamped overlay waits per agent for as long
  //  (tests/test_pipelining_gpu.py::test_overlay_under_the_prestamp_tail_at_full_size flies exactly that).)
  const bool prestamp = p->ps_on && c->sparse && c->overlap >= 2 && c->n_ready > 0 && p->pub_own && c->clear_gate;
  if (!prestamp) fcf.p_ready = nullptr;
  if (sogm::launch_finish_flow(fcf, A, wg_f, p->pp.corridor_tau, p->d_ret, p->d_npoly, p->d_status, p->d_cpts,
                               p->swarm, p->n_swarm, p->swarm_ego, p->swarm_now, t_start, drone_ids, out_records,
                               out_ok, p->d_safe, p->cw.counters, sF, p->pub_own, p->pub_table)) {
    sogm::set_error("sogm_replan: k_finish_flow", hipGetLastError());
    return SOGM_ERR_HIP;
  }
  // pre-stamp (sogm_planner_set_prestamp): the next tick's map, agent by agent as their records are published, into
  // the pool's next grid — behind the gate that keeps store streams away from the searches and point scans, and
  // behind that grid's reset
  c->prestamp_slot = -1;
  if (prestamp) {
    const int nxt = c->ready[0];
    sogm::PrestampDev d = p->ps;
    d.grid     = (void *)c->pool[nxt];
    d.lg       = sogm::mark_log(c, nxt);
    d.own      = p->pub_own;
    d.poses    = c->d_poses_next;
    d.stamps   = c->d_stamps_next;
    d.n_agents = A;
    {
      const double f = c->tune[SOGM_TUNE_PRESTAMP_GATE_FRAC];
      d.gate_agents  = f >= 1.0 ? A : f <= 0.0 ? 0 : (int)(f * A + 0.5);
    }
    // tickets per agent (tuning aids): too few and the last agents' stamps trail the replan, too many and the
    // hand-overs cost more than the work (bits + marks 16+16 / 32+64 / 64+64 / 128+256 tickets: 13.1 / 12.1 / 12.2 /
    // 14.1 ms per tick, 13.0 without the pre-stamp); finer tickets for the last agents to be published (none / 8 / 16
    // late agents: 12.06 / 11.85 / 11.98 ms per tick)
    const int nb = c->tune_i(SOGM_TUNE_PRESTAMP_BITS) > 0 ? c->tune_i(SOGM_TUNE_PRESTAMP_BITS) : 32;
    const int nm = c->tune_i(SOGM_TUNE_PRESTAMP_MARKS) > 0 ? c->tune_i(SOGM_TUNE_PRESTAMP_MARKS) : 64;
    d.n_bits       = nb;
    d.n_marks      = nm;
    d.n_late       = c->tune_i(SOGM_TUNE_PRESTAMP_LATE_AGENTS) < 0 ? 0 : c->tune_i(SOGM_TUNE_PRESTAMP_LATE_AGENTS);
    d.n_bits_late  = c->tune_i(SOGM_TUNE_PRESTAMP_LATE_BITS) > 0 ? c->tune_i(SOGM_TUNE_PRESTAMP_LATE_BITS) : nb;
    d.n_marks_late = c->tune_i(SOGM_TUNE_PRESTAMP_LATE_MARKS) > 0 ? c->tune_i(SOGM_TUNE_PRESTAMP_LATE_MARKS) : nm;
    if (int rc = sogm::prestamp_buffers(c, &d)) return rc;
    if (p->ps_world_on)  // the frame's blocks + the context's crop lists (built by each agent's first ticket)
      if (int rc = sogm::world_blocks(c, &p->ps_world, &d.cb)) return rc;
    // The pre-stamp's target grid was reset by an EARLIER replan when the pool holds three grids (the front of the ready
    // queue), yet on the resets' stream the launch would also sit behind THIS replan's reset — which is held back until
    // every agent's corridors are final and then takes a millisecond: traces showed the pre-stamp starting at 5.6 ms
    // of the tick, working off a backlog of ~100 published agents on the CUs the QP workgroups leave free, and ending
    // 1-2.5 ms after the last QP in half of the ticks — the tick's end.  It runs on a stream of its own, ordered
    // behind its target's reset (and log restart) by that grid's event and behind "every agent's corridors are final"
    // by an event recorded after a gate kernel on the resets' stream: reset and pre-stamp then run side by side.  The
    // pre-stamp's gate is "prestamp_gate_frac of the agents' corridors are final" (default 0.9): waiting for ALL of them
    // makes 127 agents' stamps wait for one late search — seen as ticks ending 1.5-2.9 ms after their slowest chain — while
    // the few agents still searching or scanning points pay little for the stores beside them (same box: burst 9.07 ->
    // 8.81 ms, 300 sustained ticks 10.3 -> 11.3 k replans/s; 0.5 starts too early and loses late in a flight).  The gate
    // with the lower target is a kernel on the resets' stream — which spins for the full gate anyway — followed by an
    // event; a first version put a spinning gate at the head of THIS stream, one more spinner for shared or
    // oversubscribed hardware queues to stall on.  (prestamp_stream = 0: round 3's placement on the resets' stream.)
    const bool  own_stream = c->tune_i(SOGM_TUNE_PRESTAMP_STREAM) != 0 && c->pstream != nullptr;
    hipStream_t pst        = own_stream ? c->pstream : c->side;
    SOGM_HIP_CHECK(hipStreamWaitEvent(pst, p->ev_in, 0));
    if (own_stream) {
      SOGM_HIP_CHECK(hipStreamWaitEvent(pst, c->pool_ev[nxt], 0));
      if (c->gate_frac_valid && d.gate_agents < A)
        SOGM_HIP_CHECK(hipStreamWaitEvent(pst, c->ev_gate_frac, 0));
      else if (c->gate_open_valid && d.gate_agents >= A)
        SOGM_HIP_CHECK(hipStreamWaitEvent(pst, c->ev_gate_open, 0));
    }
    int wg_p = 8 * n_cu;  // one-wave workgroups (512 / 1024 / 2048+: 14.0 / 12.3 / 12.1 ms per tick)
    if (c->tune_i(SOGM_TUNE_PRESTAMP_WGS) > 0) wg_p = c->tune_i(SOGM_TUNE_PRESTAMP_WGS);
    if (sogm::launch_prestamp_flow(c->geom, p->fc, d, wg_p, wg_q, wg_f < A ? wg_f : A, pst)) {
      sogm::set_error("sogm_replan: k_prestamp_flow", hipGetLastError());
      return SOGM_ERR_HIP;
    }
    c->prestamp_slot = nxt;
    c->n_stamps++;
    c->ps_fail_host = p->h_flow_fail;  // sogm_update_prestamped refuses the grid if this replan turns out to have failed
    c->ps_fail_seen = p->h_flow_fail ? p->h_flow_fail[1] : 0;
  }
  // fan-in.  Without a pre-stamp the tick's report runs on the corridor stream behind the QP and finishing kernels, so that
  // the NEXT replan's reset — same stream — is ordered behind it without an event on the caller's stream
  const bool report_on_sc = !(c->prestamp_slot >= 0);
  for (int k = 0; k < 4; ++k)
    if (k != 1) SOGM_HIP_CHECK(hipEventRecord(p->ev_fdone[k], p->fstream[k]));
  if (report_on_sc) {
    SOGM_HIP_CHECK(hipStreamWaitEvent(sC, p->ev_fdone[2], 0));
    SOGM_HIP_CHECK(hipStreamWaitEvent(sC, p->ev_fdone[3], 0));
    const int retire_c = c->tune_i(SOGM_TUNE_CLEAR_RETIRE_AT_END);
    hipLaunchKernelGGL(k_flow_report, dim3(1), dim3(1), 0, sC, (const int *)p->d_flow, p->h_flow_fail,
                       retire_c ? p->d_epoch : (int *)nullptr, c->h_tick_clock);
    SOGM_HIP_CHECK(hipGetLastError());
  }
  SOGM_HIP_CHECK(hipEventRecord(p->ev_fdone[1], sC));
  for (int k = 0; k < 4; ++k) SOGM_HIP_CHECK(hipStreamWaitEvent(main, p->ev_fdone[k], 0));
  if (int rc = sogm::join_update(c, main)) return rc;  // (complete long ago: every search waited for its agent's map)
  if (c->d_update_order && c->tune_i(SOGM_TUNE_UPDATE_FLOW) != 0) {
    // the next update flow takes the agents in the order of this tick's chain lengths, longest first (a schedule only)
    hipLaunchKernelGGL(k_update_rank, dim3(1), dim3(256), 0, main, (const long long *)p->fc.ts, c->d_update_order, A,
                       c->tune_i(SOGM_TUNE_UPDATE_ORDER));
    SOGM_HIP_CHECK(hipGetLastError());
  }
  bool reported = report_on_sc;
  if (c->prestamp_slot >= 0) {
    // the tick's report behind the pre-stamp on ITS stream (the last kernel of the tick to end), so that the caller's
    // stream goes from the fan-in straight to the next tick's first kernel instead of through one more launch
    const int retire_p = c->tune_i(SOGM_TUNE_CLEAR_RETIRE_AT_END);
    hipStream_t rst = c->tune_i(SOGM_TUNE_PRESTAMP_STREAM) != 0 && c->pstream ? c->pstream : c->side;  // = pst above
    for (int k = 1; k < 4; ++k) SOGM_HIP_CHECK(hipStreamWaitEvent(rst, p->ev_fdone[k], 0));
    hipLaunchKernelGGL(k_flow_report, dim3(1), dim3(1), 0, rst, (const int *)p->d_flow, p->h_flow_fail,
                       retire_p ? p->d_epoch : (int *)nullptr, c->h_tick_clock);
    SOGM_HIP_CHECK(hipGetLastError());
    SOGM_HIP_CHECK(hipEventRecord(p->ev_pdone, rst));
    const int defer = c->tune_i(SOGM_TUNE_SPLAT_OVERLAP) != 0;  // 0: the caller's stream waits for the pre-stamp's end here
    if (defer) {
      // the caller's stream goes on behind the fan-in: the next update's overlay waits per agent (sogm_update_prestamped),
      // everything else joins the pre-stamp's end when it is called (sogm::join_prestamp)
      c->ev_pdone      = p->ev_pdone;
      c->pdone_pending = 1;
      c->ps_stage      = p->fc.stage;
      c->ps_err        = &p->fc.hdr[FLOW_ERR];
    } else {
      SOGM_HIP_CHECK(hipStreamWaitEvent(main, p->ev_pdone, 0));
    }
    reported = true;
  }
  // the exchange of the records this replan publishes may start when the finishing kernel is done (sogm_traj_allgather)
  c->records_final_valid = 0;
  if (p->pub_own) {
    c->ev_records_final    = p->ev_fdone[3];
    c->records_final_ptr   = p->pub_own;
    c->records_final_valid = 1;
  }
  const int retire = c->tune_i(SOGM_TUNE_CLEAR_RETIRE_AT_END);  // measured: tick -2 %, but the clear 14.5 -> 15.5 ms; off
  if (!reported) {
    hipLaunchKernelGGL(k_flow_report, dim3(1), dim3(1), 0, main, (const int *)p->d_flow, p->h_flow_fail,
                       retire ? p->d_epoch : (int *)nullptr, c->h_tick_clock);
    SOGM_HIP_CHECK(hipGetLastError());
  }
  return SOGM_OK;
}

static int replan_impl(sogm_planner *p, const double *start_pva, const double *goal,
                       const double *t_start, const int32_t *drone_ids, SogmTrajRecord *out_records,
                       int32_t *out_ok, void *stream) {
  sogm_ctx   *c    = p->map;
  hipStream_t main = (hipStream_t)stream;
  const int   A = c->n_agents, G = p->n_groups;
  const MapView mv = view_of(c);
  if (int rc = sogm::join_update(c, main)) return rc;
  for (int g = 0; g < G; ++g)
    if (!p->gstream[g]) SOGM_HIP_CHECK(sogm::create_stream_partitioned(&p->gstream[g], 1));
  // the swarm's records (deconfliction) may come from an all-gather still in flight on the exchange stream
  if (p->swarm)
    if (int rc = sogm::join_exchange(c, main)) return rc;
  // fan out: every group stream starts when the caller's stream has produced the inputs
  SOGM_HIP_CHECK(hipEventRecord(p->ev_in, main));
  if (c->overlap >= 2) {
    // double / triple-buffered SOGM: the spare grid this tick's update swapped out is cleared on the side stream
    // under this whole replan (its last readers, the previous tick's corridor kernels, are ordered before ev_in).
    // Mode 2: it is the NEXT update's grid (deadline: the next stamp).  Mode 3: the next update takes the grid
    // cleared during the PREVIOUS replan, so every clear has a whole tick of slack.
    int rc = sogm::queue_spare_clears(c, p->ev_in);
    if (rc) return rc;
  }
  for (int g = 0; g < G; ++g) {
    const int a0 = (int)((long long)A * g / G), a1 = (int)((long long)A * (g + 1) / G), n = a1 - a0;
    if (n <= 0) continue;
    hipStream_t st = p->gstream[g];
    SOGM_HIP_CHECK(hipStreamWaitEvent(st, p->ev_in, 0));
    if (launch_astar(mv, p->ap, p->pp.corridor_tau, astar_ws(p), n, start_pva, goal, t_start, p->d_ret,
                     p->d_route, p->d_route_len, p->route_cap, p->d_stats, nullptr, 0, st, a0) ||
        launch_corridor(mv, p->pp, p->cw, n, start_pva, t_start, p->d_route, p->d_route_len,
                        p->route_cap, p->d_polys, p->d_nfaces, p->d_npoly, p->d_goal, st, a0, p->ev_pts[g])) {
      sogm::set_error("sogm_replan launch", hipGetLastError());
      return SOGM_ERR_HIP;
    }
    SOGM_HIP_CHECK(hipEventRecord(p->ev_corr[g], st));
  }
  if (c->overlap == 1) {
    // Single grid, cleared in place for the next update in two parts on the side stream.  Head: a narrow launch
    // as soon as every group's obstacle-point kernel — the tick's last reader of the SOGM — is done; it shares the
    // machine with the FIRI kernels (about what they take to run: ~28 GB).  Rest: a full-width launch once every
    // group's corridor stage has finished with global memory (a full-width clear starves every concurrent load;
    // k_qp only touches LDS).
    const size_t total = sogm::clear_vec4_total(c);
    size_t       head  = (size_t)28e9 / 16;
    if (head > total / 2) head = total / 2;
    for (int g = 0; g < G; ++g) SOGM_HIP_CHECK(hipStreamWaitEvent(c->side, p->ev_pts[g], 0));
    const int slot = sogm::cur_slot(c);
    if (c->sparse && c->tracked[slot] && sogm::mark_log(c, slot).entries) {
      // sparse reset: the logged sectors only, as soon as the SOGM's last readers are done
      int rc = sogm::reset_slot(c, c->side, slot, c->d_grid, true);
      if (rc) return rc;
    } else {
      int rc = sogm::launch_clear(c, c->side, c->d_grid, true, 1, head);
      if (rc) return rc;
      for (int g = 0; g < G; ++g) SOGM_HIP_CHECK(hipStreamWaitEvent(c->side, p->ev_corr[g], 0));
      rc = sogm::launch_clear(c, c->side, c->d_grid, false, 2, head);
      if (rc) return rc;
    }
    SOGM_HIP_CHECK(hipEventRecord(c->ev_cleared, c->side));
    c->precleared = 1;
    c->updated    = 0;
  }
  for (int g = 0; g < G; ++g) {
    const int a0 = (int)((long long)A * g / G), a1 = (int)((long long)A * (g + 1) / G), n = a1 - a0;
    if (n <= 0) continue;
    hipStream_t st = p->gstream[g];
    if (launch_qp(p->pp, p->qs, p->qw, p->qc, n, start_pva, p->d_goal, p->d_polys, p->d_nfaces,
                  p->d_npoly, p->d_cpts, p->d_status, p->d_iters, st, a0)) {
      sogm::set_error("sogm_replan launch_qp", hipGetLastError());
      return SOGM_ERR_HIP;
    }
    if (p->swarm) {
      if (sogm::launch_deconflict(n, p->d_cpts, p->d_npoly, p->swarm, p->n_swarm, p->swarm_ego, p->swarm_now,
                                  p->d_safe, st, a0, p->cw.counters) != 0) {
        sogm::set_error("sogm_replan launch_deconflict", hipGetLastError());
        return SOGM_ERR_HIP;
      }
    }
    hipLaunchKernelGGL(k_pack_records, dim3((n + 63) / 64), dim3(64), 0, st, a1, p->pp.corridor_tau,
                       p->d_ret, p->d_npoly, p->d_status, p->swarm ? p->d_safe : nullptr, p->d_cpts, t_start,
                       drone_ids, out_records, out_ok, a0, p->cw.counters, p->pub_own, p->pub_table);
    SOGM_HIP_CHECK(hipGetLastError());
    SOGM_HIP_CHECK(hipEventRecord(p->ev_done[g], st));
    SOGM_HIP_CHECK(hipStreamWaitEvent(main, p->ev_done[g], 0));  // fan in
  }
  return SOGM_OK;
}


}  // extern "C"
// ---------------------------------------------------------------------------------------------------------------
// sogm_flight_run (the flight is described in sogm_planner.hpp / sogm_abi.h)
// ---------------------------------------------------------------------------------------------------------------
// control block back to its start values: header / rings / counters 0, every agent's first map item published in agent
// order, tick_of = first_tick, verdicts 0 (their tags are absolute tick numbers), the flight's log cleared (an agent-tick
// that does not complete reports ok = 0 and an empty record)
__global__ __launch_bounds__(256) void k_flight_reset(FlightCtl fl, int n_words, int *verdict, long long *acc, int *log_words,
                                                      long long n_log_words) {
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
  // header, rings, plain queues, tick_done, tick_of, seg_done, stage: one block.  NOT the head / tail words of the urgent
  // queue: its consumers decide "a descriptor is there" by LOADING tail and head (every other hand-over of the flight is a
  // read-modify-write or a generation-tagged slot), so the queue runs on over the planner's life instead of restarting
  // with every call: positions only grow (modulo 2^32; a slot's generation tag follows its position), and a value a
  // consumer reads late can only be LOWER than the true one — it then looks again — never claim a descriptor that
  // does not exist yet.
  for (long long i = i0; i < n_words; i += step)
    if (i != FL_UW_TAIL && i != FL_UW_HEAD) fl.hdr[i] = 0;
  if (i0 == 0) fl.hdr[FL_UW_HEAD] = fl.hdr[FL_UW_TAIL];  // (descriptors an aborted call left behind are skipped)
  for (long long i = i0; i < n_log_words; i += step) log_words[i] = 0;
  for (long long i = i0; i < fl.n_agents; i += step) verdict[i] = 0;
  for (long long i = i0; i < 8ll * fl.n_agents; i += step) acc[i] = 0;
  for (long long i = i0; i < 16; i += step) fl.prof[i] = 0ull;
  for (long long i = i0; i < 8 * FL_WG_LOG; i += step) fl.wg_start[i] = 0;
}
__global__ __launch_bounds__(256) void k_flight_seed(FlightCtl fl) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a < fl.n_agents) {
    fl.tick_of[a] = fl.first_tick;
    fl.m_ring[a]  = (1 << 16) | a;               // position a, generation 1: every agent's first head, in agent order
    fl.ts[(size_t)a * FL_TS + 7] = wall_clock64();  // the head's publication
  }
  if (a == 0) fl.hdr[FL_M_READY] = fl.n_agents;
  for (int i = a; i < FLIGHT_MAX_TICKS * fl.n_agents; i += (int)(gridDim.x * blockDim.x)) fl.parked[i] = -1;
}
// the exchange behind a multi-rank flight (sogm_flight_run with SogmFlight::nccl_comm), on the exchange stream, per tick i of the
// call: k_flight_xwait -> ncclAllGather(rows of ver(first_tick + i)) -> k_flight_xsignal
__global__ void k_flight_xwait(FlightCtl fl, int i) {  // every local agent has finished tick i: its rows of the version are final
  if (threadIdx.x != 0) return;
  while (__hip_atomic_load(&fl.tick_done[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < fl.n_agents) {
    // (no limit of its own: every wait INSIDE the flight is bounded and sets the error word, which ends this one — the
    //  collective behind it still runs, its peers are waiting in theirs)
    if (__hip_atomic_load(&fl.hdr[FL_ERR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    flow_pause();
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__global__ void k_flight_xsignal(FlightCtl fl, int i) {  // every rank's rows of ver(first_tick + i) are here
  if (threadIdx.x != 0) return;
  __threadfence();
  __hip_atomic_store(&fl.xready[i], fl.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  if (i + fl.lag < fl.n_ticks) fl_gate_release(fl, i + fl.lag);
}
__global__ void k_flight_report(const int *__restrict__ hdr, int *__restrict__ host_words) {
  const int e = hdr[FL_ERR];
  if (e != 0) {
    host_words[0] = 100 + e;
    host_words[1] = host_words[1] + 1;
  }
}

// the 16-CU unit of a mask bit: two CUs per XCD whether bit i names XCD i / 32 or XCD i % 8 (tools/micro/cumask.hip).
// unit = 4 * shader engine + r: the compute units of one shader engine (of every XCD) whose index is congruent r mod 4.
static int flight_unit_of(int i) { return ((i / 8) % 4) * 4 + ((i / 32 + i % 8) % 4); }

// Which units each of the four kernels gets.  The workgroup dispatcher hands the workgroups of a launch to the shader
// engines of its compute-unit mask in strict ROTATION: when one engine's units are full the dispatch stops, whatever room the
// other engines have (measured, tools/diag_flight_residency.py: a mask with four units in one engine and eight in another
// held 16 + 16 one-wave workgroups per XCD, not 16 + 32 — 120 of the round-5 corridor kernel's 384 workgroups sat in the
// dispatcher for the whole flight).  A workgroup that is not running is harmless until the hardware scheduler saves and
// restores the process's queues (any queue created or destroyed on the device, by any process, does that): the waiting
// workgroups then start in the slots the save freed, one per XCD ahead of the restored waves, and the wave that lost its slot
// stays saved until somebody leaves — the 3-second stalls of round 5.  So: every kernel's mask holds the SAME number of units
// in every shader engine it touches, and a launch has exactly as many workgroups as that mask holds at once
// (engines x smallest engine share): every workgroup is resident from the first microsecond, nothing waits in a dispatcher.
// grid[se][r] = kernel that owns unit 4 se + r (-1 free).  Kernels 0, 1, 3 (QP, search, map) are placed by a small search —
// fewest engines first — such that the rest (kernel 2, corridor + finish) is balanced too; if no such layout exists the most
// balanced one is kept and the launch is cut to what is resident.
struct FlightLayout {
  int grid[4][4];
  int resident_units[4];  // engines touched x smallest share, in units
};
static int layout_rest_score(const int grid[4][4]) {  // units of the rest that are resident under the rotation
  int touched = 0, smallest = 5;
  for (int se = 0; se < 4; ++se) {
    int n = 0;
    for (int r = 0; r < 4; ++r) n += grid[se][r] == -1;
    if (n > 0) {
      ++touched;
      if (n < smallest) smallest = n;
    }
  }
  return touched ? touched * smallest : 0;
}
static void layout_search(int grid[4][4], const int want[4], const int order[3], int depth, int rest_units, int *best_score,
                          int best[4][4]) {
  if (depth == 3) {
    const int sc = layout_rest_score(grid);
    if (sc > *best_score) {
      *best_score = sc;
      std::memcpy(best, grid, sizeof(int) * 16);
    }
    return;
  }
  const int k = order[depth], u = want[k];
  for (int n_se = 1; n_se <= 4 && *best_score < rest_units; ++n_se) {
    if (u % n_se != 0 || u / n_se > 4) continue;
    const int per = u / n_se;
    for (int set = 1; set < 16 && *best_score < rest_units; ++set) {
      if (__builtin_popcount((unsigned)set) != n_se) continue;
      int  saved[4][4];
      bool ok = true;
      std::memcpy(saved, grid, sizeof(saved));
      for (int se = 0; se < 4 && ok; ++se) {
        if (!((set >> se) & 1)) continue;
        int got = 0;
        for (int r = 0; r < 4 && got < per; ++r)
          if (grid[se][r] == -1) grid[se][r] = k, ++got;
        ok = got == per;
      }
      if (ok) layout_search(grid, want, order, depth + 1, rest_units, best_score, best);
      std::memcpy(grid, saved, sizeof(saved));
    }
  }
}
// engines [se0, se0 + n_se) are the flight's (the others belong to nobody here: -2); `reserved` units of them stay free of any
// kernel (-3), taken from the highest engine's highest classes first
static FlightLayout flight_layout(const int want[4], int se0 = 0, int n_se = 4, int reserved = 0) {
  FlightLayout L;
  int          grid[4][4], best_score = -1;
  for (int i = 0; i < 16; ++i) (&grid[0][0])[i] = -1, (&L.grid[0][0])[i] = -1;
  for (int se = 0; se < 4; ++se)
    if (se < se0 || se >= se0 + n_se)
      for (int r = 0; r < 4; ++r) grid[se][r] = -2;
  for (int se = se0 + n_se - 1, left = reserved; se >= se0 && left > 0; --se)
    for (int r = 3; r >= 0 && left > 0; --r)
      if (grid[se][r] == -1) grid[se][r] = -3, --left;
  std::memcpy(L.grid, grid, sizeof(grid));
  const int order[3] = {0, 3, 1};  // QP and map (whole engines when they can have them), then the search
  layout_search(grid, want, order, 0, want[2], &best_score, L.grid);
  for (int i = 0; i < 16; ++i)
    if ((&L.grid[0][0])[i] == -1) (&L.grid[0][0])[i] = 2;
  for (int k = 0; k < 4; ++k) {
    int touched = 0, smallest = 5;
    for (int se = 0; se < 4; ++se) {
      int n = 0;
      for (int r = 0; r < 4; ++r) n += L.grid[se][r] == k;
      if (n > 0) {
        ++touched;
        if (n < smallest) smallest = n;
      }
    }
    L.resident_units[k] = touched ? touched * smallest : 0;
  }
  return L;
}

static int flight_setup(sogm_planner *p) {
  sogm_ctx *c = p->map;
  const int A = c->n_agents;
  if (p->d_fl) return SOGM_OK;
  int ring = 1;
  while (ring < 2 * A) ring <<= 1;
  if (A >= (1 << 16)) return SOGM_ERR_INVALID_ARG;
  const size_t words = FL_HDR + 4 * (size_t)ring + 6 * (size_t)FL_WQ_SLOTS + 3 * FLIGHT_MAX_TICKS + 4 * (size_t)A +
                       (size_t)FLIGHT_MAX_TICKS * A + 2;
  SOGM_HIP_CHECK(hipMalloc((void **)&p->d_fl, sizeof(int) * words));
  SOGM_HIP_CHECK(hipMemset(p->d_fl, 0, sizeof(int) * words));  // (once: the urgent / priority queues start empty at position 0)
  int *q          = p->d_fl;
  p->fl.hdr       = q;              q += FL_HDR;
  p->fl.s_ring    = q;              q += ring;
  p->fl.q_ring    = q;              q += ring;
  p->fl.m_ring    = q;              q += ring;
  p->fl.u_ring    = q;              q += ring;
  p->fl.mw        = reinterpret_cast<unsigned long long *>(q);  q += 2 * FL_WQ_SLOTS;  // (FL_HDR and 4 x ring are even: 8-byte aligned)
  p->fl.lw        = reinterpret_cast<unsigned long long *>(q);  q += 2 * FL_WQ_SLOTS;
  p->fl.tick_done = q;              q += FLIGHT_MAX_TICKS;
  p->fl.parked_n  = q;              q += FLIGHT_MAX_TICKS;
  p->fl_xready    = q;              q += FLIGHT_MAX_TICKS;
  p->fl.tick_of   = q;              q += A;
  p->fl.seg_done  = q;              q += A;
  p->fl.stage     = q;              q += A;
  p->fl.urgent    = q;              q += A;
  p->fl.parked    = q;              q += (size_t)FLIGHT_MAX_TICKS * A;
  // (behind everything a call resets: the urgent / priority queues and their head / tail words are NOT reset — below)
  if ((q - p->d_fl) & 1) q += 1;
  p->fl.uw        = reinterpret_cast<unsigned long long *>(q);  q += 2 * FL_WQ_SLOTS;
  p->fl.ring_mask = ring - 1;
  p->fl.n_agents  = A;
  SOGM_HIP_CHECK(hipMalloc((void **)&p->fl.ts, sizeof(long long) * FL_TS * (size_t)A));
  SOGM_HIP_CHECK(hipMalloc((void **)&p->fl.acc, sizeof(long long) * 8 * (size_t)A));
  SOGM_HIP_CHECK(hipMemset(p->fl.ts, 0, sizeof(long long) * FL_TS * (size_t)A));
  SOGM_HIP_CHECK(hipMemset(p->fl.acc, 0, sizeof(long long) * 8 * (size_t)A));
  SOGM_HIP_CHECK(hipMalloc((void **)&p->fl.ts_log, sizeof(long long) * FL_TS * (size_t)A * FLIGHT_MAX_TICKS));
  SOGM_HIP_CHECK(hipMalloc((void **)&p->fl.wg_start, sizeof(long long) * 8 * FL_WG_LOG));
  SOGM_HIP_CHECK(hipMalloc((void **)&p->fl.prof, sizeof(unsigned long long) * 16));
  SOGM_HIP_CHECK(hipMemset(p->fl.prof, 0, sizeof(unsigned long long) * 16));
  SOGM_HIP_CHECK(hipMalloc((void **)&p->d_fl_worlds, sizeof(FlightWorld) * FLIGHT_MAX_TICKS));
  SOGM_HIP_CHECK(hipHostMalloc((void **)&p->h_fl_worlds, sizeof(FlightWorld) * FLIGHT_MAX_TICKS, hipHostMallocDefault));
  SOGM_HIP_CHECK(hipMalloc((void **)&p->d_fl_pva, sizeof(double) * 9 * (size_t)A));
  SOGM_HIP_CHECK(hipMalloc((void **)&p->d_fl_tstart, sizeof(double) * (size_t)A));
  SOGM_HIP_CHECK(hipMalloc((void **)&p->d_fl_now, sizeof(double) * (size_t)A));
  // the four streams and their compute units: QP, search, map take flight_*_units units of 16 CUs, corridor + finish
  // the rest; workgroups = what the partition holds at once (every workgroup of a flight kernel is resident from the
  // start: nothing waits for a workgroup that is not running)
  int n_cu = 256;
  (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
  int n_se = c->tune_i(SOGM_TUNE_FLIGHT_ENGINES), se0 = c->tune_i(SOGM_TUNE_FLIGHT_ENGINE_FIRST);
  if (se0 + n_se > 4) n_se = 4 - se0;
  const int reserved    = c->tune_i(SOGM_TUNE_FLIGHT_EXCHANGE_UNITS);
  const int units_total = 4 * n_se - reserved;
  int       u[4] = {c->tune_i(SOGM_TUNE_FLIGHT_QP_UNITS), c->tune_i(SOGM_TUNE_FLIGHT_SEARCH_UNITS), 0,
                    c->tune_i(SOGM_TUNE_FLIGHT_MAP_UNITS)};
  u[2] = units_total - u[0] - u[1] - u[3];
  if (u[2] < 1) {
    sogm::set_error_text("sogm_flight_run: flight_qp_units + flight_search_units + flight_map_units must leave a unit for the corridor kernel");
    return SOGM_ERR_INVALID_ARG;
  }
  const bool         masks = c->tune_i(SOGM_TUNE_FLIGHT_MASKS) != 0;
  const FlightLayout L     = flight_layout(u, se0, n_se, reserved);
  for (int k = 0; k < 4; ++k) {
    uint32_t mask[16] = {0};
    int      cus = 0;
    for (int i = 0; i < n_cu && i < 512; ++i) {
      const int un = flight_unit_of(i);
      if (L.grid[un / 4][un % 4] == k) {
        mask[i / 32] |= 1u << (i % 32);
        ++cus;
      }
    }
    // compute units that hold workgroups from the start: the dispatcher's rotation over the mask's shader engines stops at
    // the smallest engine share (flight_layout balances the shares; an unbalanced rest is cut here)
    p->fl_cus[k] = masks ? cus * L.resident_units[k] / (u[k] > 0 ? u[k] : 1) : cus;
    if (masks)
      SOGM_HIP_CHECK(hipExtStreamCreateWithCUMask(&p->fl_stream[k], (uint32_t)((n_cu + 31) / 32), mask));
    else
      SOGM_HIP_CHECK(hipStreamCreateWithFlags(&p->fl_stream[k], hipStreamNonBlocking));
    SOGM_HIP_CHECK(hipEventCreateWithFlags(&p->fl_ev_done[k], hipEventDisableTiming));
  }
  SOGM_HIP_CHECK(hipEventCreateWithFlags(&p->fl_ev_in, hipEventDisableTiming));
  p->fl_wgs[0] = p->fl_cus[0];      // one QP workgroup per CU (a whole CU's LDS and registers)
  p->fl_wgs[1] = p->fl_cus[1];      // one search workgroup per CU (124 KB of LDS)
  p->fl_wgs[2] = c->tune_i(SOGM_TUNE_FLIGHT_LIGHT_PER_CU) * p->fl_cus[2];  // corridor / finish waves: 39.8 KB of LDS, one per SIMD
  p->fl_wgs[3] = c->tune_i(SOGM_TUNE_FLIGHT_MAP_PER_CU) * p->fl_cus[3];    // map waves: two per SIMD
  SOGM_HIP_CHECK(hipStreamSynchronize(nullptr));
  return SOGM_OK;
}

#ifdef SOGM_FLIGHT_TRACE
#include <chrono>
#define FL_TRACE(what) std::fprintf(stderr, "FLTRACE %p %s %.3f ms\n", (void *)p, what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count())
#else
#define FL_TRACE(what) (void)0
#endif
extern "C" {
int sogm_flight_run(sogm_planner *p, const SogmFlight *f, void *stream) {
  if (!p || !f || f->n_ticks < 1 || f->n_ticks > FLIGHT_MAX_TICKS || f->first_tick < 0 || !f->worlds || !f->goals ||
      !f->drone_ids || !f->hover_inout || !f->own_inout || !f->tables || !f->log_records || !f->log_ok || !(f->period > 0))
    return SOGM_ERR_INVALID_ARG;
  sogm_ctx *c = p->map;
  const int A = c->n_agents;
  if (f->n_total < A || f->agent0 < 0 || f->agent0 + A > f->n_total) return SOGM_ERR_INVALID_ARG;
  const bool xchg = f->n_total != A && f->nccl_comm != nullptr;  // the exchange runs behind the call (k_flight_xwait / xsignal)
  const int lag = c->tune_i(SOGM_TUNE_FLIGHT_NEIGHBOUR_LAG) == 1 ? 1 : 2;
  if (f->n_total != A && !xchg && f->n_ticks > lag) {
    // several ranks, no communicator: the rows of the OTHER ranks' agents in ver(k - 2) must be complete before tick k starts,
    // and only the host can put them there (an all-gather of the finished versions between two calls): two ticks per call
    sogm::set_error_text("sogm_flight_run: with n_total > n_agents (other ranks' rows in the tables) and no nccl_comm a call flies at most two ticks");
    return SOGM_ERR_INVALID_ARG;
  }
  if (!p->flow || !c->sparse || !c->d_body || c->n_body <= 0) return SOGM_ERR_STATE;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  hipStream_t main = (hipStream_t)stream;
  if (int rc = sogm::join_update(c, main)) return rc;
  if (int rc = sogm::join_prestamp(c, main)) return rc;
  if (int rc = sogm::join_exchange(c, main)) return rc;
  if (int rc = flight_setup(p)) return rc;
  {  // the work queues hold the descriptors of two ticks at most (an agent is at most one tick ahead of the slowest)
    const long long per_tick = (long long)A * (1 + c->tune_i(SOGM_TUNE_FLIGHT_RESET) + c->tune_i(SOGM_TUNE_FLIGHT_BITS) +
                                               c->tune_i(SOGM_TUNE_FLIGHT_MARKS) + c->tune_i(SOGM_TUNE_FLIGHT_SPLAT));
    if (2 * per_tick > FL_WQ_SLOTS || 2ll * A * (SOGM_MAX_PIECES + 1) > FL_WQ_SLOTS) {
      sogm::set_error_text("sogm_flight_run: agents x tickets per tick exceed the work queue (lower flight_marks / flight_bits)");
      return SOGM_ERR_CAPACITY;
    }
  }
  const size_t agent_bytes = (size_t)c->spec.T * (size_t)c->geom.V * c->cell_bytes();
  if (agent_bytes % 32 != 0 || (reinterpret_cast<uintptr_t>(c->d_grid) & 31) != 0) {
    sogm::set_error_text("sogm_flight_run: agent grids must be 32-byte aligned (V * T * cell bytes a multiple of 32)");
    return SOGM_ERR_INVALID_ARG;
  }
  int max_blocks = 0;
  for (int i = 0; i < f->n_ticks; ++i) {
    const SogmWorld &w = f->worlds[i];
    if (!w.cloud_xyz || !w.block_bounds || w.n_points < 0 || w.block_points < 64 || w.block_points > 4096 ||
        w.n_blocks != (w.n_points + w.block_points - 1) / w.block_points || w.n_cyl < 0 || (w.n_cyl > 0 && !w.cylinders))
      return SOGM_ERR_INVALID_ARG;
    if (w.n_blocks > max_blocks) max_blocks = w.n_blocks;
    p->h_fl_worlds[i] = FlightWorld{w.cloud_xyz, w.block_bounds, w.cylinders, w.n_points, w.n_blocks, w.block_points, w.n_cyl};
  }
  sogm::FlightMapDev md{};
  {
    SogmWorld big = f->worlds[0];  // (only its block count sizes the lists)
    big.n_blocks  = max_blocks;
    big.n_points  = max_blocks * big.block_points;
    if (int rc = sogm::world_blocks(c, &big, &md.cb)) return rc;
    md.cb.n_blocks = max_blocks;  // (per frame in the kernel; the lists' row length is md.cb.row = the context's capacity)
  }
  // the agent's single grid of the flight is the context's current one; it must be covered by its mark log
  const int slot = sogm::cur_slot(c);
  if (c->precleared && c->overlap >= 2) {  // (a pooled tick path left a spare grid queued: nothing to adopt here)
  }
  if (!c->tracked[slot] || !sogm::mark_log(c, slot).entries) {
    if (int rc = sogm::launch_clear(c, main, c->d_grid, false)) return rc;  // dense, once; restarts the log
    if (!c->tracked[slot]) {
      sogm::set_error_text("sogm_flight_run: the current grid has no mark log (sogm_set_sparse_reset)");
      return SOGM_ERR_STATE;
    }
  }
  int words = 0;
  {
    sogm::PrestampDev tmp{};
    if (int rc = sogm::prestamp_buffers(c, &tmp)) return rc;  // stamp scratch: bits, candidates
    md.bits   = tmp.bits;
    md.words  = tmp.words;
    md.cand   = tmp.cand;
    md.n_cand = tmp.n_cand;
    words     = tmp.words;
  }
  (void)words;
  FlightCtl fl   = p->fl;
  fl.xready      = xchg ? p->fl_xready : nullptr;
  fl.lag         = lag;
  fl.n_ticks     = f->n_ticks;
  fl.first_tick  = f->first_tick;
  md.grid        = (void *)c->d_grid;
  md.worlds      = p->d_fl_worlds;
  md.lg          = sogm::mark_log(c, slot);
  md.own         = f->own_inout;
  md.tables      = f->tables;
  md.n_total     = f->n_total;
  md.ego_ids     = f->drone_ids;
  md.body        = c->d_body;
  md.n_body      = c->n_body;
  md.t0          = f->t0;
  md.period      = f->period;
  md.start_offset = f->replan_start_offset;
  md.hover       = f->hover_inout;
  md.now         = p->d_fl_now;
  md.t_start     = p->d_fl_tstart;
  md.pva         = p->d_fl_pva;
  md.poses       = c->d_poses;
  md.stamps      = c->d_stamps;
  md.n_reset     = c->tune_i(SOGM_TUNE_FLIGHT_RESET);
  md.n_bits      = c->tune_i(SOGM_TUNE_FLIGHT_BITS);
  md.n_marks     = c->tune_i(SOGM_TUNE_FLIGHT_MARKS);
  md.n_splat     = c->tune_i(SOGM_TUNE_FLIGHT_SPLAT);
  md.n_head_wgs  = c->tune_i(SOGM_TUNE_FLIGHT_HEADS) < p->fl_wgs[3] / 2 ? c->tune_i(SOGM_TUNE_FLIGHT_HEADS)
                                                                        : (p->fl_wgs[3] / 2 > 0 ? p->fl_wgs[3] / 2 : 1);
  // the urgent lane (sogm_planner.hpp): a few heads and a share of the workers, none with flight_urgent = 0
  fl.n_urgent    = c->tune_i(SOGM_TUNE_FLIGHT_URGENT) < A ? c->tune_i(SOGM_TUNE_FLIGHT_URGENT) : A - 1;
  if (fl.n_urgent < 0) fl.n_urgent = 0;
  md.n_uhead_wgs = 0;
  md.n_uwork_wgs = 0;
  {
    const int fine = c->tune_i(SOGM_TUNE_FLIGHT_URGENT_FINE);
    auto      cut  = [fine](int n, int hi) { return n * fine < hi ? n * fine : hi; };
    md.un_reset = cut(md.n_reset, 256);
    md.un_bits  = cut(md.n_bits, 256) > 2 * md.n_bits ? 2 * md.n_bits : cut(md.n_bits, 256);  // (bits tickets are short already)
    md.un_marks = cut(md.n_marks, 256);
    md.un_splat = cut(md.n_splat, 64);
  }
  fl.gate_pace_ticks = (int)(c->tune[SOGM_TUNE_FLIGHT_GATE_PACE_US] * 100.0);
  p->fl_epoch = p->fl_epoch >= 0x7ffffff0 ? 1 : p->fl_epoch + 1;  // (never 0: the zeroed word)
  fl.epoch    = p->fl_epoch;
  fl.n_splat  = md.n_splat;
  fl.un_splat = md.un_splat;
  if (fl.n_urgent > 0) {
    const int workers = p->fl_wgs[3] - md.n_head_wgs;
    md.n_uhead_wgs = md.n_head_wgs >= 8 ? 4 : md.n_head_wgs / 2;
    md.n_uwork_wgs = c->tune_i(SOGM_TUNE_FLIGHT_URGENT_WAVES) < workers ? c->tune_i(SOGM_TUNE_FLIGHT_URGENT_WAVES) : workers;
    if (md.n_uhead_wgs < 1 || md.n_uwork_wgs < 1) fl.n_urgent = 0, md.n_uhead_wgs = 0, md.n_uwork_wgs = 0;
  }
  md.n_admit     = c->tune_i(SOGM_TUNE_FLIGHT_ADMIT);
  md.pace_ticks  = (int)(c->tune[SOGM_TUNE_FLIGHT_PACE_US] * 100.0);
  md.agent_bytes = agent_bytes;
  md.reset_stat  = c->d_reset_stat;
  const MapView mv = view_of(c);
  // frames + control block, in stream order on the caller's stream
  SOGM_HIP_CHECK(hipMemcpyAsync(p->d_fl_worlds, p->h_fl_worlds, sizeof(FlightWorld) * (size_t)f->n_ticks, hipMemcpyHostToDevice, main));
  const int       ring    = p->fl.ring_mask + 1;
  const int       n_words = FL_HDR + 4 * ring + 4 * FL_WQ_SLOTS + 3 * FLIGHT_MAX_TICKS + 4 * A;  // (the parked lists: k_flight_seed)
  const long long n_log   = (long long)f->n_ticks * A * (long long)(sizeof(SogmTrajRecord) / sizeof(int));
  hipLaunchKernelGGL(k_flight_reset, dim3(256), dim3(256), 0, main, fl, n_words, p->aw.verdict, p->fl.acc,
                     reinterpret_cast<int *>(f->log_records), n_log);
  SOGM_HIP_CHECK(hipGetLastError());
  SOGM_HIP_CHECK(hipMemsetAsync(f->log_ok, 0, sizeof(int32_t) * (size_t)f->n_ticks * A, main));
  hipLaunchKernelGGL(k_flight_seed, dim3((A + 255) / 256), dim3(256), 0, main, fl);
  SOGM_HIP_CHECK(hipGetLastError());
  SOGM_HIP_CHECK(hipEventRecord(p->fl_ev_in, main));
  for (int k = 0; k < 4; ++k) SOGM_HIP_CHECK(hipStreamWaitEvent(p->fl_stream[k], p->fl_ev_in, 0));
  FL_TRACE("waits queued");
  // QP first, then search (each wants whole CUs), then the one-wave kernels: with masks the order is immaterial
  const int spec = c->tune_i(SOGM_TUNE_FLIGHT_SPEC) != 0 ? 1 : 0;
  if (sogm::launch_flight_qp(p->pp, p->qs, p->qw, p->qc, fl, p->fl_wgs[0], p->d_fl_pva, p->d_goal, p->d_polys, p->d_nfaces,
                             p->d_npoly, p->d_cpts, p->d_status, p->d_iters, p->fl_stream[0]) ||
      sogm::launch_flight_search(mv, p->ap, p->pp.corridor_tau, astar_ws(p), fl, p->fl_wgs[1], p->d_fl_pva, f->goals,
                                 p->d_fl_tstart, p->d_ret, p->d_route, p->d_route_len, p->route_cap, p->d_stats, spec,
                                 p->fl_stream[1])) {
    sogm::set_error("sogm_flight_run: launch", hipGetLastError());
    (void)hipDeviceSynchronize();
    return SOGM_ERR_HIP;
  }
  FL_TRACE("qp + search launched");
  sogm::FlightLightDev ld{};
  ld.start_pva   = p->d_fl_pva;
  ld.t_start     = p->d_fl_tstart;
  ld.route       = p->d_route;
  ld.route_len   = p->d_route_len;
  ld.route_cap   = p->route_cap;
  ld.out_polys   = p->d_polys;
  ld.out_nfaces  = p->d_nfaces;
  ld.out_npoly   = p->d_npoly;
  ld.out_goal    = p->d_goal;
  ld.fin         = sogm::FinishArgs{p->pp.corridor_tau, p->d_ret, p->d_npoly, p->d_status, p->d_cpts, nullptr, f->n_total,
                                    f->drone_ids, p->d_fl_now, p->d_fl_tstart, f->drone_ids, nullptr, nullptr, p->d_safe,
                                    p->cw.counters, f->own_inout, nullptr};
  ld.tables      = f->tables;
  ld.n_total     = f->n_total;
  ld.agent0      = f->agent0;
  ld.log_records = f->log_records;
  ld.log_ok      = f->log_ok;
  if (sogm::launch_flight_light(mv, p->pp, p->cw, fl, ld, p->fl_wgs[2], p->fl_stream[2]) ||
      sogm::launch_flight_map(c->geom, fl, md, p->fl_wgs[3], p->fl_stream[3])) {
    sogm::set_error("sogm_flight_run: launch", hipGetLastError());
    (void)hipDeviceSynchronize();
    return SOGM_ERR_HIP;
  }
  FL_TRACE("light + map launched");
  if (xchg) {
    // The exchange behind the call: per tick [wait: every local agent has finished it] -> in-place all-gather of this rank's
    // rows of its table version -> [mark the version complete, queue the overlays parked at its gate], all queued now, on the
    // context's exchange stream, behind the control block's reset.  Every rank queues the same n_ticks collectives; a rank
    // whose flight fails still runs them (the waiting kernel gives up on the error word), so no peer hangs in a collective.
    hipStream_t xs = nullptr;
    if (int rc = sogm::exchange_stream(c, &xs)) return rc;
    SOGM_HIP_CHECK(hipStreamWaitEvent(xs, p->fl_ev_in, 0));
    int rc_x = SOGM_OK;
    for (int i = 0; i < f->n_ticks; ++i) {
      const int       k   = f->first_tick + i;
      SogmTrajRecord *tab = f->tables + (size_t)(k & 3) * f->n_total;
      hipLaunchKernelGGL(k_flight_xwait, dim3(1), dim3(64), 0, xs, fl, i);
      if (rc_x == SOGM_OK)  // (after a failed collective the remaining ones are not attempted: the communicator is broken)
        rc_x = sogm::exchange_allgather_raw(c, f->nccl_comm, tab + f->agent0, tab, sizeof(SogmTrajRecord) * (size_t)A);
      hipLaunchKernelGGL(k_flight_xsignal, dim3(1), dim3(64), 0, xs, fl, i);
      if (i == 0) FL_TRACE("first collective queued");
    }
    FL_TRACE("collectives queued");
    SOGM_HIP_CHECK(hipGetLastError());
    if (int rc = sogm::exchange_mark_pending(c)) return rc;
    if (int rc = sogm::join_exchange(c, main)) return rc;  // the call's last versions are complete when `stream` goes on
    if (rc_x != SOGM_OK) {
      (void)hipDeviceSynchronize();
      return rc_x;
    }
  }
  for (int k = 0; k < 4; ++k) {
    SOGM_HIP_CHECK(hipEventRecord(p->fl_ev_done[k], p->fl_stream[k]));
    SOGM_HIP_CHECK(hipStreamWaitEvent(main, p->fl_ev_done[k], 0));
  }
  hipLaunchKernelGGL(k_flight_report, dim3(1), dim3(1), 0, main, (const int *)p->fl.hdr, p->h_flow_fail);
  SOGM_HIP_CHECK(hipGetLastError());
  c->updated             = 1;
  c->cur_prestamped      = 0;
  c->records_final_valid = 0;
  c->n_stamps += f->n_ticks;
  return SOGM_OK;
}

}  // extern "C"
__global__ void k_flight_touch(int *word) {
  if (threadIdx.x == 0 && word) atomicAdd(word, 0);
}
extern "C" {
// Creates what the first sogm_flight_run would create — control block, the four masked streams, the exchange stream — and
// runs one empty kernel on each, so that their hardware queues EXIST before any flight is in the air, and allocates what the
// first sogm_flight_run would allocate (crop lists for frames of up to max_cloud_points points, the stamp's scratch): that
// first call otherwise synchronises the DEVICE while it grows them.  A host that flies several planners in one process (two
// ranks on one device: tests) calls this for each of them before the first flight — the second rank's first call would
// otherwise wait for the first rank's flight to end, which waits for the second rank's rows.  Optional otherwise.  Synchronises.
int sogm_flight_prepare(sogm_planner *p, int max_cloud_points) {
  if (!p || max_cloud_points < 0) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(p->map->device));
  if (int rc = flight_setup(p)) return rc;
  {  // everything a first sogm_flight_run would allocate — and synchronise the DEVICE for: the per-agent crop lists for
     // frames of up to max_cloud_points points (blocks of 256), the stamp's scratch
    SogmWorld big{};
    big.block_points = 256;
    big.n_points     = max_cloud_points;
    big.n_blocks     = (max_cloud_points + 255) / 256;
    sogm::CloudBlocks cb{};
    if (int rc = sogm::world_blocks(p->map, &big, &cb)) return rc;
    sogm::PrestampDev tmp{};
    if (p->map->sparse) {
      if (int rc = sogm::prestamp_buffers(p->map, &tmp)) return rc;
      // the current grid's mark log (its creation ends with a null-stream memset and synchronisation, which waits for every
      // BLOCKING stream of the process — masked streams are — i.e. for another planner's flight) and the one dense clear that
      // puts the grid under its log
      sogm_ctx *c    = p->map;
      const int slot = sogm::cur_slot(c);
      if (!c->tracked[slot] || !sogm::mark_log(c, slot).entries)
        if (int rc = sogm::launch_clear(c, nullptr, c->d_grid, false)) return rc;
    }
  }
  hipStream_t xs = nullptr;
  if (int rc = sogm::exchange_stream(p->map, &xs)) return rc;
  for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_flight_touch, dim3(1), dim3(64), 0, p->fl_stream[k], (int *)nullptr);
  hipLaunchKernelGGL(k_flight_touch, dim3(1), dim3(64), 0, xs, (int *)nullptr);
  SOGM_HIP_CHECK(hipGetLastError());
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  return SOGM_OK;
}

// diagnostics (tools/ only): every agent-tick's stamps of the last flight, [n_ticks][A][FL_TS] ticks of 10 ns
int sogm_debug_flight_times(sogm_planner *p, long long *out_host, int n_ticks) {
  if (!p || !p->d_fl || !out_host || n_ticks < 1 || n_ticks > FLIGHT_MAX_TICKS) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  SOGM_HIP_CHECK(hipMemcpy(out_host, p->fl.ts_log, sizeof(long long) * FL_TS * (size_t)p->map->n_agents * n_ticks, hipMemcpyDeviceToHost));
  return SOGM_OK;
}

// diagnostics (tools/ only): the control block of the last flight as it stands — out_words = [FL_COUNTERS header counters]
// [FLIGHT_MAX_TICKS tick_done][FLIGHT_MAX_TICKS parked_n][A tick_of][A seg_done][A stage][A urgent], out_ts = [A][FL_TS]
// stamps of every agent's current tick.  What a stalled flight looked like when it was aborted.
int sogm_debug_flight_dump(sogm_planner *p, int32_t *out_words, int cap_words, long long *out_ts) {
  if (!p || !p->d_fl || !out_words) return SOGM_ERR_INVALID_ARG;
  const int A = p->map->n_agents;
  if (cap_words < FL_COUNTERS + 2 * FLIGHT_MAX_TICKS + 4 * A) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(p->map->device));
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  for (int i = 0; i < FL_COUNTERS; ++i)
    SOGM_HIP_CHECK(hipMemcpy(out_words + i, p->fl.hdr + (size_t)i * FL_STRIDE, sizeof(int), hipMemcpyDeviceToHost));
  int32_t *o = out_words + FL_COUNTERS;
  SOGM_HIP_CHECK(hipMemcpy(o, p->fl.tick_done, sizeof(int) * FLIGHT_MAX_TICKS, hipMemcpyDeviceToHost));
  o += FLIGHT_MAX_TICKS;
  SOGM_HIP_CHECK(hipMemcpy(o, p->fl.parked_n, sizeof(int) * FLIGHT_MAX_TICKS, hipMemcpyDeviceToHost));
  o += FLIGHT_MAX_TICKS;
  SOGM_HIP_CHECK(hipMemcpy(o, p->fl.tick_of, sizeof(int) * (size_t)A, hipMemcpyDeviceToHost));
  o += A;
  SOGM_HIP_CHECK(hipMemcpy(o, p->fl.seg_done, sizeof(int) * (size_t)A, hipMemcpyDeviceToHost));
  o += A;
  SOGM_HIP_CHECK(hipMemcpy(o, p->fl.stage, sizeof(int) * (size_t)A, hipMemcpyDeviceToHost));
  o += A;
  SOGM_HIP_CHECK(hipMemcpy(o, p->fl.urgent, sizeof(int) * (size_t)A, hipMemcpyDeviceToHost));
  if (out_ts) SOGM_HIP_CHECK(hipMemcpy(out_ts, p->fl.ts, sizeof(long long) * FL_TS * (size_t)A, hipMemcpyDeviceToHost));
  return SOGM_OK;
}

// diagnostics (tools/, tests/): when every workgroup of the last flight's four kernels started — out_host [8][4096]: [0..3] wall-clock
// ticks (100 MHz; 0 = the workgroup never ran), [4..7] where it started (HW_ID | XCC_ID << 32), out_wgs_host[4] the launched counts (QP, search, corridor + finish, map)
int sogm_debug_flight_wg_starts(sogm_planner *p, long long *out_host, int32_t *out_wgs_host) {
  if (!p || !p->d_fl || !out_host) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(p->map->device));
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  SOGM_HIP_CHECK(hipMemcpy(out_host, p->fl.wg_start, sizeof(long long) * 8 * FL_WG_LOG, hipMemcpyDeviceToHost));
  if (out_wgs_host)
    for (int k = 0; k < 4; ++k) out_wgs_host[k] = p->fl_wgs[k];
  return SOGM_OK;
}

int sogm_flight_stats(sogm_planner *p, double *out_ms, int32_t *out_hdr) {
  if (!p || !p->d_fl) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(p->map->device));
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  const int A = p->map->n_agents;
  if (out_ms) {
    long long *tmp = new (std::nothrow) long long[8 * (size_t)A];
    if (!tmp) return SOGM_ERR_INVALID_ARG;
    const hipError_t e = hipMemcpy(tmp, p->fl.acc, sizeof(long long) * 8 * (size_t)A, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8 * A; ++i) out_ms[i] = (i % 8) == 7 ? (double)tmp[i] : (double)tmp[i] / 1.0e5;  // 100 MHz -> ms
    delete[] tmp;
    SOGM_HIP_CHECK(e);
  }
  if (out_hdr) {  // the counters, compacted (the device keeps them FL_STRIDE words apart)
    int *tmp = new (std::nothrow) int[FL_HDR];
    if (!tmp) return SOGM_ERR_INVALID_ARG;
    const hipError_t e = hipMemcpy(tmp, p->fl.hdr, sizeof(int) * FL_HDR, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i) out_hdr[i] = i < 15 ? tmp[(size_t)i * FL_STRIDE] : 0;  // (the urgent lane's four: not reported)
    delete[] tmp;
    SOGM_HIP_CHECK(e);
    {  // [15]: workgroups of the four kernels that were NOT resident from the start (first instruction > 1 ms after the
       // flight's first workgroup): must be 0 — the liveness argument (flight_layout) and tests/test_flight_gpu.py hold it
      long long *ws = new (std::nothrow) long long[4 * FL_WG_LOG];
      if (!ws) return SOGM_ERR_INVALID_ARG;
      const hipError_t e2 = hipMemcpy(ws, p->fl.wg_start, sizeof(long long) * 4 * FL_WG_LOG, hipMemcpyDeviceToHost);
      int late = 0;
      for (int k = 0; k < 4; ++k) {  // per kernel: against ITS first workgroup (the four launches may start a millisecond apart)
        long long first = 0;
        for (int b = 0; b < p->fl_wgs[k] && b < FL_WG_LOG; ++b) {
          const long long v = ws[(size_t)k * FL_WG_LOG + b];
          if (v > 0 && (first == 0 || v < first)) first = v;
        }
        for (int b = 0; b < p->fl_wgs[k] && b < FL_WG_LOG; ++b) {
          const long long v = ws[(size_t)k * FL_WG_LOG + b];
          if (v == 0 || v - first > 100000) ++late;
        }
      }
      delete[] ws;
      SOGM_HIP_CHECK(e2);
      out_hdr[15] = late;
    }
    unsigned long long prof[16];  // [16..31]: wave time by activity in units of 10 us (0-8), descriptor counts (9-15)
    SOGM_HIP_CHECK(hipMemcpy(prof, p->fl.prof, sizeof(prof), hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; ++i) out_hdr[16 + i] = (int32_t)(i < 9 ? prof[i] / 1000ull : prof[i]);
  }
  return SOGM_OK;
}

int sogm_replan(sogm_planner *p, const double *start_pva, const double *goal,
                const double *t_start, const int32_t *drone_ids, SogmTrajRecord *out_records,
                int32_t *out_ok, void *stream) {
  if (!p || !start_pva || !goal || !t_start || !drone_ids || !out_records || !out_ok)
    return SOGM_ERR_INVALID_ARG;
  if (!p->map->updated) return SOGM_ERR_STATE;
  sogm_ctx *c = p->map;
  // publication (sogm_planner_set_publish): the finishing kernel writes every agent's record into next_table while
  // other agents' deconfliction still reads the swarm table, and into own_records while out_records is written
  if (p->pub_table && p->swarm) {  // byte ranges, not pointers: a rank's slice INSIDE the all-records buffer overlaps too
    const char *a0 = (const char *)p->pub_table, *a1 = a0 + sizeof(SogmTrajRecord) * (size_t)c->n_agents;
    const char *b0 = (const char *)p->swarm, *b1 = b0 + sizeof(SogmTrajRecord) * (size_t)p->n_swarm;
    if (a0 < b1 && b0 < a1) {
      sogm::set_error_text("sogm_replan: sogm_planner_set_publish's next_table overlaps the table given to sogm_planner_set_swarm");
      return SOGM_ERR_INVALID_ARG;
    }
  }
  if (p->pub_own) {
    const char *a0 = (const char *)p->pub_own, *a1 = a0 + sizeof(SogmTrajRecord) * (size_t)c->n_agents;
    const char *b0 = (const char *)out_records, *b1 = b0 + sizeof(SogmTrajRecord) * (size_t)c->n_agents;
    if (a0 < b1 && b0 < a1) {
      sogm::set_error_text("sogm_replan: out_records overlaps sogm_planner_set_publish's own_records");
      return SOGM_ERR_INVALID_ARG;
    }
  }
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  // the in-place pre-clear (mode 1) needs the grouped path's "last reader of the SOGM" events
  const bool use_flow = p->flow && c->overlap != 1;
  if (int jr = sogm::join_prestamp(c, (hipStream_t)stream)) return jr;  // a pre-stamp nobody has waited for yet
  const int  rc = use_flow ? replan_flow(p, start_pva, goal, t_start, drone_ids, out_records, out_ok, stream)
                           : replan_impl(p, start_pva, goal, t_start, drone_ids, out_records, out_ok, stream);
  if (rc != SOGM_OK) {
    // A launch failed half-way: group / side streams may hold work the caller's stream was never joined to, and
    // a pre-clear may or may not have been issued.  Drain everything and forget the pre-clear (the next update
    // clears its grid itself); an in-place clear (mode 1) may already have eaten part of the map.
    (void)hipDeviceSynchronize();
    if (c->precleared && c->overlap == 1) c->updated = 0;
    if (c->overlap < 2) c->precleared = 0;  // modes 2 / 3: clears already queued stay valid (events recorded)
  }
  return rc;
}
}
