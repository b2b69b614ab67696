// sogm_planner.hpp — planner context + stage launchers shared by the planner translation units.
#pragma once

#include "sogm_device.hpp"

namespace sogm {

// Per-agent A* scratch in HBM (L2-resident while a search runs).
struct AstarWorkspace {
  char  *pool;         // [A][pool_stride] bytes, Node records
  size_t pool_stride;  // bytes per agent
  void  *hkeys;        // [A][hash_cap] 64-bit slots {x, y, z, t, node id}
  int    hash_cap;     // power of two >= 2 * allocate_num
  long long *dbg;      // [A][8] per-phase wall_clock64 ticks (diagnostics)
  // speculative second attempt (dataflow replan): a second set of pools / hash tables ([A..2A)) and the verdict of
  // the first attempt per agent (0 pending, 1 found a path, 2 NO_PATH); null = not available
  int *verdict;
};
int astar_pool_max();

// Per-(agent, segment) corridor scratch in HBM.
struct CorridorWorkspace {
  double  *pc;          // [A*P][cap][3] obstacle points
  double  *fpc;         // [A*P][cap][3] points in the ellipsoid frame
  double  *tang;        // [A*P][cap][4] tangent planes
  double  *distr;       // [A*P][cap]
  double  *polys;       // [A*P][max_faces][4] shrunk polytopes
  int32_t *seg_nfaces;  // [A*P]
  int32_t *seg_state;   // [A*P] 1 valid, 0 invalid, -2 no segment, -3 capacity exceeded
  int32_t *seg_npts;    // [A*P]
  long long *seg_dbg;   // [A*P][16] diagnostics: counts and wall_clock64 ticks (100 MHz) per phase
  unsigned long long *counters;  // [SOGM_CNT_N] cumulative outcome / capacity counters (sogm_planner_counters)
};

// Device-side control block of the dataflow replan (one per planner, reset at the start of every sogm_replan):
// kernels of one tick hand agents to each other through ready lists instead of stream order.
//   hdr[FLOW_*] counters; seg_done[A] finished segment slots per agent; a_ready[A] agents in A* completion order;
//   q_ready[A] agents in corridor completion order (entries are -1 until published).
//   f_ready[A] agents in QP completion order; p_ready[A] agents whose record is published (k_finish_flow done), the
//   pre-stamp's input; stage[A] per-agent progress counter of the pre-stamp (0 at the start of a replan).
enum { FLOW_A_RESIDENT = 0, FLOW_A_READY_N = 1, FLOW_C_TICKET = 2, FLOW_Q_READY_N = 3, FLOW_Q_TICKET = 4,
       FLOW_ERR = 5, FLOW_F_READY_N = 6, FLOW_F_TICKET = 7, FLOW_P_READY_N = 8, FLOW_P_TICKET = 9,
       FLOW_Q_RESIDENT = 10 /* QP workgroups that have started */, FLOW_HDR = 11 };
#define FLOW_PS_DONE (1 << 20)  // stage[agent] once the agent's pre-stamp is complete (its last marks ticket sets it)
#define FLOW_TIMEOUT_TICKS 300000000LL  // 3 s of the 100 MHz wall clock: a stuck tick fails instead of hanging
// One polling interval of the waiting loops of the dataflow replan.  A poll is a device-scope load that goes to the
// memory side (the L2s are per XCD) while the SOGM clear streams beside it; the stages waited for take hundreds of
// microseconds, so the waiting waves look every ~14 us (SOGM_POLL_PAUSES x s_sleep 127 = 4 x 3.4 us).
#ifndef SOGM_POLL_PAUSES
#define SOGM_POLL_PAUSES 4
#endif
#ifdef __HIPCC__
__device__ inline void flow_pause() {
#pragma unroll
  for (int i = 0; i < SOGM_POLL_PAUSES; ++i) __builtin_amdgcn_s_sleep(127);
}
#endif
struct FlowCtl {
  int *hdr;       // [FLOW_HDR]
  int *seg_done;  // [A]
  int *a_ready;   // [A]
  int *q_ready;   // [A]
  int *f_ready;   // [A] agents in QP completion order
  long long *ts;  // [A][8] wall_clock64 stamps (100 MHz): 0 A* start, 1 A* done, 2 first corridor item taken,
                  //        3 corridors final, 4 QP start, 5 QP done, 6 finished, 7 A* workgroup resident
                  //        (diagnostics, always written)
  int *p_ready;   // [A] agents in publication order (null: nobody consumes it)
  int *stage;     // [A]
  const int *map_ready;  // [A] update flow: the agent's map is complete when this holds map_epoch (null: it is already)
  int        map_epoch;
  // the control block's reset runs on the corridor stream, early (under the map update); the searches wait for its
  // generation word in their prologue and zero their agent's outputs there (k_astar; null: nothing to wait for)
  const int *reset_gen;
  int        reset_epoch;
  int32_t   *out_ok;       // [A] this replan's outputs, zeroed per agent by its first search workgroup
  int       *out_records;  // [A][rec_words]
  int        rec_words;
};
#ifdef __HIPCC__
// hand-over primitives of the persistent kernels: a ticket per wave, a bounded wait for a published slot
__device__ inline int flow_wait_slot(int *slot, int *err) {
  const long long t0 = wall_clock64();
  for (;;) {
    // relaxed agent-scope poll (an sc1 load); ONE acquire fence once the value is there (an acquire per poll would
    // invalidate the CU's L1 every microsecond)
    const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (v >= 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      return v;
    }
    flow_pause();
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0)
      return -1;
    if (wall_clock64() - t0 > FLOW_TIMEOUT_TICKS) {
      if ((threadIdx.x & 63) == 0) atomicExch(err, 2);
      return -1;
    }
  }
}
__device__ inline int flow_ticket(int *counter) {  // one ticket per wave, uniform
  int k = 0;
  if ((threadIdx.x & 63) == 0) k = atomicAdd(counter, 1);
  return __builtin_amdgcn_readfirstlane(k);
}
#endif
// ---------------------------------------------------------------------------------------------------------------
// Flight (sogm_flight_run): n_ticks replan ticks of every agent in ONE set of launches, every agent on its own clock.
// The reference's drones replan asynchronously, each reading whatever trajectories arrived last
// (plan_manager/src/plan_manager.cpp:92-233, traj_coordinator/src/particles.cpp:179-190).  Here the RESULTS are
// fixed by a staleness rule — agent a's tick k reads its own record of tick k - 1 and the neighbours' records as of
// their tick k - 2 (table ver(k - 2)), and may start once every agent has finished tick k - 2 — and the SCHEDULE is free:
// an agent whose chain is done goes straight on to its next tick while a straggler still solves its QP.
// Four persistent kernels, each on a stream with its own compute units (hipExtStreamCreateWithCUMask; every mask balanced over
// the shader engines it touches and every launch exactly as large as its mask holds — flight_layout in sogm_planner.hip: all
// workgroups resident from the first microsecond, which is also what lets a flight survive the hardware scheduler's queue
// save / restore; no residency gates, no dispatch-order assumptions, four hardware queues):
//   k_flight_map     a few admitting waves + role-less one-wave workgroups over ONE work queue of ready map work (a descriptor
//                    is pushed when its prerequisites are complete, so no worker sits waiting on another).  The admitting
//                    waves take agents in the order their previous tick finished and let `flight_admit` maps be under
//                    construction at once: agents leave the map stage one after the other and stay spread over the stages
//                    — a swarm whose agents all share every stage equally moves in step, and then every kernel's compute
//                    units idle while another kernel's are busy.  Head (start state from the own record, cull of
//                    cylinders and cloud blocks of the tick's SogmWorld frame) -> sparse reset of the agent's grid through its
//                    mark log + occupancy bits -> marks -> [gate: every agent has finished tick k - 2] -> neighbour overlay
//                    (the only phase that reads table ver(k - 2)) -> s_ring
//   k_flight_search  one workgroup per (agent, attempt) ticket: hybrid A* (both attempts side by side) -> 16 corridor descriptors
//   k_flight_light   role-less one-wave workgroups over ONE work queue: corridor segments (-> q_ring) and finish items
//                    (deconfliction, record, publication, tick accounting, the agent's next map head descriptor)
//   k_flight_qp      one workgroup per CU: the Bezier QP -> a finish descriptor
// Hand-over to the search and QP kernels: rings in HBM indexed by a monotonic position; a slot holds
// ((position / R + 1) << 16) | agent, so a reader with ticket t takes its item when the slot's generation is t / R + 1
// (R >= 2 A: an agent has one item in flight).  To the one-wave kernels: work queues (below).
// Per-agent buffers (start state, route, polytopes, control points, grid, mark log) are single: an agent's chain is
// strictly sequential.  Swarm tables: a ring of four versions, ver(j) at slot j & 3.
// Every counter of the header sits 4 KiB from the next: idle waves poll words of it, and with all of them in one 128-byte
// line ~900 polling waves saturated that line's memory channel — every claim of every kernel queued behind the polls.
#define FL_STRIDE 1024
enum { FL_S_READY = 0 * FL_STRIDE, FL_S_TICKET = 1 * FL_STRIDE, FL_Q_READY = 2 * FL_STRIDE, FL_Q_TICKET = 3 * FL_STRIDE,
       FL_ERR = 4 * FL_STRIDE, FL_FINISHED = 5 * FL_STRIDE /* agent-ticks finished */,
       FL_MW_TAIL = 6 * FL_STRIDE, FL_MW_HEAD = 7 * FL_STRIDE,   // work queue of the map kernel: descriptors pushed / tickets taken
       FL_LW_TAIL = 8 * FL_STRIDE, FL_LW_HEAD = 9 * FL_STRIDE,   // work queue of the corridor + finish kernel
       FL_M_READY = 10 * FL_STRIDE, FL_M_TICKET = 11 * FL_STRIDE,  // map heads: agents whose previous tick is finished / admitted
       FL_MAPS_DONE = 12 * FL_STRIDE,                               // maps completed (admission control)
       FL_ADMITTED = 13 * FL_STRIDE,                                // heads admitted so far (they are admitted in ticket order)
       FL_PACE_CLOCK = 14 * FL_STRIDE,                              // (two words) wall clock of the last admission
       FL_U_READY = 15 * FL_STRIDE, FL_U_TICKET = 16 * FL_STRIDE,   // urgent lane (below): heads published / taken
       FL_UW_TAIL = 17 * FL_STRIDE, FL_UW_HEAD = 18 * FL_STRIDE,    // urgent lane: map descriptors pushed / tickets taken
       FL_END = 19 * FL_STRIDE,                                     // the epoch of the call whose last agent-tick is finished
       FL_COUNTERS = 20, FL_HDR = 20 * FL_STRIDE };
// The urgent lane of the map kernel.  The flight's rate is the rate of its SLOWEST agent's own chain (tools/diag_flight.py:
// the critical path follows one agent with long corridors / QPs for many ticks in a row), and that agent — always behind,
// never gated — queued like everybody else: behind a burst of leaders the gate had just released (up to 1.3 ms in the
// in-order admission, 0.4 ms for a free head wave) and then shared the map workers with ~24 other maps (1.0 ms for a map
// that takes 0.3 by itself).  The leaders have slack by definition, the laggards have none: an agent that finishes tick k
// among the last `flight_urgent` of the swarm builds the map of its tick k + 1 through a lane of its own — `u_ring` ->
// urgent heads (no admission order, no pace, no window) -> work queue `uw`, which the workers look at before they take
// plain work and while they wait for it, in finer tickets.  The cells, records and logs do not depend on the schedule
// (the staleness rule fixes every input).
// Work queues (map kernel, corridor + finish kernel): ONE FIFO of ready work per kernel.  A producer reserves positions with
// one atomicAdd on the tail and stores a descriptor per position, tagged with the position's generation; a consumer takes a
// ticket with one atomicAdd on the head and waits for ITS position (idle waves therefore poll distinct words).  Every
// published descriptor is taken by the lowest waiting ticket, whatever its kind: no wave ever waits for work that depends
// on work nobody is free to do.  (Two earlier forms: all tickets of an item handed out in order and waiting for each
// other — 512 waves / 61 tickets = 8 maps in flight; a compare-and-swap claim per phase queue — hundreds of waves
// retrying on one counter, the map stage took 5-28 ms per agent and got SLOWER with more waves or tickets.)
// descriptor: kind << 28 | sub << 16 | agent
enum { WK_MAP_HEAD = 0, WK_MAP_RESET = 1, WK_MAP_BITS = 2, WK_MAP_MARKS = 3, WK_MAP_SPLAT = 4, WK_CORRIDOR = 5, WK_FINISH = 6 };
#define FL_TS 16            // stamps per agent-tick (FlightCtl::ts)
#define FL_WQ_SLOTS 131072  // per queue (a tick of 128 agents pushes 8-25 k map descriptors; at most two ticks are in flight)
#define FLIGHT_MAX_TICKS 64
struct FlightWorld {  // one SogmWorld frame as the kernels read it
  const float        *cloud, *bounds;
  const SogmCylinder *cyl;
  int                 n_points, n_blocks, block_points, n_cyl;
};
struct FlightCtl {
  int *hdr;                                         // [FL_HDR]
  int *s_ring, *q_ring, *m_ring, *u_ring;           // [ring_mask + 1] each: maps ready for the search, corridors final for the QP,
                                                    // agents whose previous tick is finished (map heads; u_ring: the urgent ones)
  unsigned long long *mw, *lw, *uw;                 // [FL_WQ_SLOTS] work queues of the map / the corridor + finish kernel / the
                                                    // map kernel's urgent lane
  int  ring_mask;
  int *urgent;      // [A] 1: the agent's current tick goes through the urgent lane
  int  n_urgent;    // an agent among the last n_urgent finishers of a tick is urgent in its next one (0: no urgent lane)
  int  n_splat, un_splat;  // overlay tickets of a map in the plain / the urgent lane (the finish that opens a gate queues them)
  int  epoch;              // this call's number (never 0, never repeated while the planner lives): the waves whose work has
                           // no known count leave when hdr[FL_END] holds it — a word the call's last finish stores, compared for
                           // EQUALITY, so that a value left by an earlier call can end nothing (the counters are zeroed by a
                           // kernel before the flight's kernels start, but a poll is a load, and "FINISHED >= all" would
                           // hold for the previous call's final count)
  int  gate_pace_ticks;    // 100 MHz ticks between two overlays that a gate releases (they reach the search and the corridors
                           // one after the other instead of as a burst)
  int *tick_done;   // [FLIGHT_MAX_TICKS] agents that have finished tick first_tick + i
  int *parked_n;    // [FLIGHT_MAX_TICKS] maps of tick first_tick + i whose overlay is parked at the gate "tick i - 2 is complete" ...
  int *parked;      // [FLIGHT_MAX_TICKS][A] ... the agents (-1 empty, -2 released)
  int *xready;      // [FLIGHT_MAX_TICKS] several ranks with the exchange behind the call (SogmFlight::nccl_comm): == epoch once the
                    // all-gather of table ver(first_tick + i) — every rank's rows — has completed here; null: one process owns
                    // every row.  The gate of tick k's overlay is then xready[k - 2] instead of tick_done[k - 2] (which the
                    // collective itself waited for), and the parked overlays are released by the kernel behind the collective
                    // on the exchange stream (k_flight_xsignal) instead of by the finish that completes the tick.
  int *tick_of;     // [A] the tick the agent is in (absolute index)
  int *seg_done;    // [A] cumulative corridor segment slots finished
  int *stage;       // [A] cumulative map tickets finished
  long long *ts;    // [A][FL_TS] stamps of the agent's current tick: 0 A* start, 1 A* done, 2 first corridor item, 3 corridors
                    //         final, 4 QP start, 5 QP done, 6 finished, 7 map item published, 8 map head start, 9 gate passed, 10 marks done, 11 map ready,
                    //         12 head done (reset / bits tickets queued), 13 grid reset and bits set (marks tickets queued),
                    //         14 overlay tickets queued (the gate "tick k - 2 is complete" lies between 10 and 14)
  long long *acc;   // [A][8] sums over the flight (100 MHz ticks): gate wait, map, search queue + A*, corridors, QP queue + QP,
                    //        finish, whole chain, ticks completed
  long long *ts_log;         // [FLIGHT_MAX_TICKS][A][FL_TS] every agent-tick's stamps (sogm_debug_flight_times)
  unsigned long long *prof;  // [16] wave time (100 MHz ticks) by activity, summed over the flight: 0 map workers idle (waiting
                             //      for a descriptor), 1 reset, 2 bits, 3 marks, 4 overlay, 5 heads (incl. their waits),
                             //      6 light waves idle, 7 corridor segments, 8 finish; 9.. descriptor counts of 1-4, 7, 8
  long long *wg_start;       // [8][FL_WG_LOG] ([4..7]: where, HW_ID | XCC_ID << 32) wall clock at which workgroup b of kernel k (0 QP, 1 search, 2 corridor + finish, 3 map)
                             //      executed its first instruction in this call (0: never) — the residency evidence of
                             //      sogm_debug_flight_wg_starts: a workgroup that starts late was NOT resident from the start
  int  n_agents, n_ticks, first_tick;
  int  lag;         // tick k reads the neighbours' records of tick k - lag: 2 (the flight's rule: the most overlap) or 1 (the
                    // reference's staleness — a record one broadcast old, particles.cpp:179-190; tuning key flight_neighbour_lag)
};
#define FL_WG_LOG 4096
#ifdef __HIPCC__
__device__ inline void fl_wg_started(const FlightCtl &fl, int kernel) {
  if (threadIdx.x == 0 && fl.wg_start && blockIdx.x < FL_WG_LOG) {
    fl.wg_start[(size_t)kernel * FL_WG_LOG + blockIdx.x] = wall_clock64();
    // where: HW_ID (wave / SIMD / CU / SH / SE) | XCC_ID << 32
    fl.wg_start[(size_t)(4 + kernel) * FL_WG_LOG + blockIdx.x] =
        (long long)(unsigned)__builtin_amdgcn_s_getreg(63492) | ((long long)(unsigned)__builtin_amdgcn_s_getreg(63508) << 32);
  }
}
__device__ inline void fl_publish(int *ring, int mask, int *ready_n, int agent) {  // one lane; the item's data is written
  __threadfence();
  const int r = atomicAdd(ready_n, 1);
  __hip_atomic_store(ring + (r & mask), (((r / (mask + 1)) + 1) << 16) | agent, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// the agent at ring position `pos` (wave-uniform; bounded wait; -1 = the flight failed)
__device__ inline int fl_wait_item(const int *ring, int mask, int pos, int *err) {
  const int       want = (pos / (mask + 1)) + 1;
  const long long t0   = wall_clock64();
  for (;;) {
    const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ring + (pos & mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if ((v >> 16) == want) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      return v & 0xFFFF;
    }
    flow_pause();
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) return -1;
    if (wall_clock64() - t0 > FLOW_TIMEOUT_TICKS) {
      if ((threadIdx.x & 63) == 0) atomicExch(err, 12);
      return -1;
    }
  }
}
// the gate of the staleness rule in front of tick kl's overlay (kl relative to first_tick): is table ver(kl - lag) complete?
__device__ inline bool fl_gate_open(const FlightCtl &fl, int kl) {
  if (kl < fl.lag) return true;  // (versions of an earlier call: complete before this call's kernels started)
  if (fl.xready) return __hip_atomic_load(&fl.xready[kl - fl.lag], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == fl.epoch;
  return __hip_atomic_load(&fl.tick_done[kl - fl.lag], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= fl.n_agents;
}
// table ver(kl - 2) has just become complete: queue the overlays of tick kl that were parked at the gate so far (ONE lane;
// the parking side re-checks the gate after it has written its slot: list + compare-and-swap on both sides)
__device__ inline void wq_push(unsigned long long *wq, int *tail, unsigned desc0, int count);
__device__ inline void fl_gate_release(const FlightCtl &fl, int kl) {
  const int A_ = fl.n_agents;
  __threadfence();
  int      *lst = fl.parked + (size_t)kl * A_;
  const int n   = __hip_atomic_load(&fl.parked_n[kl], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
  for (int i = 0; i < n && i < A_; ++i) {
    const int v = __hip_atomic_load(&lst[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= 0 && atomicCAS(&lst[i], v, -2) == v) {
      const bool u = fl.urgent[v] != 0;
      if (fl.gate_pace_ticks > 0 && i > 0) {
        const long long p0 = wall_clock64();
        while (wall_clock64() - p0 < fl.gate_pace_ticks) __builtin_amdgcn_s_sleep(32);
      }
      fl.ts[(size_t)v * FL_TS + 14] = wall_clock64();
      wq_push(u ? fl.uw : fl.mw, &fl.hdr[u ? FL_UW_TAIL : FL_MW_TAIL], ((unsigned)WK_MAP_SPLAT << 28) | (unsigned)v,
              u ? fl.un_splat : fl.n_splat);
    }
  }
}
// the same for the map kernel's lanes, whose item counts are not known in advance (an agent-tick goes through the plain or
// the urgent lane): -2 once the call's last agent-tick is finished (hdr[FL_END] == epoch); `timed` = false: no time limit of its own (the
// urgent lane may see no item for a whole flight; a stalled flight ends through the other waiters' limits and `err`)
__device__ inline int fl_wait_item_end(const int *ring, int mask, int pos, int *err, const int *end_word, int epoch, bool timed) {
  const int       want = (pos / (mask + 1)) + 1;
  const long long t0   = wall_clock64();
  for (;;) {
    const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ring + (pos & mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if ((v >> 16) == want) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      return v & 0xFFFF;
    }
    if (timed)
      flow_pause();
    else
      __builtin_amdgcn_s_sleep(127);  // (the few urgent heads poll every 3.4 us)
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(end_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == epoch) return -2;
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) return -1;
    if (timed && wall_clock64() - t0 > FLOW_TIMEOUT_TICKS) {
      if ((threadIdx.x & 63) == 0) atomicExch(err, 17);
      return -1;
    }
  }
}
// work queue, producer side (ONE lane): `count` descriptors desc0, desc0 + (1 << 16), ... (consecutive `sub` fields)
__device__ inline void wq_push(unsigned long long *wq, int *tail, unsigned desc0, int count) {
  __threadfence();
  const unsigned base = (unsigned)atomicAdd(tail, count);
  for (int i = 0; i < count; ++i) {
    const unsigned pos = base + (unsigned)i;
    const unsigned long long v = ((unsigned long long)(pos / FL_WQ_SLOTS + 1u) << 32) | (desc0 + ((unsigned)i << 16));
    __hip_atomic_store(wq + (pos % FL_WQ_SLOTS), v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// work queue, consumer side (wave-uniform): the descriptor at position `pos` (bounded wait; -1 = the flight failed)
__device__ inline int wq_take(const unsigned long long *wq, unsigned pos, int *err) {
  const unsigned  want = pos / FL_WQ_SLOTS + 1u;
  const long long t0   = wall_clock64();
  int             naps = 0;
  for (;;) {
    const unsigned long long v  = __hip_atomic_load(wq + (pos % FL_WQ_SLOTS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned           hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    if (hi == want) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      return (int)__builtin_amdgcn_readfirstlane((unsigned)v);
    }
    for (int i = 0; i <= (naps < 7 ? naps : 7); ++i) flow_pause();  // 14 us ... 110 us: an idle wave polls less and less
    ++naps;
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) return -1;
    if (wall_clock64() - t0 > FLOW_TIMEOUT_TICKS) {
      if ((threadIdx.x & 63) == 0) atomicExch(err, 15);
      return -1;
    }
  }
}
// the same with the end-of-flight exit (see fl_wait_item_end): -2 = every agent-tick is finished
__device__ inline int wq_take_end(const unsigned long long *wq, unsigned pos, int *err, const int *end_word, int epoch, bool timed) {
  const unsigned  want = pos / FL_WQ_SLOTS + 1u;
  const long long t0   = wall_clock64();
  int             naps = 0;
  for (;;) {
    const unsigned long long v  = __hip_atomic_load(wq + (pos % FL_WQ_SLOTS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned           hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    if (hi == want) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      return (int)__builtin_amdgcn_readfirstlane((unsigned)v);
    }
    if (timed) {
      for (int i = 0; i <= (naps < 7 ? naps : 7); ++i) flow_pause();  // 14 us ... 110 us: an idle wave polls less and less
    } else {
      __builtin_amdgcn_s_sleep(127);  // 3.4 us: the urgent lane exists for latency, and few waves poll it
    }
    ++naps;
    if ((naps & (timed ? 1 : 7)) == 0 &&
        __builtin_amdgcn_readfirstlane(__hip_atomic_load(end_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == epoch)
      return -2;
    if ((timed || (naps & 7) == 0) &&
        __builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0)
      return -1;
    if (timed && wall_clock64() - t0 > FLOW_TIMEOUT_TICKS) {
      if ((threadIdx.x & 63) == 0) atomicExch(err, 15);
      return -1;
    }
  }
}
// A worker of a kernel with a plain FIFO and a priority queue.  It holds a ticket of the plain queue, as wq_take's callers do,
// and looks at the priority queue first — before it takes its plain descriptor and while it waits for it — claiming a
// priority descriptor that is THERE with a compare-and-swap on that queue's head (never a ticket for one that is not: a
// worker must not be lost to the plain queue waiting for priority work; one try per look, so the waves do not spin on the
// counter).  Returns the descriptor (>= 0; `prio` says from which queue), -2 once every agent-tick of the flight is finished
// (the queues' item counts are not known in advance), -1 if the flight failed.  Wave-uniform.
struct WqWorker {
  bool     have_plain = false;
  unsigned plain_t    = 0;
  int      seen_ph    = 0;
};
__device__ inline int wq_take2(const unsigned long long *plain, int *plain_head, const unsigned long long *prioq, int *prio_tail,
                               int *prio_head, WqWorker &w, bool look, int max_naps, int *err, const int *end_word, int epoch,
                               bool &prio) {
  const long long t0 = wall_clock64();
  prio               = false;
  for (int naps = 0;;) {
    if (look) {
      const int pt = __builtin_amdgcn_readfirstlane(__hip_atomic_load(prio_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (pt - w.seen_ph > 0) {
        const int h = __builtin_amdgcn_readfirstlane(__hip_atomic_load(prio_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        w.seen_ph   = h;
        if (pt - h > 0) {
          int got = 0;
          if ((threadIdx.x & 63) == 0) {
            int e = h;
            got   = __hip_atomic_compare_exchange_strong(prio_head, &e, h + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
          }
          if (__builtin_amdgcn_readfirstlane(got)) {  // position h is reserved by its producer: the descriptor is there or about to be
            prio = true;
            return wq_take_end(prioq, (unsigned)h, err, end_word, epoch, false);
          }
        }
      }
    }
    if (!w.have_plain) {
      w.plain_t    = (unsigned)flow_ticket(plain_head);
      w.have_plain = true;
    }
    {
      const unsigned long long v  = __hip_atomic_load(plain + (w.plain_t % FL_WQ_SLOTS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned           hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      if (hi == w.plain_t / FL_WQ_SLOTS + 1u) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        w.have_plain = false;
        return (int)__builtin_amdgcn_readfirstlane((unsigned)v);
      }
    }
    for (int i = 0; i <= (naps < max_naps ? naps : max_naps); ++i) flow_pause();  // 14 us ... an idle wave polls less and less
    ++naps;
    if ((naps & 1) == 0 &&
        __builtin_amdgcn_readfirstlane(__hip_atomic_load(end_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == epoch)
      return -2;
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) return -1;
    if (wall_clock64() - t0 > FLOW_TIMEOUT_TICKS) {
      if ((threadIdx.x & 63) == 0) atomicExch(err, 15);
      return -1;
    }
  }
}
#endif
// the map role's arguments (csrc/sogm_map.hip, k_flight_map)
struct FlightMapDev {
  void                 *grid;          // the context's current grid (all agents)
  unsigned             *bits;          // [A][words]
  int                   words;
  const FlightWorld    *worlds;        // dev [n_ticks]
  CloudBlocks           cb;            // crop lists (bounds / counts per frame come from worlds[k])
  void                 *cand;          // [A][SOGM_MAX_CYL_LDS] CylCand
  int                  *n_cand;
  MarkLog               lg;
  const SogmTrajRecord *own;           // [A] executed records (latest wins)
  const SogmTrajRecord *tables;        // [4][n_total] ring of swarm tables
  int                   n_total;
  const int32_t        *ego_ids;
  const double         *body;
  int                   n_body;
  double                t0, period, start_offset;
  double               *hover, *now, *t_start, *pva;
  float                *poses;         // the context's map centres / stamps (queries read them)
  double               *stamps;
  int                   n_reset, n_bits, n_marks, n_splat;  // one-wave tickets per agent and tick
  int                   un_reset, un_bits, un_marks, un_splat;  // ... of a map in the urgent lane
  int                   n_head_wgs;    // workgroups 0 .. n_head_wgs - 1 of the launch admit agents (heads), the rest work off the queue
  int                   n_uhead_wgs;   // the first n_uhead_wgs of the heads serve the urgent ring
  int                   n_uwork_wgs;   // the first n_uwork_wgs of the workers look at the urgent queue first
  int                   n_admit;       // agents whose map may be under construction at once
  int                   pace_ticks;    // 100 MHz ticks between two admissions (0 = as fast as the heads run)
  size_t                agent_bytes;
  unsigned long long   *reset_stat;    // the context's reset statistics (sogm_sparse_reset_state / sogm_map_traffic), or null
};
int launch_flight_map(const GridGeom &g, const FlightCtl &fl, const FlightMapDev &d, int n_workgroups, hipStream_t st);
// what finish_agent (csrc/sogm_corridor.hip) reads and writes for one agent
struct FinishArgs {
  double                corridor_tau;
  const int32_t        *ret, *npoly, *status;
  const double         *cpts;
  const SogmTrajRecord *swarm;
  int                   n_swarm;
  const int32_t        *swarm_ego;
  const double         *swarm_now, *t_start;
  const int32_t        *drone_ids;
  SogmTrajRecord       *out;
  int32_t              *out_ok, *out_safe;
  unsigned long long   *counters;
  SogmTrajRecord       *pub_own, *pub_table;
};
// the light roles' arguments (k_flight_light): corridor stage buffers + the finishing role's
struct FlightLightDev {
  const double   *start_pva, *t_start, *route;
  const int32_t  *route_len;
  int             route_cap;
  double         *out_polys;
  int32_t        *out_nfaces, *out_npoly;
  double         *out_goal;
  FinishArgs      fin;          // swarm / pub_table / out / out_ok are set per tick from the fields below
  SogmTrajRecord *tables;       // [4][n_total] ring of swarm tables (null: no deconfliction, no table)
  int             n_total, agent0;
  SogmTrajRecord *log_records;  // [n_ticks][A]
  int32_t        *log_ok;       // [n_ticks][A]
};
struct CorridorWorkspace;
int launch_flight_light(const MapView &m, const SogmPlannerParams &pp, const CorridorWorkspace &ws, const FlightCtl &fl,
                        const FlightLightDev &d, int n_workgroups, hipStream_t st);
struct AstarWorkspace;
int launch_flight_search(const MapView &m, const SogmAstarParams &ap, double corridor_tau, const AstarWorkspace &wsp,
                         const FlightCtl &fl, int n_workgroups, const double *start_pva, const double *goal,
                         const double *t_start, int32_t *out_ret, double *out_route, int32_t *out_route_len, int route_cap,
                         int32_t *out_stats, int spec, hipStream_t st);
struct QpWorkspace;
struct QpConst;
int launch_flight_qp(const SogmPlannerParams &pp, const SogmQpSettings &qs, const QpWorkspace &ws, const QpConst &qc,
                     const FlightCtl &fl, int n_workgroups, const double *start_pva, const double *goal_pv,
                     const double *polys, const int32_t *nfaces, const int32_t *npoly, double *out_cpts,
                     int32_t *out_status, int32_t *out_iters, hipStream_t st);

// Arguments of the pre-stamp kernel (csrc/sogm_map.hip, k_prestamp_flow): the next tick's update inputs, the grid and
// mark log it builds into, and where the next tick's start states go.
struct PrestampDev {
  void                 *grid;         // the pool's next grid (all agents)
  unsigned             *bits;         // [A][words] occupancy bits of slice 0
  int                   words;
  const float          *cloud;
  const int32_t        *cloud_range;  // per-agent {begin, end} (null with cb.bounds set)
  CloudBlocks           cb;           // SogmWorld frame: blocks + the context's crop lists (bounds null = ranges above)
  const SogmCylinder   *cyl;
  int                   n_cyl;
  void                 *cand;         // [A][1024] CylCand
  int                  *n_cand;       // [A]
  MarkLog               lg;
  const SogmTrajRecord *own;          // the records the replan publishes into
  double                stamp, start_offset;
  double               *hover, *now, *t_start, *pva;  // sogm_tick_inputs' outputs for the next tick
  float                *poses;        // the context's NEXT poses / stamps (swapped in by sogm_update_prestamped)
  float                *poses_host;   // the caller's copy of the next map centres (optional)
  double               *stamps;
  int                   n_agents;
  int                   n_bits, n_marks;  // one-wave tickets per agent for the two passes of the stamp
  int                   n_late, n_bits_late, n_marks_late;  // ... and for the last n_late agents to be published
  int                   gate_agents;  // agents whose corridors must be final before the pre-stamp starts (tuning key prestamp_gate_frac)
};
// n_qp / n_finish: the QP workgroups and finishing waves of this replan — the pre-stamp's waves are not dispatched before
// all of them are resident (they wait for what those produce, and a QP workgroup needs a whole CU)
int launch_prestamp_flow(const GridGeom &g, const FlowCtl &fc, const PrestampDev &ps, int n_workgroups, int n_qp,
                         int n_finish, hipStream_t st);

// ParticleATC::isSafeAfterOpt for agents [agent0, agent0 + n_agents): out_safe[a] = 1 / 0
int launch_deconflict(int n_agents, const double *cpts, const int32_t *npoly, const SogmTrajRecord *rec,
                      int n_rec, const int32_t *ego_ids, const double *t_now, int32_t *out_safe,
                      hipStream_t st, int agent0, unsigned long long *counters = nullptr);
int launch_corridor(const MapView &m, const SogmPlannerParams &pp, const CorridorWorkspace &ws,
                    int n_agents, const double *start_pva, const double *t_start,
                    const double *route, const int32_t *route_len, int route_cap,
                    double *out_polys, int32_t *out_nfaces, int32_t *out_npoly, double *out_goal,
                    hipStream_t st, int agent0 = 0, hipEvent_t ev_map_read = nullptr);

// dataflow replan launchers (persistent kernels; see k_corridor_flow / k_qp_flow / k_finish_flow)
int launch_flow_gate(const FlowCtl &fc, int expected, hipStream_t st);
int launch_corridor_flow(const MapView &m, const SogmPlannerParams &pp, const CorridorWorkspace &ws,
                         const FlowCtl &fc, int n_agents, int n_workgroups, const double *start_pva,
                         const double *t_start, const double *route, const int32_t *route_len, int route_cap,
                         double *out_polys, int32_t *out_nfaces, int32_t *out_npoly, double *out_goal,
                         hipStream_t st);
int launch_finish_flow(const FlowCtl &fc, int n_agents, int n_workgroups, double corridor_tau, const int32_t *ret,
                       const int32_t *npoly, const int32_t *status, const double *cpts, const SogmTrajRecord *swarm,
                       int n_swarm, const int32_t *swarm_ego, const double *swarm_now, const double *t_start,
                       const int32_t *drone_ids, SogmTrajRecord *out, int32_t *out_ok, int32_t *out_safe,
                       unsigned long long *counters, hipStream_t st, SogmTrajRecord *pub_own = nullptr,
                       SogmTrajRecord *pub_table = nullptr);

// Per-agent QP row storage in HBM, used only when a problem's rows do not fit in LDS.
struct QpWorkspace {
  char  *scratch;         // [A][scratch_stride]
  size_t scratch_stride;  // bytes per agent (rows at M = SOGM_MAX_PIECES, max_faces faces)
  int    dyn_lds_bytes;   // dynamic LDS per workgroup of k_qp
  double *k1_scratch;     // [A][120 * 18] band of A^T diag(rho / rho_cur) A when it does not fit in LDS beside the rows
  // BezierOpt::setup in full (sogm_bezier_qp_solve_timed): per-piece time allocation [A][SOGM_MAX_PIECES] and an
  // end state with acceleration (goal rows of 9 doubles instead of 6).  nullptr / 6: every piece = corridor_tau,
  // final acceleration 0 — what replan() asks for (baseline.cpp:411,423).
  const double *t_alloc;
  int           goal_stride;
  // diagnostics (tools/): [A][16] wall_clock64 ticks (100 MHz) of the agent's last solve — 0 total, 1 set-up (assembly,
  // scaling, first factorisation), 2 later refactorisations, 3 their count, 4 checks (residual passes + certificates),
  // 5 their count, 6 iterations run, 8-10 the three phases of factor() summed over all factorisations; always written (a handful of clock reads per 25 iterations)
  long long    *dbg;
  int           ablate;  // phase ablation mask of the staged k_qp launch (profiling builds only; tuning key qp_ablate)
};
size_t qp_scratch_bytes_per_agent(int max_faces);
size_t qp_k1_scratch_bytes_per_agent();
int    qp_dynamic_lds_bytes();
struct QpConst {
  double QM[225];  // per-piece min-jerk cost block (bezier_optimizer.cpp:96-111)
};
int astar_resident_workgroups(int device);
int launch_qp(const SogmPlannerParams &pp, const SogmQpSettings &qs, const QpWorkspace &ws,
              const QpConst &qc, int n_agents, const double *start_pva, const double *goal_pv,
              const double *polys, const int32_t *nfaces, const int32_t *npoly, double *out_cpts,
              int32_t *out_status, int32_t *out_iters, hipStream_t st, int agent0 = 0);

int launch_qp_flow(const SogmPlannerParams &pp, const SogmQpSettings &qs, const QpWorkspace &ws,
                   const QpConst &qc, const FlowCtl &fc, int n_agents, int n_workgroups, const double *start_pva,
                   const double *goal_pv, const double *polys, const int32_t *nfaces, const int32_t *npoly,
                   double *out_cpts, int32_t *out_status, int32_t *out_iters, hipStream_t st);

size_t astar_node_bytes();
int    launch_astar(const MapView &m, const SogmAstarParams &ap, double corridor_tau,
                    const AstarWorkspace &wsp, int n_agents, const double *start_pva,
                    const double *goal, const double *t_start, int32_t *out_ret,
                    double *out_route, int32_t *out_route_len, int route_cap, int32_t *out_stats,
                    int32_t *out_trace, int trace_cap, hipStream_t st, int agent0 = 0, const FlowCtl *fc = nullptr,
                 int search_mode = 0);

}  // namespace sogm

#define SOGM_MAX_GROUPS 64

struct sogm_planner {
  sogm_ctx            *map;
  SogmAstarParams      ap;
  SogmPlannerParams    pp;
  SogmQpSettings       qs;
  sogm::AstarWorkspace aw;
  sogm::CorridorWorkspace cw;
  sogm::QpWorkspace qw;
  sogm::QpConst qc;
  // internal buffers used by sogm_replan (device)
  int32_t *d_ret, *d_route_len, *d_stats;
  double  *d_route;
  int      route_cap;
  double  *d_polys, *d_goal, *d_cpts;
  int32_t *d_nfaces, *d_npoly, *d_status, *d_iters;
  // post-optimisation deconfliction (ParticleATC::isSafeAfterOpt); off while swarm == nullptr
  int32_t              *d_safe;
  const SogmTrajRecord *swarm;
  int                   n_swarm;
  const int32_t        *swarm_ego;
  const double         *swarm_now;
  // publication inside the replan (sogm_planner_set_publish): the host's own-record table and the next swarm table
  SogmTrajRecord       *pub_own, *pub_table;
  // agent groups: sogm_replan runs each group's search -> corridors -> QP chain on its own stream,
  // so one slow agent (a long A* search, an infeasible QP) only delays its own group
  int         n_groups;
  hipStream_t gstream[SOGM_MAX_GROUPS];
  hipEvent_t  ev_in, ev_corr[SOGM_MAX_GROUPS], ev_done[SOGM_MAX_GROUPS];
  hipEvent_t  ev_pts[SOGM_MAX_GROUPS];  // after a group's obstacle-point kernel: its last read of the SOGM
  // dataflow replan (persistent kernels chained per agent through device-side ready lists)
  int            flow;        // 1 = use it (pipelining modes other than the in-place pre-clear)
  int           *d_flow;      // FLOW_HDR + 4 A ints: header, seg_done, a_ready, q_ready, f_ready
  long long     *d_flow_ts;   // [A][8]
  sogm::FlowCtl  fc;
  hipStream_t    fstream[4];  // A*, corridors, QP, finish
  hipEvent_t     ev_pdone;
  sogm::PrestampDev ps;       // the host's part of the pre-stamp arguments
  int            ps_on;
  int            ps_world_on;  // the pre-stamp's inputs are a SogmWorld frame (copied at sogm_planner_set_prestamp)
  SogmWorld      ps_world;
  hipEvent_t     ev_gate, ev_fdone[4];
  int           *d_epoch;      // device word: the clear epoch of the replan in flight (sogm_ctx::clear_epoch_word)
  int            reset_epoch;  // generation of the control block's reset (k_flow_reset writes it into d_flow's last word)
  int           *h_flow_fail;  // pinned, device-visible: {last FLOW_ERR code, ticks that failed} (k_flow_report)
  // per-object use of the per-stage entries (sogm_planner_select_agents / _set_search_mode)
  int sel_first, sel_count;  // agents the per-stage entries process; (0, A) by default
  int search_mode;           // 0 the replan's two-call pattern, 1 / 2 one search with init_search true / false
  int spec_astar;            // dataflow replan: run the second search attempt speculatively beside the first
  hipStream_t peek;          // sogm_debug_flow_peek's private stream (created on first use)
  // flight (sogm_flight_run): control block, per-agent tick inputs, frames, masked streams (all created on first use)
  sogm::FlightCtl     fl;
  int                *d_fl;          // header + rings + tick_done + tick_of + seg_done + stage
  sogm::FlightWorld  *d_fl_worlds, *h_fl_worlds;  // [FLIGHT_MAX_TICKS] device / pinned staging
  double             *d_fl_pva, *d_fl_tstart, *d_fl_now;
  hipStream_t         fl_stream[4];  // QP, search, corridor + finish, map
  hipEvent_t          fl_ev_in, fl_ev_done[4];
  int                 fl_cus[4];     // compute units of each stream's mask
  int                 fl_wgs[4];     // workgroups of each kernel
  int                 fl_epoch = 0;  // number of the last sogm_flight_run call (FlightCtl::epoch)
  int                *fl_xready = nullptr;  // [FLIGHT_MAX_TICKS] FlightCtl::xready of a multi-rank flight
};
