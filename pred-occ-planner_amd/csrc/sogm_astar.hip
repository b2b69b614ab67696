// sogm_astar.hip — batched 4-D (x,y,z,t) kinodynamic hybrid A* on the SOGM, one workgroup per agent.
//
// Reference: FakeRiskHybridAstar (path_searching/src/fake_risk_hybrid_a_star.cpp:84-426,428-587,
// 663-694,795-836; node/hash/heap types path_node.h:37-97, grid_node.h:10-51).
//
// Mapping to CDNA4: one workgroup of two waves (128 lanes) per agent.  The search is a serial program
// (best-first pop, ordered merge of the children into hash table / heap), but almost all of its work
// is not: lane i evaluates motion primitive i of the popped node (75 of the 128 lanes) — state
// transition, voxel/time index, velocity gate, the K-cell SOGM collision gather, the quartic-root
// heuristic AND the hash probe of the child's key.  The reference's ordered "first child of a voxel
// wins, later ones only compare costs" merge is resolved in parallel as well (leader / duplicate
// detection with a prefix-min over primitives), producing a compact event list (new node / cheaper
// duplicate / re-open / error) with ballot ranks; one lane replays only those few events against the
// binary heap (f-scores mirrored in LDS, 16-bit heap slots), all lanes write the new nodes and CAS
// their keys (x,y,z 13 bit, t 9 bit, id 14 bit packed in 64 bits) into the open-addressing table.
// The results are pure functions of (node, primitive), so evaluating them eagerly for children the
// reference would have skipped changes nothing.  Everything is gather / latency bound against an
// L2-resident working window of the SOGM; there is no dense contraction (no MFMA).
//
// Bit-exactness contract (north_star): expansions are bit-identical to the CPU oracle.  fp64
// arithmetic is written in the same operation order, compiled with -ffp-contract=off;
// cbrt/acos/cos come from include/sogm_detmath.h on both sides; the open list reproduces
// libstdc++'s push_heap/pop_heap (std::priority_queue) including its behaviour when f-scores are
// edited in place; the closed/open hash keeps the reference's insert-does-not-overwrite semantics
// and its (int)time vs time_idx key mismatch (:387 vs :271).
#include <hip/hip_runtime.h>

#include "../../include/sogm_detmath.h"
#include "sogm_planner.hpp"

namespace sogm {

namespace {

enum { IN_CLOSE_SET = 1, IN_OPEN_SET = 2, NOT_EXPAND = 3 };
enum { NO_PATH = 0, INIT_ERR, SEARCH_ERR, REACH_HORIZON, REACH_END, NEAR_END };

__device__ inline double dot3(const double *a, const double *b) {
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

// stateTransit (:812-824): phi * x0 + [0.5 tau^2 u ; tau u]
__device__ inline void state_transit(const double *s0, double *s1, const double *um, double tau) {
  const double h = 0.5 * (tau * tau);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    s1[i]     = (s0[i] + tau * s0[i + 3]) + h * um[i];
    s1[i + 3] = s0[i + 3] + tau * um[i];
  }
}

// cubic / quartic (:525-587)
__device__ inline int cubic(double a, double b, double c, double d, double *out) {
  const double a2 = b / a, a1 = c / a, a0 = d / a;
  const double Q = (3 * a1 - a2 * a2) / 9;
  const double R = (9 * a1 * a2 - 27 * a0 - 2 * a2 * a2 * a2) / 54;
  const double D = Q * Q * Q + R * R;
  if (D > 0) {
    const double S = sogm_det::cbrt(R + sogm_det::sqrt_rn(D));
    const double T = sogm_det::cbrt(R - sogm_det::sqrt_rn(D));
    out[0]         = -a2 / 3 + (S + T);
    return 1;
  } else if (D == 0) {
    const double S = sogm_det::cbrt(R);
    out[0]         = -a2 / 3 + S + S;
    out[1]         = -a2 / 3 - S;
    return 2;
  } else {
    const double PI    = 3.14159265358979323846;
    const double theta = sogm_det::acos(R / sogm_det::sqrt_rn(-Q * Q * Q));
    out[0]             = 2 * sogm_det::sqrt_rn(-Q) * sogm_det::cos(theta / 3) - a2 / 3;
    out[1]             = 2 * sogm_det::sqrt_rn(-Q) * sogm_det::cos((theta + 2 * PI) / 3) - a2 / 3;
    out[2]             = 2 * sogm_det::sqrt_rn(-Q) * sogm_det::cos((theta + 4 * PI) / 3) - a2 / 3;
    return 3;
  }
}

// Roots land in FIXED slots (out[0..1] the D pair, out[2..3] the E pair) with a validity mask instead of a
// running count: a dynamically indexed local array would live in scratch memory.  Reading the valid slots in
// slot order reproduces the reference's root order.
__device__ inline unsigned quartic(double a, double b, double c, double d, double e, double out[4]) {
  const double a3 = b / a, a2 = c / a, a1 = d / a, a0 = e / a;
  double       ys[3];
  cubic(1, -a2, a1 * a3 - 4 * a0, 4 * a2 * a0 - a1 * a1 - a3 * a3 * a0, ys);
  const double y1 = ys[0];
  const double r  = a3 * a3 / 4 - a2 + y1;
  out[0] = out[1] = out[2] = out[3] = 0.0;
  if (r < 0) return 0u;
  const double R = sogm_det::sqrt_rn(r);
  double       D, E;
  if (R != 0) {
    D = sogm_det::sqrt_rn(0.75 * a3 * a3 - R * R - 2 * a2 +
                          0.25 * (4 * a3 * a2 - 8 * a1 - a3 * a3 * a3) / R);
    E = sogm_det::sqrt_rn(0.75 * a3 * a3 - R * R - 2 * a2 -
                          0.25 * (4 * a3 * a2 - 8 * a1 - a3 * a3 * a3) / R);
  } else {
    D = sogm_det::sqrt_rn(0.75 * a3 * a3 - 2 * a2 + 2 * sogm_det::sqrt_rn(y1 * y1 - 4 * a0));
    E = sogm_det::sqrt_rn(0.75 * a3 * a3 - 2 * a2 - 2 * sogm_det::sqrt_rn(y1 * y1 - 4 * a0));
  }
  unsigned mask = 0;
  if (!(D != D)) {
    out[0] = -a3 / 4 + R / 2 + D / 2;
    out[1] = -a3 / 4 + R / 2 - D / 2;
    mask |= 3u;
  }
  if (!(E != E)) {
    out[2] = -a3 / 4 - R / 2 + E / 2;
    out[3] = -a3 / 4 - R / 2 - E / 2;
    mask |= 12u;
  }
  return mask;
}

// estimateHeuristic (:428-468)
__device__ inline double estimate_heuristic(const SogmAstarParams &ap, const double *x1,
                                            const double *x2, double &optimal_time) {
  double dp[3], v0[3], v1[3], vs[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    dp[i] = x2[i] - x1[i];
    v0[i] = x1[i + 3];
    v1[i] = x2[i + 3];
    vs[i] = v0[i] + v1[i];
  }
  const double c1 = -36 * dot3(dp, dp);
  const double c2 = 24 * dot3(vs, dp);
  const double c3 = -4 * (dot3(v0, v0) + dot3(v0, v1) + dot3(v1, v1));
  const double c4 = 0;
  const double c5 = ap.w_time;
  double       ts[5];
  unsigned     mask  = quartic(c5, c4, c3, c2, c1, ts);
  const double v_max = ap.max_vel * 0.5;
  double       linf  = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double d = sogm_det::fabs_(x1[i] - x2[i]);
    linf           = d > linf ? d : linf;
  }
  const double t_bar = linf / v_max;
  ts[4]              = t_bar;  // the reference appends t_bar after the roots
  mask |= 16u;
  double cost = 100000000, t_d = t_bar;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const double t = ts[i];
    if (!((mask >> i) & 1u) || t < t_bar) continue;
    const double c = -c1 / (3 * t * t * t) - c2 / (2 * t * t) - c3 / t + ap.w_time * t;
    if (c < cost) {
      cost = c;
      t_d  = t;
    }
  }
  optimal_time             = t_d;
  const double tie_breaker = 1.0 + 1.0 / 10000;
  return 1.0 * (1 + tie_breaker) * cost;
}

// ---- per-agent search state ---------------------------------------------------------------------
// Node records live in HBM (128 B each, written once per new node, re-read only on the final path).
// What the serial master loop touches on its critical path is on-chip:
//   * the open list (binary heap of node ids) and an f-score mirror are in LDS;
//   * the closed/open hash is probed by all lanes in parallel BEFORE the ordered merge (an entry
//     inserted during an expansion can never change that expansion's outcome: a later child that
//     would find it is in the same (voxel, time index) as the inserting child and is therefore
//     pruned by the same-parent rule first, :345-362) and filled by all lanes in parallel AFTER it;
//   * each hash slot is one 64-bit word {13-bit x, y, z, 9-bit t, 14-bit node id} claimed by CAS,
//     so a probe is a single load and concurrent inserts need no lock.
struct __attribute__((aligned(16))) Node {
  double state[6];
  double input[3];
  double duration;
  double time;
  double g, f;
  int    index[3];
  int    time_idx;
  int    parent;
  int    node_state;
};  // 128 B

#define ASTAR_POOL_MAX 10240  // LDS f mirror: 80 KB
#define HASH_EMPTY 0xFFFFFFFFFFFFFFFFull

__device__ inline bool pack_ok(int a, int b, int c, int t) {
  return a >= -4096 && a < 4096 && b >= -4096 && b < 4096 && c >= -4096 && c < 4096 && t >= -256 && t < 256;
}
__device__ inline unsigned long long pack_key(int a, int b, int c, int t) {
  return ((unsigned long long)(a + 4096) << 49) | ((unsigned long long)(b + 4096) << 36) |
         ((unsigned long long)(c + 4096) << 23) | ((unsigned long long)(t + 256) << 14);
}
__device__ inline unsigned hash_of(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  return (unsigned)k;
}
// NodeHashTable::find (path_node.h:86-89)
__device__ inline int hash_find(const unsigned long long *tab, int cap, unsigned long long key) {
  unsigned s = hash_of(key) & (cap - 1);
  while (true) {
    const unsigned long long v = __hip_atomic_load(tab + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v == HASH_EMPTY) return -1;
    if ((v & ~0x3FFFull) == key) return (int)(v & 0x3FFF);
    s = (s + 1) & (cap - 1);
  }
}
// NodeHashTable::insert (path_node.h:79-82): unordered_map::insert keeps an existing entry
__device__ inline void hash_insert(unsigned long long *tab, int cap, unsigned long long key, int val) {
  unsigned s = hash_of(key) & (cap - 1);
  while (true) {
    unsigned long long expect = HASH_EMPTY;
    if (__hip_atomic_compare_exchange_strong(tab + s, &expect, key | (unsigned long long)val,
                                             __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT))
      return;
    if ((expect & ~0x3FFFull) == key) return;
    s = (s + 1) & (cap - 1);
  }
}

// libstdc++ std::push_heap / std::pop_heap with NodeComparator (f(a) > f(b)), restated on an LDS
// heap of node ids with the f-scores mirrored in LDS.
// (std::__push_heap.  The ids of up to three ancestors and then their f are read as two batches of independent LDS
//  loads — a dependent LDS read costs ~64 cycles — before the level-by-level comparisons; the positions read ahead are
//  only written once the hole has moved past them, so the values are the ones the plain loop would see.)
__device__ inline void heap_push_up(const double *f, unsigned short *h, int hole, int top, int value, double fv) {
  while (hole > top) {
    const int p1 = (hole - 1) / 2;
    const int p2 = p1 > top ? (p1 - 1) / 2 : p1;
    const int p3 = p2 > top ? (p2 - 1) / 2 : p2;
    const int i1 = h[p1], i2 = h[p2], i3 = h[p3];
    const double f1 = f[i1], f2 = f[i2], f3 = f[i3];
    if (!(f1 > fv)) break;
    h[hole] = (unsigned short)i1;
    hole    = p1;
    if (!(hole > top) || !(f2 > fv)) break;
    h[hole] = (unsigned short)i2;
    hole    = p2;
    if (!(hole > top) || !(f3 > fv)) break;
    h[hole] = (unsigned short)i3;
    hole    = p3;
  }
  h[hole] = (unsigned short)value;
}
__device__ inline void heap_push(const double *f, unsigned short *h, int &n, int value, double fv) {  // fv = f[value]
  h[n] = (unsigned short)value;
  ++n;
  heap_push_up(f, h, n - 1, 0, value, fv);  // new nodes rarely beat their parent: the loop exits after a level or two
}
// std::push_heap by a whole wave (all 64 lanes call it with the same arguments; n, value, fv uniform): the path
// from the new leaf to the root is known in advance, so lane j-1 fetches the level-j ancestor's id and f (two LDS
// latencies for the whole path instead of two per level — the children of an expansion have f close to the open
// list's minimum and climb most of its ~12 levels), the hole climbs while "ancestor's f > fv" holds (the first lane
// where it does not ends it: the loop of std::__push_heap, also where in-place f updates have bent the heap
// property), and the moved ancestors are written one level down by their lanes.
__device__ inline void wave_heap_push(const double *f, unsigned short *h, int &n, int value, double fv, int lane) {
  const int q0 = n + 1;  // 1-based position of the new leaf
  ++n;
  const int                j     = lane + 1;
  const int                q     = j < 31 ? q0 >> j : 0;  // 1-based position of the level-j ancestor (0: above the root)
  const bool               valid = q >= 1;
  const int                id    = valid ? h[q - 1] : 0;
  const double             fj    = f[id];
  const unsigned long long mv    = __ballot(valid && fj > fv);
  const int                k     = __builtin_ctzll(~mv);  // levels the hole climbs (lanes >= 14 are never valid)
  if (j <= k) h[(q0 >> (j - 1)) - 1] = (unsigned short)id;
  if (lane == 0) h[(q0 >> k) - 1] = (unsigned short)value;
}
__device__ inline void heap_pop(const double *f, unsigned short *h, int &n) {
  if (n > 1) {
    // __pop_heap(first, last-1, last-1): value = *(last-1); *(last-1) = *first; adjust(first,0,len-1)
    const int value = h[n - 1];
    h[n - 1]        = h[0];
    const int len   = n - 1;
    int       hole  = 0;
    int       child = 0;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (f[h[child]] > f[h[child - 1]]) child--;
      h[hole] = h[child];
      hole    = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
      child   = 2 * (child + 1);
      h[hole] = h[child - 1];
      hole    = child - 1;
    }
    heap_push_up(f, h, hole, 0, value, f[value]);
  }
  --n;
}

#define ASTAR_MAX_INPUTS 128
// Waves 0-1 evaluate one motion primitive per lane; wave 2's lane 0 is the MASTER that owns the open list (pop,
// ordered replay of the children's events, termination): being on a wave of its own it can restore the heap after a
// pop (a full-depth sift-down of dependent LDS reads, ~2 us) WHILE the other two waves evaluate the children.
#define ASTAR_THREADS 192
#define ASTAR_MASTER 128

// per-child action decided in parallel, replayed in child order by the master
enum { EV_NONE = 0, EV_NEW = 1, EV_DUP = 2, EV_OPEN = 3, EV_ERR = 4 };

__device__ inline void pos_to_index(const double *p, const double *center, double inv_res,
                                    int *out) {
#pragma unroll
  for (int i = 0; i < 3; ++i) out[i] = (int)floor((p[i] - center[i]) * inv_res);
}

}  // namespace

// One search by one workgroup (three 64-lane waves): waves 0-1 evaluate one motion primitive per lane, wave 2 owns the open list.
// `second` / `spec`: this workgroup runs the agent's SECOND attempt speculatively beside the first (the two searches are
// independent; the second one's result only counts if the first returns NO_PATH, baseline_fake.cpp:284-291) — a NO_PATH
// pair, the longest search a tick can hold, then takes the time of one search instead of two.  The two workgroups meet
// through wsp.verdict[agent]: vbase + 1 = the first attempt found a path, vbase + 2 = it did not, anything else =
// pending (vbase: 0 for the per-tick launches, whose verdicts are zeroed per replan; 4 x (tick + 1) in a flight, where
// the word is never reset).  Returns true (uniformly) when THIS workgroup wrote the agent's search outputs — the caller
// publishes the agent — and false when it leaves them to the other attempt.
__device__ __forceinline__ bool astar_search_wg(
    const MapView &m, const SogmAstarParams &ap, double corridor_tau, const AstarWorkspace &wsp,
    const double *__restrict__ start_pva, const double *__restrict__ goal,
    const double *__restrict__ t_start, int32_t *__restrict__ out_ret,
    double *__restrict__ out_route, int32_t *__restrict__ out_route_len, int route_cap,
    int32_t *__restrict__ out_stats, int32_t *__restrict__ out_trace, int trace_cap, int agent, int second, bool spec,
    int vbase, int *flow_err, int search_mode) {
  const int  tid    = threadIdx.x;

  __shared__ double             s_f[ASTAR_POOL_MAX];     // f-score mirror of every allocated node
  __shared__ unsigned short     s_heap[ASTAR_POOL_MAX];  // open list
  __shared__ double             s_inputs[ASTAR_MAX_INPUTS][3];
  __shared__ double             s_cstate[ASTAR_MAX_INPUTS][6];  // child states
  __shared__ double             s_cf[ASTAR_MAX_INPUTS], s_cg[ASTAR_MAX_INPUTS];
  __shared__ unsigned long long s_key[ASTAR_MAX_INPUTS];  // packed (voxel, time index) of child i
  __shared__ int                s_found[ASTAR_MAX_INPUTS];
  __shared__ unsigned char      s_gate[ASTAR_MAX_INPUTS];  // 1 = passes every gate of :272-331
  __shared__ unsigned char      s_ev[ASTAR_MAX_INPUTS];    // EV_* of child i
  __shared__ short              s_leader[ASTAR_MAX_INPUTS];  // first gate-passing child with same key
  __shared__ short              s_rank[ASTAR_MAX_INPUTS];    // #EV_NEW before child i
  __shared__ int                s_src[ASTAR_MAX_INPUTS];     // leader -> child whose data its node takes
  __shared__ short              s_events[ASTAR_MAX_INPUTS];  // children with an event, in order
  __shared__ unsigned long long s_ckey[ASTAR_MAX_INPUTS];    // keys of the gate-passing children, compacted
  __shared__ short              s_cidx[ASTAR_MAX_INPUTS];    // their child indices
  __shared__ int2               s_erec[ASTAR_MAX_INPUTS];    // events in child order: {type, operand}
  __shared__ double             s_erec_f[ASTAR_MAX_INPUTS];  // ... and the child's f
  __shared__ int                s_upd_node[ASTAR_MAX_INPUTS];  // open nodes lowered by this expansion so far
  __shared__ double             s_upd_g[ASTAR_MAX_INPUTS];
  __shared__ double             s_og[ASTAR_MAX_INPUTS];      // g of the open node a child hit, before this expansion
  __shared__ int                s_n_events, s_n_new_w0, s_n_ev_w0;
  __shared__ int                s_n_inputs;
  __shared__ double             s_cur_state[6];
  __shared__ double             s_cur_time, s_cur_g;
  __shared__ int                s_cur_index[3], s_cur_tidx;
  __shared__ int                s_n_active;  // primitives of this expansion (0 = stop, < 0 = abort)
  __shared__ int                s_first;     // 1 = "init" expansion (single input = start acc)
  __shared__ int                s_was_first, s_cur_node, s_base_node, s_n_written;
  __shared__ int                s_ret;
  __shared__ unsigned           s_closed[ASTAR_POOL_MAX / 32];  // bit n: node n is in the closed set
  __shared__ short              s_new_leader[ASTAR_MAX_INPUTS];  // rank of a new node -> its leader child
  __shared__ int                s_stop;  // set by the replay: the search ends after this expansion's nodes are written

  // the second attempt's pool / hash table follow the first attempts' ([n_agents_total .. 2 n_agents_total))
  const size_t        slot = (size_t)agent + (second ? (size_t)m.n_agents : 0);
  Node               *pool = (Node *)(wsp.pool + slot * wsp.pool_stride);
  unsigned long long *htab = (unsigned long long *)wsp.hkeys + slot * wsp.hash_cap;
  const int           hcap = wsp.hash_cap;

  const double *pva = start_pva + agent * 9;
  double        start_pt[3] = {pva[0], pva[1], pva[2]};
  double        start_v[3]  = {pva[3], pva[4], pva[5]};
  double        start_a[3]  = {pva[6], pva[7], pva[8]};
  double        end_state[6] = {goal[agent * 3], goal[agent * 3 + 1], goal[agent * 3 + 2], 0, 0, 0};
  const float  *pose      = m.poses + agent * 3;
  const double  center[3] = {(double)pose[0], (double)pose[1], (double)pose[2]};
  const double  inv_res   = 1.0 / ap.resolution;
  const double  inv_tres  = 1.0 / ap.time_resolution;
  // baseline_fake.cpp:282: t_after_map = traj_start_time_ - map_->getMapTime()
  // search_mode bit 2: t_start already is RiskHybridAstar::search's time_start argument (seconds after the map stamp)
  // search_mode bit 4: RiskHybridAstar::search(..., dynamic = false, ...) — the reference's branch never writes the
  // nodes' time / time_idx and reads them all the same (risk_hybrid_a_star.cpp:153-158,177,271,324): defined here, as
  // in oracle/astar_oracle.cpp, with every node's time and time index ZERO, under which the 4-D table, the prune and
  // the same-voxel test coincide with the branch's 3-D ones and the SOGM is sampled over [0, tau]
  const bool   static_time = (search_mode & 16) != 0;
  const double time_start  = static_time ? 0.0 : (search_mode & 4) ? t_start[agent] : t_start[agent] - m.stamps[agent];
  const double time_origin = time_start;
  const double tau         = ap.time_resolution;

  // motion primitive table (:245-250), generated with the reference's accumulating fp64 loops
  if (tid == 0) {
    int          n  = 0;
    const double ma = ap.max_acc, res = 1 / 2.0;
    for (double ax = -ma; ax <= ma + 1e-3; ax += ma * res)
      for (double ay = -ma; ay <= ma + 1e-3; ay += ma * res)
        for (double az = -0.5 * ma; az <= 0.5 * ma + 1e-3; az += ma * res) {
          if (n < ASTAR_MAX_INPUTS) {
            s_inputs[n][0] = ax;
            s_inputs[n][1] = ay;
            s_inputs[n][2] = az;
          }
          ++n;
        }
    s_n_inputs = n < ASTAR_MAX_INPUTS ? n : ASTAR_MAX_INPUTS;
  }

  int end_index[3];
  pos_to_index(end_state, center, inv_res, end_index);

  // master-only state
  int  use_node_num = 0, iter_num = 0, heap_n = 0, n_trace = 0;
  int  ret = NO_PATH, searches = 0, terminal = -1;
  bool is_shot_succ = false, need_pop = false;
  // The nodes an expansion creates are written to HBM by the lanes WHILE the master already checks the next pop (no
  // barrier in between).  If that pop is one of them (ids >= prev_base), the master takes its record from the
  // children's LDS arrays, which stay as they are until the next evaluation.
  int    prev_base = 0x7fffffff, prev_cur = -1, prev_new_ti = 0;
  double prev_new_t = 0.0;
  long long tk[6] = {0, 0, 0, 0, 0, 0};  // wall_clock64 ticks (100 MHz): pop, eval, dup, merge, write, n_exp
  long long tmark = 0;
  const bool timed = wsp.dbg != nullptr;  // phase statistics (sogm_debug_astar_stats): s_memrealtime is not free

  // search_mode 0: the replan's call pattern (init_search = true, then false if NO_PATH, baseline_fake.cpp:284-291);
  // 1 / 2: exactly one search(…, init = true / false, …) for the per-object shim
  const int attempt_lo = spec ? second : ((search_mode & 3) == 2 ? 1 : 0);
  const int attempt_hi = spec ? second + 1 : ((search_mode & 3) == 1 ? 1 : 2);
  bool      dropped    = false;  // second attempt: the first one found a path
  for (int attempt = attempt_lo; attempt < attempt_hi; ++attempt) {
    // reset(): clear the hash table (all lanes)
    for (int i = tid; i < hcap; i += ASTAR_THREADS) htab[i] = HASH_EMPTY;
    for (int i = tid; i < ASTAR_POOL_MAX / 32; i += ASTAR_THREADS) s_closed[i] = 0u;
    __syncthreads();
    bool done = false;
    prev_base = 0x7fffffff;
    if (tid == ASTAR_MASTER) {
      use_node_num = 0;
      iter_num     = 0;
      heap_n       = 0;
      is_shot_succ = false;
      terminal     = -1;
      ret          = NO_PATH;
      ++searches;
      Node &n0  = pool[0];
      n0.parent = -1;
      for (int i = 0; i < 3; ++i) {
        n0.state[i]     = start_pt[i];
        n0.state[i + 3] = start_v[i];
      }
      pos_to_index(start_pt, center, inv_res, n0.index);
      n0.g = 0.0;
      double ttg;
      n0.f          = ap.lambda_heu * estimate_heuristic(ap, n0.state, end_state, ttg);
      s_f[0]        = n0.f;
      n0.node_state = IN_OPEN_SET;
      heap_push(s_f, s_heap, heap_n, 0, n0.f);
      use_node_num += 1;
      n0.time     = time_start;
      n0.time_idx = (int)floor((time_start - time_origin) * inv_tres);
      if (pack_ok(n0.index[0], n0.index[1], n0.index[2], n0.time_idx))
        hash_insert(htab, hcap, pack_key(n0.index[0], n0.index[1], n0.index[2], n0.time_idx), 0);
      else
        ret = SEARCH_ERR;
      s_first = attempt == 0 ? 1 : 0;
      s_ret   = ret;
    }
    __syncthreads();
    if (s_ret == SEARCH_ERR) done = true;
    int cur = -1;
    while (!done) {
      // ---------------- master: pop / terminate ----------------
      if (timed) tmark = wall_clock64();
      if (tid == ASTAR_MASTER) {
        s_n_active = 0;
        // speculative second attempt: every 8th expansion, look whether the first attempt has found a path
        if (second && (iter_num & 7) == 0 &&
            __hip_atomic_load(&wsp.verdict[agent], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == vbase + 1) {
          s_ret = -1;  // dropped
        } else if (heap_n == 0) {
          ret = NO_PATH;  // open set empty (:419-422)
        } else {
          cur          = s_heap[0];
          Node cn;
          if (cur >= prev_base) {
            // created by the expansion that has just been replayed: its record is on its way to HBM; the same data
            // from the children's arrays (what the write phase stores)
            const int L = s_new_leader[cur - prev_base], i = s_src[L];
            for (int q = 0; q < 6; ++q) cn.state[q] = s_cstate[i][q];
            cn.g        = s_cg[i];
            cn.f        = s_cf[i];
            cn.time     = prev_new_t;
            cn.time_idx = prev_new_ti;
            pos_to_index(s_cstate[L], center, inv_res, cn.index);
            cn.parent = prev_cur;
          } else {
            cn = pool[cur];  // the whole record in one batch of loads
          }
          // (what the lanes need of it goes to LDS at once, whether or not the search stops here: every field is
          //  then fetched by the first batch instead of a second round trip after the termination tests)
          for (int i = 0; i < 6; ++i) s_cur_state[i] = cn.state[i];
          s_cur_time = cn.time;
          s_cur_g    = cn.g;
          for (int i = 0; i < 3; ++i) s_cur_index[i] = cn.index[i];
          s_cur_tidx = cn.time_idx;
          double d3[3] = {cn.state[0] - start_pt[0], cn.state[1] - start_pt[1],
                          cn.state[2] - start_pt[2]};
          const bool reach_horizon = sogm_det::sqrt_rn(dot3(d3, d3)) >= ap.horizon;
          const bool near_end      = abs(cn.index[0] - end_index[0]) <= ap.tolerance &&
                                abs(cn.index[1] - end_index[1]) <= ap.tolerance &&
                                abs(cn.index[2] - end_index[2]) <= ap.tolerance;
          const bool exceed_time = cn.time >= ap.max_tau;
          bool       stop        = false;
          if (reach_horizon || near_end || exceed_time) {
            terminal = cur;
            if (near_end) {
              // estimateHeuristic + computeShotTraj (:470-523); only feasibility is consumed
              double t_d;
              estimate_heuristic(ap, cn.state, end_state, t_d);
              double a[3], b[3], c[3], d[3];
              for (int i = 0; i < 3; ++i) {
                const double p0 = cn.state[i], dp = end_state[i] - p0, v0 = cn.state[i + 3],
                             v1 = end_state[i + 3], dv = v1 - v0;
                a[i] = 1.0 / 6.0 *
                       (-12.0 / (t_d * t_d * t_d) * (dp - v0 * t_d) + 6 / (t_d * t_d) * dv);
                b[i] = 0.5 * (6.0 / (t_d * t_d) * (dp - v0 * t_d) - 2 / t_d * dv);
                c[i] = v0;
                d[i] = p0;
              }
              const double t_delta = t_d / 10;
              bool         ok      = true;
              int          guard   = 0;  // t_d == 0 would spin forever in the reference
              for (double time = t_delta; time <= t_d && guard < 64; time += t_delta, ++guard) {
                const double t1 = time, t2 = time * time, t3 = (time * time) * time;
                double       co[3];
                for (int dim = 0; dim < 3; ++dim)
                  co[dim] = ((d[dim] * 1.0 + c[dim] * t1) + b[dim] * t2) + a[dim] * t3;
                // FakeRiskHybridAstar passes the shot-relative time (:521); RiskHybridAstar asks for slice 0
                // (risk_hybrid_a_star.cpp:514 -> risk_base.cpp:251-253)
                const int hit = ap.shot_ignores_time ? query_clear_idx(m, agent, co[0], co[1], co[2], 0)
                                                     : query_clear_time(m, agent, co[0], co[1], co[2], time);
                if (hit != 0) {
                  ok = false;
                  break;
                }
              }
              if (ok) is_shot_succ = true;
            }
          }
          if (reach_horizon) {
            ret  = is_shot_succ ? REACH_END : REACH_HORIZON;
            stop = true;
          } else if (near_end) {
            ret  = is_shot_succ ? REACH_END : (cn.parent >= 0 ? NEAR_END : NO_PATH);
            stop = true;
          } else if (exceed_time) {
            ret  = REACH_HORIZON;
            stop = true;
          }
          if (!stop) {
            need_pop = true;  // the heap is restored after the barrier, under the children's evaluation
            // closed: the lanes read this bit; the record's node_state is stored after the barrier (a global store
            // before it would be waited for)
            s_closed[cur >> 5] |= 1u << (cur & 31);
            iter_num += 1;
            s_n_active  = s_first ? 1 : s_n_inputs;
            s_was_first = s_first;
            s_first     = 0;  // init_search = false after the first expansion (:243)
            s_cur_node  = cur;
            s_base_node = use_node_num;
          }
        }
      }
      __syncthreads();
      const int n_act = s_n_active;
      if (n_act == 0) {
        if (second && s_ret == -1) dropped = true;  // uniform: s_ret is only set to -1 by the check above
        done = true;
        break;
      }
      if (tid == ASTAR_MASTER && need_pop) {  // std::pop_heap's sift-down, beside the evaluation below
        pool[cur].node_state = IN_CLOSE_SET;
        if (out_trace && n_trace < trace_cap) out_trace[(size_t)agent * trace_cap + n_trace] = cur;
        ++n_trace;
        heap_pop(s_f, s_heap, heap_n);
        need_pop = false;
      }
      if (timed) {
        const long long t2 = wall_clock64();
        tk[0] += t2 - tmark;
        tmark = t2;
      }
      // ---------------- all lanes: evaluate primitive `tid` + probe the hash ----------------
      const bool   first  = s_was_first != 0;
      const double new_t  = static_time ? 0.0 : s_cur_time + tau;
      const int    new_ti = static_time ? 0 : (int)floor((new_t - time_origin) * inv_tres);
      const int    my_base = s_base_node, my_cur = s_cur_node;  // (the master rewrites them while the lanes still write)
      prev_base   = my_base;
      prev_cur    = my_cur;
      prev_new_t  = new_t;
      prev_new_ti = new_ti;
      {
        const int i = tid;
        if (i < n_act) {
          double um[3];
          if (first) {
            um[0] = start_a[0];
            um[1] = start_a[1];
            um[2] = start_a[2];
          } else {
            um[0] = s_inputs[i][0];
            um[1] = s_inputs[i][1];
            um[2] = s_inputs[i][2];
          }
          double cs[6], ps[6];
          for (int q = 0; q < 6; ++q) cs[q] = s_cur_state[q];
          state_transit(cs, ps, um, tau);
          int id[3];
          pos_to_index(ps, center, inv_res, id);
          const int t_id = new_ti;  // timeToIndex(cur->time + tau), the same for every child
          bool      gate = !(fabs(ps[3]) > ap.max_vel || fabs(ps[4]) > ap.max_vel || fabs(ps[5]) > ap.max_vel);
          const bool same = id[0] == s_cur_index[0] && id[1] == s_cur_index[1] &&
                            id[2] == s_cur_index[2] && (t_id - s_cur_tidx) == 0;
          gate = gate && !same;
          // closed / open lookup (:271): find(pro_id, pro_t_id).  The first probe of the hash table is issued
          // before the collision gather and consumed after it, so the two global-memory latencies overlap (the
          // gates commute: a child is dropped if EITHER its node is closed or a sample collides).
          int                found = -1, found_st = NOT_EXPAND, ev = EV_NONE;
          unsigned long long key   = HASH_EMPTY, slot0 = HASH_EMPTY;
          const bool packable = pack_ok(id[0], id[1], id[2], t_id) && pack_ok(id[0], id[1], id[2], (int)new_t);
          if (packable) {
            key   = pack_key(id[0], id[1], id[2], t_id);
            slot0 = __hip_atomic_load(htab + (hash_of(key) & (hcap - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          bool collide = false;
          if (gate) {
            // collision gate (:296-331)
            for (int k = 1; k <= ap.check_num; ++k) {
              const double dt = tau * (double)k / (double)ap.check_num;
              double       xt[6];
              state_transit(cs, xt, um, dt);
              if (query_clear_time(m, agent, xt[0], xt[1], xt[2], s_cur_time + dt) != 0) {
                collide = true;
                break;
              }
            }
          }
          double found_g = 0.0;
          if (packable) {
            if (slot0 == HASH_EMPTY)
              found = -1;
            else if ((slot0 & ~0x3FFFull) == key)
              found = (int)(slot0 & 0x3FFF);
            else
              found = hash_find(htab, hcap, key);
            if (found >= 0) {
              // a node reached through the table is open or closed (LDS bit); its g matters for an open one only
              found_st = ((s_closed[found >> 5] >> (found & 31)) & 1u) ? IN_CLOSE_SET : IN_OPEN_SET;
              if (found_st == IN_OPEN_SET) found_g = pool[found].g;
            }
          }
          if (collide) gate = false;
          double cg = 0.0, cf = 0.0;
          if (gate) {
            double       ttg;
            const double usq = (um[0] * um[0] + um[1] * um[1]) + um[2] * um[2];
            cg               = (usq + ap.w_time) * tau + s_cur_g;
            cf = cg + ap.lambda_heu * estimate_heuristic(ap, ps, end_state, ttg);
          }
          if (found >= 0 && found_st == IN_CLOSE_SET) gate = false;
          if (gate) {
            if (!packable)
              ev = EV_ERR;  // index outside the packable range
            else if (found >= 0)
              ev = found_st == IN_OPEN_SET ? EV_OPEN : EV_ERR;
          }
          for (int q = 0; q < 6; ++q) s_cstate[i][q] = ps[q];
          s_cf[i]    = cf;
          s_cg[i]    = cg;
          s_key[i]   = key;
          s_found[i] = found;
          s_og[i]    = found_g;
          s_gate[i]  = gate ? 1 : 0;
          s_ev[i]    = (unsigned char)ev;
          s_src[i]   = i;
        }
      }
      __syncthreads();
      if (timed) {
        const long long t2 = wall_clock64();
        tk[1] += t2 - tmark;
        tmark = t2;
      }
      // ---------------- all lanes: same-parent duplicates (:345-362) and event list ----------------
      // leader(i) = first gate-passing child with the same key.  A child without a hash hit creates
      // a node if it is its own leader (EV_NEW); otherwise it is pruned against its leader's node
      // and replaces that node's data iff its f beats every earlier member of the group (EV_DUP).
      // The gate-passing children are first compacted (ascending child order): the entries before a child's own
      // slot are exactly the earlier members it has to look at, and their keys are read eight at a time.
      int my_ev = EV_NONE;
      {
        const int                lane = tid & 63, wave = tid >> 6;
        const unsigned long long lt   = lane ? (~0ull >> (64 - lane)) : 0ull;
        const bool               mg   = tid < n_act && s_gate[tid] != 0;
        const unsigned long long bg   = __ballot(mg);
        // (wave 1 counts wave 0's gate-passing children itself, from their flags: no barrier for a hand-over)
        const unsigned long long bg0  = __ballot(lane < n_act && s_gate[lane] != 0);
        const int crank = __popcll(bg & lt) + (wave == 1 ? __popcll(bg0) : 0);
        if (mg) {
          s_ckey[crank] = s_key[tid];
          s_cidx[crank] = (short)tid;
        }
        __syncthreads();
        if (tid < n_act) {
          const int i = tid;
          my_ev       = s_ev[i];
          if (mg && my_ev == EV_NONE) {
            const unsigned long long key = s_key[i];
            int                      L   = i;
            double                   mn  = 0.0;
            bool                     has = false;
            for (int c0 = 0; c0 < crank; c0 += 8) {
              unsigned long long kk[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) kk[u] = s_ckey[c0 + u < crank ? c0 + u : c0];
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                if (c0 + u < crank && kk[u] == key) {
                  const int    j  = s_cidx[c0 + u];
                  const double cj = s_cf[j];
                  if (!has) {
                    L   = j;
                    mn  = cj;
                    has = true;
                  } else if (cj < mn) {
                    mn = cj;
                  }
                }
              }
            }
            s_leader[i] = (short)L;
            if (!has) {
              my_ev = EV_NEW;
            } else if (s_cf[i] < mn) {
              my_ev = EV_DUP;
              atomicMax(&s_src[L], i);  // the last applied replacement wins (= first to reach the min)
            }
            s_ev[i] = (unsigned char)my_ev;
          }
        }
      }
      // ranks: number of EV_NEW / of events before child i (two waves)
      {
        const int                lane = tid & 63, wave = tid >> 6;
        const unsigned long long lt   = lane ? (~0ull >> (64 - lane)) : 0ull;
        const unsigned long long bn   = __ballot(my_ev == EV_NEW);
        const unsigned long long be   = __ballot(my_ev != EV_NONE);
        if (wave == 0 && lane == 0) {
          s_n_new_w0 = __popcll(bn);
          s_n_ev_w0  = __popcll(be);
        }
        __syncthreads();
        const int rnew = __popcll(bn & lt) + (wave == 1 ? s_n_new_w0 : 0);
        const int rev  = __popcll(be & lt) + (wave == 1 ? s_n_ev_w0 : 0);
        if (tid < n_act) {
          s_rank[tid] = (short)rnew;
          if (my_ev == EV_NEW) s_new_leader[rnew] = (short)tid;
          if (my_ev != EV_NONE) {
            s_events[rev] = (short)tid;
            // event record for the master's replay: {type, operand} + f.  operand = rank of the new node
            // (EV_NEW), leader child (EV_DUP), child (EV_OPEN / EV_ERR)
            s_erec[rev]   = make_int2(my_ev, my_ev == EV_NEW ? rnew : (my_ev == EV_DUP ? (int)s_leader[tid] : tid));
            s_erec_f[rev] = s_cf[tid];
          }
        }
        if (wave == 1 && lane == 0) s_n_events = s_n_ev_w0 + __popcll(be);
      }
      __syncthreads();
      if (timed) {
        const long long t2 = wall_clock64();
        tk[2] += t2 - tmark;
        tmark = t2;
      }
      // ---------------- master wave: replay the events in child order (:366-414) ----------------
      // Every lane of the master's wave runs the replay with the master's open-list state (the pushes use the
      // lanes, wave_heap_push); lane 0 alone writes what is not a push.
      if (tid >= ASTAR_MASTER) {
        const int lane = tid - ASTAR_MASTER;
        int       hn   = __builtin_amdgcn_readfirstlane(heap_n);
        int       unn  = __builtin_amdgcn_readfirstlane(use_node_num);
        int       rt   = __builtin_amdgcn_readfirstlane(ret);
        const int curb = __builtin_amdgcn_readfirstlane(cur);
        bool      dn   = false;
        const int n_ev = __builtin_amdgcn_readfirstlane(s_n_events);  // (uniform values in SGPRs: scalar branches)
        const int baseb = __builtin_amdgcn_readfirstlane(s_base_node);
        int       n_written = 0, n_upd = 0;
        int2      rec  = s_erec[0];
        double    rcf  = s_erec_f[0];
        for (int k = 0; k < n_ev && !dn; ++k) {
          const int    kn  = k + 1 < n_ev ? k + 1 : k;  // next record in flight while this one is replayed
          const int2   nrec = s_erec[kn];
          const double ncf  = s_erec_f[kn];
          const int    ev = __builtin_amdgcn_readfirstlane(rec.x), opd = __builtin_amdgcn_readfirstlane(rec.y);
          if (ev == EV_NEW) {
            const int node = baseb + opd;
            if (lane == 0) s_f[node] = rcf;
            wave_heap_push(s_f, s_heap, hn, node, rcf, lane);
            unn += 1;
            n_written = opd + 1;
            if (unn == ap.allocate_num) {  // "run out of memory" (:393-396)
              rt = NO_PATH;
              dn = true;
            }
          } else if (ev == EV_DUP) {
            if (lane == 0) s_f[baseb + s_rank[opd]] = rcf;
          } else if (ev == EV_OPEN) {
            if (lane == 0) {
              const int i     = opd;
              const int fnode = s_found[i];
              Node     &pn    = pool[fnode];
              // pn.g as this expansion's earlier children left it: the value fetched with the hash probe, or the g
              // of the last earlier child that lowered it (kept in a short list) — no global load on the serial path
              double g_now = s_og[i];
              int    slot  = -1;
              for (int u = 0; u < n_upd; ++u)
                if (s_upd_node[u] == fnode) {
                  g_now = s_upd_g[u];
                  slot  = u;
                }
              if (s_cg[i] < g_now) {
                for (int q = 0; q < 6; ++q) pn.state[q] = s_cstate[i][q];
                pn.f       = rcf;
                pn.g       = s_cg[i];
                s_f[fnode] = rcf;
                for (int q = 0; q < 3; ++q) pn.input[q] = first ? start_a[q] : s_inputs[i][q];
                pn.duration = tau;
                pn.parent   = curb;
                pn.time     = new_t;
                if (slot < 0) slot = n_upd++;
                s_upd_node[slot] = fnode;
                s_upd_g[slot]    = s_cg[i];
              }
            }
          } else {
            rt = SEARCH_ERR;
            dn = true;
          }
          rec = nrec;
          rcf = ncf;
        }
        // (the same values in every lane of the wave; the master lane's copies are the ones read later)
        heap_n       = hn;
        use_node_num = unn;
        ret          = rt;
        if (dn) done = true;
        if (lane == 0) {
          s_n_written = n_written;
          s_stop      = dn ? 1 : 0;
        }
      }
      __syncthreads();
      if (s_stop) done = true;  // (uniform; read here: the master is about to reuse the control words)
      if (timed) {
        const long long t2 = wall_clock64();
        tk[3] += t2 - tmark;
        tmark = t2;
      }
      // ---------------- all lanes: write the new nodes + their hash entries ----------------
      // (no barrier behind this phase: the master goes on to the next pop check meanwhile, see prev_base above; the
      //  lanes' stores are waited for at the barrier that ends that check, before anybody reads a node record)
      if (tid < n_act && s_ev[tid] == EV_NEW && s_rank[tid] < s_n_written) {
        const int L    = tid;
        const int i    = s_src[L];  // child whose data the node ends up with
        const int node = my_base + s_rank[L];
        Node     &pn   = pool[node];
        int       id[3];
        pos_to_index(s_cstate[L], center, inv_res, id);
        for (int q = 0; q < 3; ++q) pn.index[q] = id[q];
        for (int q = 0; q < 6; ++q) pn.state[q] = s_cstate[i][q];
        pn.f = s_cf[i];
        pn.g = s_cg[i];
        for (int q = 0; q < 3; ++q) pn.input[q] = first ? start_a[q] : s_inputs[i][q];
        pn.duration   = tau;
        pn.parent     = my_cur;
        pn.node_state = IN_OPEN_SET;
        pn.time       = new_t;
        pn.time_idx   = new_ti;
        // :387 quirk — insert(pro_id, pro_node->time, ...): double -> int truncation
        hash_insert(htab, hcap, pack_key(id[0], id[1], id[2], (int)new_t), node);
      }
      if (timed) {
        const long long t2 = wall_clock64();
        tk[4] += t2 - tmark;
        tk[5] += 1;
      }
    }
    __syncthreads();  // the last expansion's node records are written
    // broadcast the verdict of this attempt
    if (dropped) break;
    if (tid == ASTAR_MASTER) s_ret = ret;
    __syncthreads();
    const int r = s_ret;
    __syncthreads();
    if (r != NO_PATH) break;
  }
  if (spec) {
    // first attempt: announce the verdict; a NO_PATH first attempt leaves the agent's outputs to the second one.
    // second attempt: wait for that verdict (bounded) and go on only if the first attempt failed.
    if (!second) {
      if (tid == ASTAR_MASTER) {
        s_ret = ret;
        __hip_atomic_store(&wsp.verdict[agent], vbase + (ret != NO_PATH ? 1 : 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (s_ret == NO_PATH) return false;
    } else {
      if (dropped) return false;
      if (tid == ASTAR_MASTER) {
        const long long t0 = wall_clock64();
        int             v;
        for (;;) {
          v = __hip_atomic_load(&wsp.verdict[agent], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - vbase;
          if (v == 1 || v == 2) break;
          flow_pause();
          if (wall_clock64() - t0 > FLOW_TIMEOUT_TICKS) {
            if (flow_err) atomicExch(flow_err, 4);
            v = 0;
            break;
          }
        }
        s_ret = v;
      }
      __syncthreads();
      if (s_ret != 2) return false;
      if (tid == ASTAR_MASTER) searches = 2;  // as the sequential pattern counts them
    }
  }

  // ---------------- master: getPathWithVel(corridor_tau) (:663-694) ----------------
  if (tid == ASTAR_MASTER) {
    int n = 0;
    if (ret != NO_PATH && ret != SEARCH_ERR && terminal >= 0) {
      double *route = out_route + (size_t)agent * route_cap * 6;
      // walk back from the terminal node; points are produced last-to-first, then reversed
      int    node     = terminal;
      double t_node   = 0, t_sample = corridor_tau;
      for (int q = 0; q < 6; ++q) route[q] = pool[node].state[q];
      n = 1;
      while (pool[node].parent >= 0) {
        const Node  &nd       = pool[node];
        const double duration = nd.duration;
        const Node  &par      = pool[nd.parent];
        t_node                = duration;
        while (true) {
          if (t_sample > t_node) {
            node = nd.parent;
            t_sample -= t_node;
            break;
          }
          t_node -= t_sample;
          double xt[6];
          state_transit(par.state, xt, nd.input, t_node);
          if (n < route_cap)
            for (int q = 0; q < 6; ++q) route[n * 6 + q] = xt[q];
          ++n;
          t_sample = corridor_tau;
        }
      }
      const int kept = n < route_cap ? n : route_cap;
      for (int i = 0; i < kept / 2; ++i)
        for (int q = 0; q < 6; ++q) {
          const double tmp              = route[i * 6 + q];
          route[i * 6 + q]              = route[(kept - 1 - i) * 6 + q];
          route[(kept - 1 - i) * 6 + q] = tmp;
        }
    }
    // number of nodes on the retrieved path (retrievePath :826-836)
    int n_path = 0;
    if (terminal >= 0) {
      int c  = terminal;
      n_path = 1;
      while (pool[c].parent >= 0) {
        c = pool[c].parent;
        ++n_path;
      }
    }
    out_ret[agent]           = ret;
    out_route_len[agent]     = n;
    out_stats[agent * 4 + 0] = use_node_num;
    out_stats[agent * 4 + 1] = iter_num;
    out_stats[agent * 4 + 2] = n_path;
    out_stats[agent * 4 + 3] = searches;
    if (wsp.dbg)
      for (int k = 0; k < 6; ++k) wsp.dbg[(size_t)agent * 8 + k] = tk[k];
    if (out_trace && n_trace < trace_cap) out_trace[(size_t)agent * trace_cap + n_trace] = -1;
  }
  return true;
}

// One workgroup per (agent, attempt): lane i evaluates motion primitive i.
__global__ __launch_bounds__(ASTAR_THREADS) void k_astar(
    MapView m, SogmAstarParams ap, double corridor_tau, AstarWorkspace wsp,
    const double *__restrict__ start_pva, const double *__restrict__ goal,
    const double *__restrict__ t_start, int32_t *__restrict__ out_ret,
    double *__restrict__ out_route, int32_t *__restrict__ out_route_len, int route_cap,
    int32_t *__restrict__ out_stats, int32_t *__restrict__ out_trace, int trace_cap, int agent0, FlowCtl fc,
    int search_mode) {
  // search_mode bit 3 (dataflow replan): the launch has 2 x n workgroups; workgroup b >= n runs the SECOND attempt of
  // agent b - n speculatively beside the first
  const bool spec   = (search_mode & 8) != 0;
  const int  n_half = spec ? (int)gridDim.x / 2 : (int)gridDim.x;
  const int  second = spec && (int)blockIdx.x >= n_half ? 1 : 0;
  const int  agent  = ((int)blockIdx.x - second * n_half) + agent0;
  const int  tid    = threadIdx.x;
  // dataflow replan: tell the gate kernel that this workgroup holds its CU resources (the corridor kernel's
  // waiting workgroups must not be dispatched before every search is resident, or they could starve it)
  if (fc.reset_gen) {
    // dataflow replan: the control block's reset runs on another stream (under the map update, normally long done): wait
    // for its generation word before touching the block; then this agent's outputs back to "failed, empty record"
    if (tid == 0) {
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(fc.reset_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != fc.reset_epoch) {
        if (wall_clock64() - t0 > FLOW_TIMEOUT_TICKS) break;  // (the finishing kernel's own timeout fails the tick)
        __builtin_amdgcn_s_sleep(32);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!second) {
      if (tid == 0) fc.out_ok[agent] = 0;
      for (int i = tid; i < fc.rec_words; i += ASTAR_THREADS) fc.out_records[(size_t)agent * fc.rec_words + i] = 0;
    }
  }
  if (fc.hdr && tid == 0) {
    atomicAdd(&fc.hdr[FLOW_A_RESIDENT], 1);
    if (!second) {
      fc.ts[agent * 8 + 7] = wall_clock64();
      fc.ts[agent * 8 + 2] = 0;
      fc.ts[agent * 8 + 0] = wall_clock64();
    }
  }
  if (fc.map_ready) {
    // update flow: this tick's maps are being built beside this launch, agent by agent — wait for this agent's (bounded
    // like every wait of the tick: a flow that does not deliver fails the tick, code 16)
    __shared__ int s_map_ok;
    if (tid == 0) {
      const long long t0 = wall_clock64();
      int             ok = 0;
      for (;;) {
        if (__hip_atomic_load(fc.map_ready + agent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fc.map_epoch) {
          ok = 1;
          break;
        }
        if (__hip_atomic_load(&fc.hdr[FLOW_ERR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        if (wall_clock64() - t0 > FLOW_TIMEOUT_TICKS) {
          atomicExch(&fc.hdr[FLOW_ERR], 16);
          break;
        }
        __builtin_amdgcn_s_sleep(32);  // ~1 us: the wait is on the tick's critical path
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      s_map_ok = ok;
      if (!second) fc.ts[agent * 8 + 0] = wall_clock64();  // the search starts now
    }
    __syncthreads();
    if (!s_map_ok) return;
  }
  const bool mine = astar_search_wg(m, ap, corridor_tau, wsp, start_pva, goal, t_start, out_ret, out_route, out_route_len,
                                    route_cap, out_stats, out_trace, trace_cap, agent, second, spec, 0,
                                    fc.hdr ? &fc.hdr[FLOW_ERR] : nullptr, search_mode);
  if (mine && fc.hdr && tid == ASTAR_MASTER) {  // publish the agent, in completion order, to the corridor kernel
    fc.ts[agent * 8 + 1] = wall_clock64();
    __threadfence();
    const int r = atomicAdd(&fc.hdr[FLOW_A_READY_N], 1);
    __hip_atomic_store(fc.a_ready + r, agent, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Flight kernel S (sogm_flight_run): persistent search workgroups.  A ticket is one (agent, attempt) of the agent whose
// map became ready ticket / 2-th (both attempts of an agent are handed out back to back, the first one first, so the
// workgroup running the second can always wait for the first's verdict).  The attempt whose result counts publishes the
// agent to the corridor queue.
__global__ __launch_bounds__(ASTAR_THREADS) void k_flight_search(
    MapView m, SogmAstarParams ap, double corridor_tau, AstarWorkspace wsp, FlightCtl fl,
    const double *__restrict__ start_pva, const double *__restrict__ goal, const double *__restrict__ t_start,
    int32_t *__restrict__ out_ret, double *__restrict__ out_route, int32_t *__restrict__ out_route_len, int route_cap,
    int32_t *__restrict__ out_stats, int spec) {
  __shared__ int s_item;
  const int tid   = threadIdx.x;
  const int per   = spec ? 2 : 1;
  const int total = fl.n_agents * fl.n_ticks * per;
  fl_wg_started(fl, 1);
  for (;;) {
    __syncthreads();  // (s_item of the previous trip has been read by everybody)
    if (tid == 0) s_item = atomicAdd(&fl.hdr[FL_S_TICKET], 1);
    __syncthreads();
    const int t = s_item;
    if (t >= total) break;
    const int agent = fl_wait_item(fl.s_ring, fl.ring_mask, t / per, &fl.hdr[FL_ERR]);
    if (agent < 0) break;
    const int second = spec ? (t & 1) : 0;
    const int k      = fl.tick_of[agent];
    if (tid == 0 && !second) fl.ts[agent * FL_TS + 0] = wall_clock64();
    const bool mine = astar_search_wg(m, ap, corridor_tau, wsp, start_pva, goal, t_start, out_ret, out_route, out_route_len,
                                      route_cap, out_stats, nullptr, 0, agent, second, spec != 0, 4 * (k + 1),
                                      &fl.hdr[FL_ERR], 0);
    if (mine && tid == ASTAR_MASTER) {
      fl.ts[agent * FL_TS + 1] = wall_clock64();
      wq_push(fl.lw, &fl.hdr[FL_LW_TAIL], ((unsigned)WK_CORRIDOR << 28) | (unsigned)agent, SOGM_MAX_PIECES);
    }
  }
}

int launch_flight_search(const MapView &m, const SogmAstarParams &ap, double corridor_tau, const AstarWorkspace &wsp,
                         const FlightCtl &fl, int n_workgroups, const double *start_pva, const double *goal,
                         const double *t_start, int32_t *out_ret, double *out_route, int32_t *out_route_len, int route_cap,
                         int32_t *out_stats, int spec, hipStream_t st) {
  hipLaunchKernelGGL(k_flight_search, dim3(n_workgroups), dim3(ASTAR_THREADS), 0, st, m, ap, corridor_tau, wsp, fl, start_pva,
                     goal, t_start, out_ret, out_route, out_route_len, route_cap, out_stats, spec);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

size_t astar_node_bytes() { return sizeof(Node); }
int    astar_pool_max() { return ASTAR_POOL_MAX; }

int launch_astar(const MapView &m, const SogmAstarParams &ap, double corridor_tau,
                 const AstarWorkspace &wsp, int n_agents, const double *start_pva,
                 const double *goal, const double *t_start, int32_t *out_ret, double *out_route,
                 int32_t *out_route_len, int route_cap, int32_t *out_stats, int32_t *out_trace,
                 int trace_cap, hipStream_t st, int agent0, const FlowCtl *fc, int search_mode) {
  const FlowCtl none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const bool spec = (search_mode & 8) != 0 && wsp.verdict && fc && !out_trace;
  if (!spec) search_mode &= ~8;
  hipLaunchKernelGGL(k_astar, dim3(spec ? 2 * n_agents : n_agents), dim3(ASTAR_THREADS), 0, st, m, ap, corridor_tau, wsp,
                     start_pva, goal, t_start, out_ret, out_route, out_route_len, route_cap,
                     out_stats, out_trace, trace_cap, agent0, fc ? *fc : none, search_mode);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// How many k_astar workgroups the device holds at once.  The speculative second attempts (workgroups A .. 2A - 1)
// wait for the first attempts' verdicts: that is only free of a dispatch-order assumption while all 2 A are resident.
int astar_resident_workgroups(int device) {
  int per_cu = 0, n_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_astar, ASTAR_THREADS, 0) != hipSuccess ||
      hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess)
    return 0;
  return per_cu * n_cu;
}

}  // namespace sogm
