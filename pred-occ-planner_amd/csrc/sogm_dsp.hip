// sogm_dsp.hip — batched particle-filter SOGM (dsp_map::DSPMap, plan_env/include/plan_env/dsp_dynamic.h)
// for gfx950, behind sogm_dsp_* / sogm_update_dsp in include/sogm_abi.h.
//
// The reference sweeps the voxels sequentially and lets every particle take the first empty slot of
// its destination voxel / FOV pyramid at the moment it is processed.  That order dependence is the
// only coupling between particles, so it is reproduced exactly WITHOUT a sequential sweep:
//
//   * every particle's new state is a pure function of its old state (LIMIT_MOVEMENT_IN_XY_PLANE
//     keeps vz == 0, so the velocity-noise branches of mapPrediction/moveParticle never draw:
//     |vx*vy*vz| < 1e-6 always) -> one thread per slot (k_dsp_predict);
//   * the sweep order is the key  src_voxel*16 + src_slot.  For a destination voxel B the events in
//     time order are: arrivals with key < B*16 (B still holds all of its own particles), B's own
//     departures, arrivals with key > B*16.  Arrivals are linked into a per-voxel list; one thread
//     per voxel keeps the <= S smallest keys of either phase and replays "lowest empty slot"
//     (k_dsp_place).  Only S arrivals per phase can ever be placed, so the selection is exact;
//   * pyramid lists hold the first SP (=20) placed particles in key order (k_dsp_pyramids); a later
//     one vanishes (moveParticle returns -2), which frees its slot for later arrivals — resolved by
//     iterating place/pyramids to the fixed point (monotone; rounds after convergence exit at once);
//   * new-born particles: Gaussian-table offsets are closed-form prefix sums over (point, particle)
//     order (k_dsp_newborn, block scan), slots again by ordered first-fit (k_dsp_place_born).
//
// Data layout (HBM, per agent): particle store as SoA with 16 slots per voxel —
//   flag u8 [V][16] (one 16-B load tells whether a voxel holds anything) and six fp32 planes
//   vx, vy, px, py, pz, w [V][16] (one 64-B line per voxel and field, touched only where occupied);
//   future-occupancy accumulators fut[T][V] in the SOGM's own time-major layout, so that publishing
//   is a straight copy.  vz and the per-particle update time are not stored (always 0 / never read).
// Everything is HBM/latency-bound gather-scatter work; no MFMA.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "sogm_device.hpp"

namespace sogm {

enum : uint8_t { F_EMPTY = 0, F_VALID = 1, F_RESAMP = 2, F_MOVED = 3, F_NEW = 4, F_DEPART = 5, F_KILLED = 6 };
#define DSP_SLOTS 16   // storage slots per voxel (>= SAFE_PARTICLE_NUM_VOXEL)
#define DSP_ROUNDS 6   // place/pyramid fixed-point rounds issued per update

#define VEL_MAX_CLUSTERS 256  // accepted clusters per frame (more: error counter)
#define VEL_MAX_DYN 64        // possibly-dynamic clusters per frame taking part in the association
#define VEL_ADJ_CAP 128       // neighbours within the cluster tolerance kept per point

struct DspAgent {  // per-agent scalars (device)
  float  last_p[3];
  int    first;
  double last_t;
  float  cur[3];
  float  update_time;
  float  quat[4];
  float  odom[4];  // -dx, -dy, -dz, dt
  int    pseq, vseq, rseq;
  int    ok;
  int    n_born, n_obs, valid_points;
  float  enb;  // expected_new_born_objects
  float  w_new;
  int    cand_cnt, born_cnt;
  int    changed[DSP_ROUNDS];
  int    dbg_voxel_full, dbg_pyr_full, dbg_out;
  int    err_unconverged, err_pool, err_points;
  int    n_occupied;
  // velocity estimation (velocityEstimationThread): previous frame's possibly-dynamic clusters and counters
  int    vel_n_last;
  int    vel_clusters, vel_dynamic, vel_matched, vel_err;
  float  vel_last[VEL_MAX_DYN][5];  // centre x, y, z, intensity, point_num (as float)
};

struct DspDev {
  int   A, V, S, L, W, H, T, NP, nph, npv, SP, OM, cand_cap, max_pts, nb, maxp, n_gauss, n_rand;
  float res, hx, hy, hz, sigma, Pd, kappa, w_nb, thick;
  float pred_t[SOGM_DSP_MAX_T];
  DspAgent *ag;
  uint8_t  *flag;  // [A][V][16]
  float    *f[6];  // vx vy px py pz w : [A][V][16]
  float    *fut;   // [A][T][V]
  float    *occ;   // [A][4][V]
  float    *pc;    // [A][NP][OM][5]
  int      *nobs;  // [A][NP]
  int      *maxlen;  // [A][NP] float bits (>= 0) or -1
  int      *obs_list;  // [A][max_pts] pi*OM + seq of every stored observation
  float    *bp_h, *bp_v;  // [A][(nph+1)*3], [A][(npv+1)*3]
  float    *born;  // [A][max_pts][7]
  unsigned short *vel_adj;  // [A][max_pts][VEL_ADJ_CAP] neighbour lists of the clustering (velocity estimation)
  int            *vel_deg;  // [A][max_pts]
  // candidates of the prediction step (movers + in-FOV stays)
  int     *c_key, *c_dest, *c_pyr, *c_assign, *c_vnext, *c_pnext;  // [A][cand_cap]
  uint8_t *c_kill;                                                 // [A][cand_cap]
  float   *c_pay;                                                  // [A][cand_cap][6]
  int     *vhead;  // [A][V]
  int     *phead;  // [A][NP]
  int     *pyr_list, *pyr_n;  // [A][NP][SP] loc, [A][NP]
  int     *pyr_key, *pyr_id;  // [A][NP][SP] selection scratch (sweep key, candidate id)
  // new-born scratch
  int   *b_valid, *b_nstatic, *b_vi;  // [A][max_pts]
  float *b_c;                          // [A][max_pts][3]
  int8_t *b_cls;                       // [A][max_pts*nb]
  int   *b_dest, *b_vnext;             // [A][max_pts*nb]
  float *b_pay;                        // [A][max_pts*nb][5] px py pz vx vy
  const float *pg, *vg, *pdf, *bp_ori_h, *bp_ori_v;
  const int   *rnd, *nbr;
};

// ---- small device helpers (arithmetic mirrors the reference, see oracle/dsp_oracle.cpp) ---------
__device__ inline void qmul(const float a[4], const float b[4], float o[4]) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
  o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}
// rotateVectorByQuaternion dsp_dynamic.h:1391-1411
__device__ inline void rotate_q(const float *v, const float *q, float *o) {
  float vq[4] = {0.f, v[0], v[1], v[2]}, t[4], r[4];
  float n2    = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  float inv[4] = {q[0] / n2, -q[1] / n2, -q[2] / n2, -q[3] / n2};
  qmul(q, vq, t);
  qmul(t, inv, r);
  o[0] = r[1];
  o[1] = r[2];
  o[2] = r[3];
}
__device__ inline float dot3(float x, float y, float z, const float *n) { return x * n[0] + y * n[1] + z * n[2]; }

// ifInPyramidsArea + findPointPyramid{Horizontal,Vertical}Index (:1413-1474): pyramid index or -1
__device__ inline int pyramid_of(const float *bh, const float *bv, int nph, int npv, float x, float y, float z) {
  if (!(dot3(x, y, z, bh) >= 0.f && dot3(x, y, z, bh + nph * 3) <= 0.f && dot3(x, y, z, bv) <= 0.f &&
        dot3(x, y, z, bv + npv * 3) >= 0.f))
    return -1;
  int h = -1, v = -1;
  if (fabsf(x) + fabsf(y) + fabsf(z) < 1e-12f) {
    // degenerate lengths: the products last*t of the reference's scan may underflow — replay it literally
    float last = 1.f;
    for (int i = 0; i < nph; i++) {
      float t = dot3(x, y, z, bh + (i + 1) * 3);
      if (last * t <= 0.f) {
        h = i;
        break;
      }
      last = t;
    }
    last = -1.f;
    for (int j = 0; j < npv; j++) {
      float t = dot3(x, y, z, bv + (j + 1) * 3);
      if (last * t <= 0.f) {
        v = j;
        break;
      }
      last = t;
    }
  } else {
    // The reference scans the boundary planes for the first sign change.  Inside the FOV the dot
    // products are positive (h) / negative (v) up to the point's pyramid and change sign once (planes
    // are 1 degree apart, far more than fp32 rounding), so the first index with t <= 0 (h) / t >= 0 (v)
    // is found by bisection with the same fp32 dot products: ~13 instead of ~70 plane tests.
    int lo = 0, hi = nph - 1;  // predicate true at nph-1 (ifInPyramidsArea: dot with plane nph <= 0)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (dot3(x, y, z, bh + (mid + 1) * 3) <= 0.f)
        hi = mid;
      else
        lo = mid + 1;
    }
    h  = lo;
    lo = 0;
    hi = npv - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (dot3(x, y, z, bv + (mid + 1) * 3) >= 0.f)
        hi = mid;
      else
        lo = mid + 1;
    }
    v = lo;
  }
  if (h < 0 || v < 0) return -1;  // "should not happen" in the reference (:1449,1471)
  return h * npv + v;
}
// getParticleVoxelsIndex (:1152-1166)
__device__ inline int voxel_index(const DspDev &d, float px, float py, float pz) {
  if (px >= d.hx || px <= -d.hx || py >= d.hy || py <= -d.hy || pz >= d.hz || pz <= -d.hz) return -1;
  int x = (int)((px + d.hx) / d.res);
  int y = (int)((py + d.hy) / d.res);
  int z = (int)((pz + d.hz) / d.res);
  int index = z * d.W * d.L + y * d.L + x;
  if (index < 0 || index >= d.V) return -1;
  return index;
}
// queryNormalPDF (:1380-1389)
__device__ inline float query_pdf(const float *pdf, float x, float mu, float sigma) {
  float c = (x - mu) / sigma;
  if (c > 9.9f)
    c = 9.9f;
  else if (c < -9.9f)
    c = -9.9f;
  return pdf[(int)(c * 1000 + 10000)];
}

// ---- update: begin ------------------------------------------------------------------------------
// DSPMap::update :176-229: odometry checks, deltas, boundary-plane rotation.
__global__ void k_dsp_begin(DspDev d, const float *__restrict__ pos, const float *__restrict__ quat,
                            const double *__restrict__ stamps, int32_t *__restrict__ out_ok) {
  const int a  = blockIdx.x;
  DspAgent &s  = d.ag[a];
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    const float  px = pos[a * 3], py = pos[a * 3 + 1], pz = pos[a * 3 + 2];
    const float  qw = quat[a * 4], qx = quat[a * 4 + 1], qy = quat[a * 4 + 2], qz = quat[a * 4 + 3];
    const double t  = stamps[a];
    if (s.first) {
      s.last_p[0] = px;
      s.last_p[1] = py;
      s.last_p[2] = pz;
      s.last_t    = t;
      s.first     = 0;
    }
    int ok = 1;
    if (fabsf(qw) > 1.001f || fabsf(qx) > 1.001f || fabsf(qy) > 1.001f || fabsf(qz) > 1.001f) ok = 0;
    const float dx = px - s.last_p[0], dy = py - s.last_p[1], dz = pz - s.last_p[2];
    const float dt = (float)(t - s.last_t);
    if (ok && (fabsf(dx) > 10.f || fabsf(dy) > 10.f || fabsf(dz) > 10.f || dt < 0.f || dt > 10.f)) ok = 0;
    if (ok) {
      s.cur[0] = s.last_p[0] = px;
      s.cur[1] = s.last_p[1] = py;
      s.cur[2] = s.last_p[2] = pz;
      s.last_t  = t;
      s.quat[0] = qw;
      s.quat[1] = qx;
      s.quat[2] = qy;
      s.quat[3] = qz;
      s.odom[0] = -dx;
      s.odom[1] = -dy;
      s.odom[2] = -dz;
      s.odom[3] = dt;
      s.update_time += dt;
      s.cand_cnt = 0;
      s.born_cnt = 0;
      s.n_obs    = 0;
      for (int r = 0; r < DSP_ROUNDS; ++r) s.changed[r] = 0;
    }
    s.ok = ok;
    if (out_ok) out_ok[a] = ok;
    s_ok = ok;
  }
  __syncthreads();
  if (!s_ok) return;
  const float q[4] = {quat[a * 4], quat[a * 4 + 1], quat[a * 4 + 2], quat[a * 4 + 3]};
  for (int i = threadIdx.x; i < d.nph + 1; i += blockDim.x)
    rotate_q(d.bp_ori_h + i * 3, q, d.bp_h + ((size_t)a * (d.nph + 1) + i) * 3);
  for (int i = threadIdx.x; i < d.npv + 1; i += blockDim.x)
    rotate_q(d.bp_ori_v + i * 3, q, d.bp_v + ((size_t)a * (d.npv + 1) + i) * 3);
}

// ---- update: observation binning (:231-297) -------------------------------------------------------
// One workgroup per agent walks the cloud in input order, 256 points per trip; the slot of a point
// inside its pyramid is (points of earlier trips) + (earlier lanes of this trip in the same pyramid).
__global__ __launch_bounds__(256) void k_dsp_observe(DspDev d, const float *__restrict__ pts,
                                                     const float *__restrict__ labels,
                                                     const int32_t *__restrict__ range) {
  extern __shared__ int s_dyn[];
  const int a = blockIdx.x;
  DspAgent &s = d.ag[a];
  if (!s.ok) return;
  int *s_cnt = s_dyn;          // [NP]
  int *s_max = s_dyn + d.NP;   // [NP] float bits
  __shared__ int   s_pi[256];
  __shared__ float s_bh[181 * 3], s_bv[181 * 3];
  __shared__ int   s_valid, s_nobs;
  for (int i = threadIdx.x; i < d.NP; i += 256) {
    s_cnt[i] = 0;
    s_max[i] = -1;
  }
  for (int i = threadIdx.x; i < (d.nph + 1) * 3; i += 256) s_bh[i] = d.bp_h[(size_t)a * (d.nph + 1) * 3 + i];
  for (int i = threadIdx.x; i < (d.npv + 1) * 3; i += 256) s_bv[i] = d.bp_v[(size_t)a * (d.npv + 1) * 3 + i];
  if (threadIdx.x == 0) {
    s_valid = 0;
    s_nobs  = 0;
  }
  __syncthreads();
  const int begin = range[a * 2];
  int       n     = range[a * 2 + 1] - begin;
  if (n > d.max_pts) {
    n = d.max_pts;
    if (threadIdx.x == 0) s.err_points += 1;
  }
  const float q[4] = {s.quat[0], s.quat[1], s.quat[2], s.quat[3]};
  const float c0 = s.cur[0], c1 = s.cur[1], c2 = s.cur[2];
  float      *born = d.born + (size_t)a * d.max_pts * 7;
  float      *pc   = d.pc + (size_t)a * d.NP * d.OM * 5;
  for (int base = 0; base < n; base += 256) {
    const int k  = base + threadIdx.x;
    int       pi = -1;
    float     r[3] = {0.f, 0.f, 0.f};
    if (k < n) {
      rotate_q(pts + (size_t)(begin + k) * 3, q, r);
      // input_cloud_with_velocity (:1494-1500,1651-1660): rotated + current_position, labels as given
      born[(size_t)k * 7 + 0] = r[0] + c0;
      born[(size_t)k * 7 + 1] = r[1] + c1;
      born[(size_t)k * 7 + 2] = r[2] + c2;
      for (int j = 0; j < 4; ++j)  // labels == NULL: k_dsp_velocity fills (and reorders) the list afterwards
        born[(size_t)k * 7 + 3 + j] = labels ? labels[(size_t)(begin + k) * 4 + j] : 0.f;
      pi = pyramid_of(s_bh, s_bv, d.nph, d.npv, r[0], r[1], r[2]);
    }
    s_pi[threadIdx.x] = pi;
    __syncthreads();
    if (pi >= 0) {
      int rank = 0;
      for (int j = 0; j < (int)threadIdx.x; ++j) rank += (s_pi[j] == pi);
      const int seq = s_cnt[pi] + rank;  // s_cnt is only advanced after the barrier below
      const float len = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
      if (seq < d.OM - 1) {  // slot OM-1 is overwritten by every later point and never read (:287-290)
        float *o = pc + ((size_t)pi * d.OM + seq) * 5;
        o[0]     = r[0];
        o[1]     = r[1];
        o[2]     = r[2];
        o[3]     = 0.f;
        o[4]     = len;
        d.obs_list[(size_t)a * d.max_pts + atomicAdd(&s_nobs, 1)] = pi * d.OM + seq;
      }
      atomicMax(&s_max[pi], __float_as_int(len));
      atomicAdd(&s_valid, 1);
    }
    __syncthreads();
    if (pi >= 0) atomicAdd(&s_cnt[pi], 1);
    __syncthreads();
  }
  for (int i = threadIdx.x; i < d.NP; i += 256) {
    const int c                  = s_cnt[i];
    d.nobs[(size_t)a * d.NP + i]   = c >= d.OM ? d.OM - 1 : c;
    d.maxlen[(size_t)a * d.NP + i] = s_max[i];
  }
  if (threadIdx.x == 0) {
    s.valid_points = s_valid;
    s.n_obs        = s_nobs;
    s.enb          = d.w_nb * (float)s_valid * (float)d.nb;  // :299-300
    if (n > 0) s.n_born = n;  // velocityEstimationThread returns early on an empty cloud (:1488)
  }
}


// ---- velocityEstimationThread (:1487-1678) --------------------------------------------------------------
// One workgroup per agent (see oracle/dsp_oracle.cpp::velocityEstimation for the restated third-party pieces):
//   ground split -> Euclidean clustering of the non-ground points (connected components under "squared distance
//   < tolerance^2": neighbour lists by brute force over the LDS-resident cloud, min-label propagation with pointer
//   jumping; a component's label = its smallest index = PCL's seed) -> clusters ordered like
//   std::sort(rbegin, rend, size <) (libstdc++'s introsort restated, ties included) -> centres summed in ascending
//   index order -> gated association with the previous frame's clusters (optimal assignment, same scan order as
//   the oracle) -> input_cloud_with_velocity rewritten in the reference's order:
//   [possibly-dynamic clusters][ground points][static clusters].  Points of rejected (small) clusters vanish.
namespace vel {
struct Item {
  int size, id;
};
__device__ inline bool less_(const Item &a, const Item &b) { return a.size < b.size; }
__device__ inline void swap_(Item &a, Item &b) {
  const Item t = a;
  a            = b;
  b            = t;
}
// libstdc++ <bits/stl_heap.h> __adjust_heap / __push_heap with comp = less_
__device__ inline void adjust_heap(Item *f, int hole, int len, Item value) {
  const int top = hole;
  int       child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less_(f[child], f[child - 1])) child--;
    f[hole] = f[child];
    hole    = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child   = 2 * (child + 1);
    f[hole] = f[child - 1];
    hole    = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && less_(f[parent], value)) {
    f[hole] = f[parent];
    hole    = parent;
    parent  = (hole - 1) / 2;
  }
  f[hole] = value;
}
__device__ inline void heap_sort(Item *f, int n) {  // __partial_sort(first, last, last): make_heap + sort_heap
  if (n >= 2)
    for (int parent = (n - 2) / 2;; --parent) {
      adjust_heap(f, parent, n, f[parent]);
      if (parent == 0) break;
    }
  for (int last = n; last > 1; --last) {
    const Item v = f[last - 1];
    f[last - 1]  = f[0];
    adjust_heap(f, 0, last - 1, v);
  }
}
__device__ inline void unguarded_linear_insert(Item *f, int last) {
  const Item val = f[last];
  int        next = last - 1;
  while (less_(val, f[next])) {
    f[last] = f[next];
    last    = next;
    --next;
  }
  f[last] = val;
}
__device__ inline void insertion_sort(Item *f, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (less_(f[i], f[first])) {
      const Item val = f[i];
      for (int k = i; k > first; --k) f[k] = f[k - 1];
      f[first] = val;
    } else {
      unguarded_linear_insert(f, i);
    }
  }
}
// std::sort(first, last, less_) of libstdc++ (<bits/stl_algo.h>: __introsort_loop + __final_insertion_sort)
__device__ inline void std_sort(Item *f, int n) {
  if (n <= 0) return;
  int depth = 0;
  for (int t = n; t > 1; t >>= 1) ++depth;  // __lg(n)
  depth *= 2;
  // explicit stack for the recursion on the right part
  int st_first[40], st_last[40], st_depth[40], sp = 0;
  int first = 0, last = n, dl = depth;
  for (;;) {
    while (last - first > 16) {
      if (dl == 0) {
        heap_sort(f + first, last - first);
        break;
      }
      --dl;
      // __unguarded_partition_pivot
      const int mid = first + (last - first) / 2;
      {  // __move_median_to_first(first, first + 1, mid, last - 1)
        const int a = first + 1, b = mid, c = last - 1;
        if (less_(f[a], f[b])) {
          if (less_(f[b], f[c])) swap_(f[first], f[b]);
          else if (less_(f[a], f[c])) swap_(f[first], f[c]);
          else swap_(f[first], f[a]);
        } else if (less_(f[a], f[c])) swap_(f[first], f[a]);
        else if (less_(f[b], f[c])) swap_(f[first], f[c]);
        else swap_(f[first], f[b]);
      }
      int lo = first + 1, hi = last;
      for (;;) {  // __unguarded_partition(first + 1, last, first)
        while (less_(f[lo], f[first])) ++lo;
        --hi;
        while (less_(f[first], f[hi])) --hi;
        if (!(lo < hi)) break;
        swap_(f[lo], f[hi]);
        ++lo;
      }
      // recurse on [lo, last), continue with [first, lo)
      st_first[sp] = lo;
      st_last[sp]  = last;
      st_depth[sp] = dl;
      ++sp;
      last = lo;
    }
    if (sp == 0) break;
    --sp;
    first = st_first[sp];
    last  = st_last[sp];
    dl    = st_depth[sp];
  }
  if (n > 16) {  // __final_insertion_sort
    insertion_sort(f, 0, 16);
    for (int i = 16; i < n; ++i) unguarded_linear_insert(f, i);
  } else {
    insertion_sort(f, 0, n);
  }
}
}  // namespace vel

__global__ __launch_bounds__(1024) void k_dsp_velocity(DspDev d, const int32_t *__restrict__ range, float vres) {
  extern __shared__ __attribute__((aligned(16))) float s_vel[];
  const int a   = blockIdx.x;
  DspAgent &s   = d.ag[a];
  const int tid = threadIdx.x;
  if (!s.ok) return;
  int n = range[a * 2 + 1] - range[a * 2];
  if (n > d.max_pts) n = d.max_pts;
  if (n <= 0) return;  // :1488 — input_cloud_with_velocity and the previous clusters stay as they are
  const int MP = d.max_pts;
  float *sx = s_vel, *sy = sx + MP, *sz = sy + MP;  // non-ground points, compacted in input order
  int   *lab = (int *)(sz + MP);                    // component label (-> cluster slot later)
  int   *aux = lab + MP;                            // sizes per root, then rank of a point inside its cluster
  __shared__ int           s_scan[1024], s_base_ng, s_base_g, s_changed, s_nc, s_err;
  __shared__ vel::Item     s_item[VEL_MAX_CLUSTERS];
  __shared__ int           s_root[VEL_MAX_CLUSTERS], s_size[VEL_MAX_CLUSTERS], s_off[VEL_MAX_CLUSTERS],
      s_dynseq[VEL_MAX_CLUSTERS];
  __shared__ float         s_cx[VEL_MAX_CLUSTERS], s_cy[VEL_MAX_CLUSTERS], s_cz[VEL_MAX_CLUSTERS];
  __shared__ float         s_v[VEL_MAX_DYN][4];  // vx vy vz intensity of the possibly-dynamic clusters
  __shared__ int           s_total_dyn, s_n_static;
  float *born = d.born + (size_t)a * MP * 7;
  if (tid == 0) {
    s_base_ng = 0;
    s_base_g  = 0;
    s_err     = 0;
  }
  __syncthreads();
  // ---- ground split (:1497-1507), order-preserving compaction of both lists; a thread keeps its points' data
  float px[8][3];
  int   cidx[8];  // >= 0: index in the non-ground list; < 0: -(ground rank) - 1
  const int trips = (n + 1023) / 1024;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t >= trips) break;
    const int k = t * 1024 + tid;
    float x = 0.f, y = 0.f, z = 0.f;
    bool  in = k < n, ng = false;
    if (in) {
      x  = born[(size_t)k * 7];
      y  = born[(size_t)k * 7 + 1];
      z  = born[(size_t)k * 7 + 2];
      ng = z > vres;
    }
    // two exclusive scans over the workgroup (non-ground and ground flags)
    const unsigned long long bn = __ballot(in && ng), bg = __ballot(in && !ng);
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    if (lane == 0) {
      s_scan[wave]      = __popcll(bn);
      s_scan[16 + wave] = __popcll(bg);
    }
    __syncthreads();
    int on = 0, og = 0, tn = 0, tg = 0;
    for (int w = 0; w < 16; ++w) {
      on += w < wave ? s_scan[w] : 0;
      og += w < wave ? s_scan[16 + w] : 0;
      tn += s_scan[w];
      tg += s_scan[16 + w];
    }
    const int rn = s_base_ng + on + __popcll(bn & lt), rg = s_base_g + og + __popcll(bg & lt);
    px[t][0] = x;
    px[t][1] = y;
    px[t][2] = z;
    cidx[t]  = !in ? 0x7fffffff : (ng ? rn : -rg - 1);
    if (in && ng) {
      sx[rn] = x;
      sy[rn] = y;
      sz[rn] = z;
    }
    __syncthreads();
    if (tid == 0) {
      s_base_ng += tn;
      s_base_g += tg;
    }
    __syncthreads();
  }
  // (max_points <= 8192 is checked by sogm_dsp_create: the register staging above holds 8 points per thread)
  const int m = s_base_ng, n_ground = s_base_g;
  // ---- neighbour lists: FLANN L2_Simple squared distance, strictly below (float)(tolerance^2)
  const double tol = (double)(2 * vres);
  const float  r2  = (float)(tol * tol);
  unsigned short *adj = d.vel_adj + (size_t)a * MP * VEL_ADJ_CAP;
  int            *deg = d.vel_deg + (size_t)a * MP;
  for (int i = tid; i < m; i += 1024) {
    const float ax = sx[i], ay = sy[i], az = sz[i];
    int         c  = 0;
    for (int j = 0; j < m; ++j) {
      float df = ax - sx[j], d2 = 0.f;
      d2 += df * df;
      df = ay - sy[j];
      d2 += df * df;
      df = az - sz[j];
      d2 += df * df;
      if (d2 < r2 && j != i) {
        if (c < VEL_ADJ_CAP) adj[(size_t)i * VEL_ADJ_CAP + c] = (unsigned short)j;
        ++c;
      }
    }
    if (c > VEL_ADJ_CAP) {
      s_err = 2;  // denser than a voxel-filtered cloud: neighbours were dropped
      c     = VEL_ADJ_CAP;
    }
    deg[i] = c;
    lab[i] = i;
  }
  __syncthreads();
  // ---- connected components: min-label propagation + pointer jumping to the fixed point
  for (int round = 0; round < 4096; ++round) {
    if (tid == 0) s_changed = 0;
    __syncthreads();
    for (int i = tid; i < m; i += 1024) {
      int mn = lab[i];
      const int dg = deg[i];
      for (int q = 0; q < dg; ++q) {
        const int l = lab[adj[(size_t)i * VEL_ADJ_CAP + q]];
        mn          = l < mn ? l : mn;
      }
      if (mn < lab[i]) {
        atomicMin(&lab[lab[i]], mn);  // hook the old root as well
        atomicMin(&lab[i], mn);
        s_changed = 1;
      }
    }
    __syncthreads();
    for (int i = tid; i < m; i += 1024) {
      int l = lab[i];
      while (lab[l] < l) l = lab[l];
      lab[i] = l;
    }
    __syncthreads();
    if (!s_changed) break;
    __syncthreads();
  }
  // ---- sizes, accepted clusters (5 <= size <= 10000) in seed (= root index) order
  for (int i = tid; i < m; i += 1024) aux[i] = 0;
  __syncthreads();
  for (int i = tid; i < m; i += 1024) atomicAdd(&aux[lab[i]], 1);
  __syncthreads();
  if (tid == 0) {
    int nc = 0;
    for (int i = 0; i < m; ++i)
      if (lab[i] == i && aux[i] >= 5 && aux[i] <= 10000) {
        if (nc < VEL_MAX_CLUSTERS) {
          s_root[nc] = i;
          s_size[nc] = aux[i];
        }
        ++nc;
      }
    if (nc > VEL_MAX_CLUSTERS) {
      s_err = 3;
      nc    = VEL_MAX_CLUSTERS;
    }
    // EuclideanClusterExtraction::extract: std::sort(clusters.rbegin(), clusters.rend(), size <)
    for (int k = 0; k < nc; ++k) {  // the reversed sequence
      s_item[k].size = s_size[nc - 1 - k];
      s_item[k].id   = nc - 1 - k;
    }
    vel::std_sort(s_item, nc);
    s_nc = nc;
  }
  __syncthreads();
  const int nc = s_nc;
  // cluster slot c (extraction order) = s_item[nc - 1 - c]; point -> slot through its root
  for (int i = tid; i < m; i += 1024) aux[i] = -1;
  __syncthreads();
  for (int c = tid; c < nc; c += 1024) aux[s_root[s_item[nc - 1 - c].id]] = c;  // slot of a root
  __syncthreads();
  for (int i = tid; i < m; i += 1024) {
    const int sl = aux[lab[i]];     // lab[i] is the point's root; aux[root] the cluster's slot (or -1)
    lab[i]       = sl >= 0 ? sl : 0xFFFF;  // 0xFFFF = in no accepted cluster
  }
  __syncthreads();
  // ---- centres (sums in ascending index order, :1533-1542) and rank of every point inside its cluster
  for (int c = tid; c < nc; c += 1024) {
    float cx = 0.f, cy = 0.f, cz = 0.f;
    int   cnt = 0;
    for (int i = 0; i < m; ++i)
      if (lab[i] == c) {
        cx += sx[i];
        cy += sy[i];
        cz += sz[i];
        aux[i] = cnt;
        ++cnt;
      }
    s_cx[c]   = cx / (float)cnt;
    s_cy[c]   = cy / (float)cnt;
    s_cz[c]   = cz / (float)cnt;
    s_size[c] = cnt;  // now in extraction order
  }
  __syncthreads();
  // ---- possibly-dynamic split, association with the previous frame, output offsets (one lane)
  if (tid == 0) {
    int nd = 0, off = 0;
    for (int c = 0; c < nc; ++c) {
      const bool stat = s_size[c] > 200 || s_cz[c] > 1.5;  // DYNAMIC_CLUSTER_MAX_POINT_NUM / _MAX_CENTER_HEIGHT
      if (!stat && nd >= VEL_MAX_DYN) s_err = 4;
      s_dynseq[c] = (!stat && nd < VEL_MAX_DYN) ? nd : -1;
      if (s_dynseq[c] >= 0) {
        s_off[c] = off;
        off += s_size[c];
        s_v[nd][0] = -10000.f;
        s_v[nd][1] = -10000.f;
        s_v[nd][2] = -10000.f;
        s_v[nd][3] = 0.55f;
        ++nd;
      }
    }
    s_total_dyn = off;
    off += n_ground;
    for (int c = 0; c < nc; ++c)
      if (s_dynseq[c] < 0) {
        s_off[c] = off;
        off += s_size[c];
      }
    s_n_static = off - s_total_dyn - n_ground;
    // association (:1562-1622)
    const int   R = nd, C = s.vel_n_last;
    const float dt = s.odom[3];  // delt_t_from_last_observation (:213)
    int         matched = 0;
    if (R * C > MP) s_err = 5;
    if (R > 0 && C > 0 && R * C <= MP && dt > 0.00001 && dt < 10.0) {
      // cost / gate matrices live in the (now free) coordinate arrays of the dynamic LDS
      float *cost = sx, *gate = sy;  // R * C <= max_pts floats each: guarded just above (error 5 otherwise)
      int    rows[VEL_MAX_DYN];
      {
        int k = 0;
        for (int c = 0; c < nc; ++c)
          if (s_dynseq[c] >= 0) rows[k++] = c;
      }
      const float distance_gate = 1.5f, maximum_velocity = 5.f;
      const int   point_num_gate = 100;
      for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) {
          const int   k  = rows[r];
          const float ex = s_cx[k] - s.vel_last[c][0], ey = s_cy[k] - s.vel_last[c][1], ez = s_cz[k] - s.vel_last[c][2];
          const float dist = sqrtf(ex * ex + ey * ey + ez * ez);
          const int   dn   = s_size[k] - (int)s.vel_last[c][4];
          if ((dn < 0 ? -dn : dn) > point_num_gate || dist >= distance_gate) {
            gate[r * C + c] = 0.f;
            cost[r * C + c] = distance_gate * 5000.f;
          } else {
            gate[r * C + c] = 1.f;
            cost[r * C + c] = dist / distance_gate * 1000.f;
          }
        }
      // minimum-cost assignment, potentials form, rows <= cols (transposed view otherwise): oracle's scan order
      const bool  tr = R > C;
      const int   NR = tr ? C : R, NCc = tr ? R : C;
      const float INF = 3.0e38f;
      float u[VEL_MAX_DYN + 1], v[VEL_MAX_DYN + 1], minv[VEL_MAX_DYN + 1];
      int   pj[VEL_MAX_DYN + 1], way[VEL_MAX_DYN + 1];
      bool  used[VEL_MAX_DYN + 1];
      for (int j = 0; j <= NCc; ++j) {
        v[j]  = 0.f;
        pj[j] = 0;
        way[j] = 0;
      }
      for (int i = 0; i <= NR; ++i) u[i] = 0.f;
      auto at = [&](int i, int j) { return tr ? cost[j * C + i] : cost[i * C + j]; };  // (row i, col j) of the NR x NCc view
      for (int i = 1; i <= NR; ++i) {
        pj[0]  = i;
        int j0 = 0;
        for (int j = 0; j <= NCc; ++j) {
          minv[j] = INF;
          used[j] = false;
        }
        do {
          used[j0]     = true;
          const int i0 = pj[j0];
          float     delta = INF;
          int       j1 = 0;
          for (int j = 1; j <= NCc; ++j)
            if (!used[j]) {
              const float cur = at(i0 - 1, j - 1) - u[i0] - v[j];
              if (cur < minv[j]) {
                minv[j] = cur;
                way[j]  = j0;
              }
              if (minv[j] < delta) {
                delta = minv[j];
                j1    = j;
              }
            }
          for (int j = 0; j <= NCc; ++j)
            if (used[j]) {
              u[pj[j]] += delta;
              v[j] -= delta;
            } else {
              minv[j] -= delta;
            }
          j0 = j1;
        } while (pj[j0] != 0);
        do {
          const int j1 = way[j0];
          pj[j0]       = pj[j1];
          j0           = j1;
        } while (j0);
      }
      for (int j = 1; j <= NCc; ++j) {
        if (pj[j] <= 0) continue;
        const int r = tr ? j - 1 : pj[j] - 1, c = tr ? pj[j] - 1 : j - 1;  // new cluster r <-> old cluster c
        if (!(gate[r * C + c] > 0.01f)) continue;
        const int k = rows[r];
        float vx = (s_cx[k] - s.vel_last[c][0]) / dt, vy = (s_cy[k] - s.vel_last[c][1]) / dt,
              vz = (s_cz[k] - s.vel_last[c][2]) / dt;
        const float vv = sqrtf(vx * vx + vy * vy + vz * vz);
        if (vv > maximum_velocity) vx = vy = vz = 0.f;
        s_v[r][0] = vx;
        s_v[r][1] = vy;
        s_v[r][2] = vz;
        s_v[r][3] = s.vel_last[c][3];
        ++matched;
      }
    }
    // clusters_feature_vector_dynamic_last = clusters_feature_vector_dynamic (:1665)
    {
      int k = 0;
      for (int c = 0; c < nc; ++c)
        if (s_dynseq[c] >= 0) {
          s.vel_last[k][0] = s_cx[c];
          s.vel_last[k][1] = s_cy[c];
          s.vel_last[k][2] = s_cz[c];
          s.vel_last[k][3] = s_v[k][3];
          s.vel_last[k][4] = (float)s_size[c];
          ++k;
        }
      s.vel_n_last = k;
    }
    s.vel_clusters = nc;
    s.vel_dynamic  = nd;
    s.vel_matched  = matched;
    if (s_err) s.vel_err = s_err;
    s.n_born = s_total_dyn + n_ground + s_n_static;
  }
  __syncthreads();
  // ---- input_cloud_with_velocity in the reference's order (every thread writes the points it read)
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t >= trips) break;
    const int ci = cidx[t];
    if (ci == 0x7fffffff) continue;
    int   pos;
    float lv[4] = {0.f, 0.f, 0.f, 0.f};
    if (ci < 0) {
      pos = s_total_dyn + (-ci - 1);
    } else {
      const int c = lab[ci];
      if (c == 0xFFFF) continue;  // no accepted cluster: the point is dropped
      pos = s_off[c] + aux[ci];
      if (s_dynseq[c] >= 0)
        for (int j = 0; j < 4; ++j) lv[j] = s_v[s_dynseq[c]][j];
    }
    float *o = born + (size_t)pos * 7;
    o[0]     = px[t][0];
    o[1]     = px[t][1];
    o[2]     = px[t][2];
    for (int j = 0; j < 4; ++j) o[3 + j] = lv[j];
  }
}

// ---- update: prediction (:663-748, moveParticle :1295-1372) ------------------------------------------
// One thread per voxel: a single 16-B load of the slot flags rejects empty voxels (the vast majority);
// occupied slots are then processed independently of each other.
__global__ __launch_bounds__(256) void k_dsp_predict(DspDev d) {
  const int a = blockIdx.y;
  DspAgent &s = d.ag[a];
  if (!s.ok) return;
  __shared__ float s_bh[181 * 3], s_bv[181 * 3];
  for (int i = threadIdx.x; i < (d.nph + 1) * 3; i += 256) s_bh[i] = d.bp_h[(size_t)a * (d.nph + 1) * 3 + i];
  for (int i = threadIdx.x; i < (d.npv + 1) * 3; i += 256) s_bv[i] = d.bp_v[(size_t)a * (d.npv + 1) * 3 + i];
  __syncthreads();
  const int    v   = blockIdx.x * 256 + threadIdx.x;
  const size_t at0 = ((size_t)a * d.V + (v < d.V ? v : 0)) * DSP_SLOTS;
  uint4        f4  = {0u, 0u, 0u, 0u};
  if (v < d.V) f4 = *reinterpret_cast<const uint4 *>(d.flag + at0);
  if (!__any((f4.x | f4.y | f4.z | f4.w) != 0u)) return;  // wave-uniform: whole wave of empty voxels
  const unsigned w4[4] = {f4.x, f4.y, f4.z, f4.w};
  const float    ox = s.odom[0], oy = s.odom[1], oz = s.odom[2], dt = s.odom[3];
  const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Compact the wave's occupied (voxel, slot) pairs into LDS, then process them 64 at a time: one round
  // of memory latency per 64 particles instead of one per slot index.
  __shared__ unsigned short s_list[4][64 * DSP_SLOTS];
  int n_act = 0;
#pragma unroll
  for (int p = 0; p < DSP_SLOTS; ++p) {
    const uint8_t            c   = (uint8_t)((w4[p >> 2] >> ((p & 3) * 8)) & 0xffu);
    const bool               act = (c == F_VALID || c == F_RESAMP);
    const unsigned long long m   = __ballot(act);
    if (act) s_list[wave][n_act + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(lane * DSP_SLOTS + p);
    n_act += __popcll(m);
  }
  __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS is in-order per wave; keep the compiler honest
  const int v0 = v - lane;
  for (int base = 0; base < n_act; base += 64) {
    const bool act = base + lane < n_act;
    const int  e   = act ? s_list[wave][base + lane] : 0;
    const int  vv = v0 + (e >> 4), p = e & (DSP_SLOTS - 1);
    const size_t at  = ((size_t)a * d.V + (act ? vv : 0)) * DSP_SLOTS + p;
    const int    gid = vv * DSP_SLOTS + p;
    float        vx = 0.f, vy = 0.f, px = 0.f, py = 0.f, pz = 0.f;
    int          nv = -1, pyr = -1;
    bool         need = false;
    if (act) {
      vx = d.f[0][at];
      vy = d.f[1][at];
      px = d.f[2][at];
      py = d.f[3][at];
      pz = d.f[4][at];
      px += dt * vx + ox;
      py += dt * vy + oy;
      pz += dt * 0.f + oz;
      d.f[2][at] = px;
      d.f[3][at] = py;
      d.f[4][at] = pz;
      nv = voxel_index(d, px, py, pz);
      if (nv < 0) {  // moved out (:738-741): the slot frees up at this particle's own turn in the sweep
        d.flag[at] = F_DEPART;
        atomicAdd(&s.dbg_out, 1);
      } else {
        pyr = pyramid_of(s_bh, s_bv, d.nph, d.npv, px, py, pz);
        if (nv == vv) {
          d.flag[at] = F_VALID;
          need       = pyr >= 0;
        } else {
          d.flag[at] = F_DEPART;
          need       = true;
        }
      }
    }
    // one counter bump per wave instead of one same-address atomic per particle
    const unsigned long long m = __ballot(need);
    if (!m) continue;
    int base_c = 0;
    if (lane == __ffsll((long long)m) - 1) base_c = atomicAdd(&s.cand_cnt, __popcll(m));
    base_c = __shfl(base_c, __ffsll((long long)m) - 1);
    if (!need) continue;
    const int c_i = base_c + __popcll(m & ((1ull << lane) - 1ull));
    if (c_i >= d.cand_cap) {
      atomicAdd(&s.err_pool, 1);
      if (nv != vv) d.flag[at] = F_EMPTY;
      continue;
    }
    const size_t ci = (size_t)a * d.cand_cap + c_i;
    d.c_key[ci]    = gid;
    d.c_dest[ci]   = nv;
    d.c_pyr[ci]    = pyr;
    d.c_kill[ci]   = 0;
    if (nv == vv) {
      d.c_assign[ci] = p;   // stay: slot known
      d.c_vnext[ci]  = -2;  // marks "stay"
    } else {
      d.c_assign[ci] = -1;
      float *pay     = d.c_pay + ci * 6;
      pay[0]         = vx;
      pay[1]         = vy;
      pay[2]         = px;
      pay[3]         = py;
      pay[4]         = pz;
      pay[5]         = d.f[5][at];
      d.c_vnext[ci]  = atomicExch(&d.vhead[(size_t)a * d.V + nv], c_i);
    }
    if (pyr >= 0) d.c_pnext[ci] = atomicExch(&d.phead[(size_t)a * d.NP + pyr], c_i);
  }
}

// ---- update: ordered first-fit of arrivals, one thread per destination voxel ---------------------
__global__ __launch_bounds__(64) void k_dsp_place(DspDev d, int round) {
  const int a = blockIdx.y;
  DspAgent &s = d.ag[a];
  if (!s.ok) return;
  if (round > 0 && s.changed[round - 1] == 0) return;
  const int B = blockIdx.x * 64 + threadIdx.x;
  if (B >= d.V) return;
  int head = d.vhead[(size_t)a * d.V + B];
  if (head < 0) return;
  __shared__ int s_key[64][2 * DSP_SLOTS], s_id[64][2 * DSP_SLOTS];
  int *kb = s_key[threadIdx.x], *ib = s_id[threadIdx.x];  // [0,S): before, [16,16+S): after
  int  nb_ = 0, na_ = 0;
  const int    S = d.S, lo = B * DSP_SLOTS;
  const size_t cb = (size_t)a * d.cand_cap;
  for (int m = head; m >= 0; m = d.c_vnext[cb + m]) {
    d.c_assign[cb + m] = -1;
    if (d.c_kill[cb + m]) continue;
    const int key = d.c_key[cb + m];
    int *K, *I, *N;
    if (key < lo) {
      K = kb;
      I = ib;
      N = &nb_;
    } else {
      K = kb + DSP_SLOTS;
      I = ib + DSP_SLOTS;
      N = &na_;
    }
    // keep the S smallest keys, sorted ascending
    int n = *N;
    if (n == S && key > K[S - 1]) continue;
    int j = n < S ? n : S - 1;
    while (j > 0 && K[j - 1] > key) {
      K[j] = K[j - 1];
      I[j] = I[j - 1];
      --j;
    }
    K[j] = key;
    I[j] = m;
    if (n < S) *N = n + 1;
  }
  // occupancy of B before its own turn: everything non-empty (incl. departing / vanishing stays)
  const uint8_t *fl = d.flag + ((size_t)a * d.V + B) * DSP_SLOTS;
  unsigned occ = 0, leaving = 0;
  for (int p = 0; p < S; ++p) {
    const uint8_t c = fl[p];
    if (c != F_EMPTY) occ |= 1u << p;
    if (c == F_DEPART || c == F_KILLED) leaving |= 1u << p;
  }
  const unsigned all = (1u << S) - 1u;
  for (int i = 0; i < nb_; ++i) {
    const unsigned fr = ~occ & all;
    if (!fr) break;
    const int sl = __ffs(fr) - 1;
    occ |= 1u << sl;
    d.c_assign[cb + ib[i]] = sl;
  }
  occ &= ~leaving;
  for (int i = 0; i < na_; ++i) {
    const unsigned fr = ~occ & all;
    if (!fr) break;
    const int sl = __ffs(fr) - 1;
    occ |= 1u << sl;
    d.c_assign[cb + ib[DSP_SLOTS + i]] = sl;
  }
}

// ---- update: pyramid lists = first SP placed particles in sweep order, one thread per pyramid -----
__global__ __launch_bounds__(64) void k_dsp_pyramids(DspDev d, int round) {
  const int a = blockIdx.y;
  DspAgent &s = d.ag[a];
  if (!s.ok) return;
  if (round > 0 && s.changed[round - 1] == 0) return;
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= d.NP) return;
  const int head = d.phead[(size_t)a * d.NP + q];
  int      *out  = d.pyr_list + ((size_t)a * d.NP + q) * d.SP;
  if (head < 0) {
    d.pyr_n[(size_t)a * d.NP + q] = 0;
    return;
  }
  // selection scratch in HBM/L2: SAFE_PARTICLE_NUM_PYRAMID grows with the grid (20 at 66x66x20, 218 at
  // 100^3) while a pyramid rarely holds more than a handful of particles
  int *K = d.pyr_key + ((size_t)a * d.NP + q) * d.SP, *I = d.pyr_id + ((size_t)a * d.NP + q) * d.SP;
  int  n = 0, placed = 0;
  const int    SP = d.SP;
  const size_t cb = (size_t)a * d.cand_cap;
  for (int m = head; m >= 0; m = d.c_pnext[cb + m]) {
    if (d.c_kill[cb + m] || d.c_assign[cb + m] < 0) continue;
    ++placed;
    const int key = d.c_key[cb + m];
    if (n == SP && key > K[SP - 1]) continue;
    int j = n < SP ? n : SP - 1;
    while (j > 0 && K[j - 1] > key) {
      K[j] = K[j - 1];
      I[j] = I[j - 1];
      --j;
    }
    K[j] = key;
    I[j] = m;
    if (n < SP) n = n + 1;
  }
  for (int i = 0; i < n; ++i) out[i] = d.c_dest[cb + I[i]] * DSP_SLOTS + d.c_assign[cb + I[i]];
  d.pyr_n[(size_t)a * d.NP + q] = n;
  if (placed > SP) {  // the rest found the list full: they vanish (moveParticle -> -2)
    const int last = K[SP - 1];
    int       newly = 0;
    for (int m = head; m >= 0; m = d.c_pnext[cb + m]) {
      if (d.c_kill[cb + m] || d.c_assign[cb + m] < 0) continue;
      if (d.c_key[cb + m] > last) {
        d.c_kill[cb + m] = 1;
        if (d.c_vnext[cb + m] == -2)  // a stay: its slot frees up at its own turn
          d.flag[(size_t)a * d.V * DSP_SLOTS + d.c_key[cb + m]] = F_KILLED;
        ++newly;
      }
    }
    if (newly) atomicAdd(&s.changed[round], newly);
  }
}

// ---- update: commit ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dsp_commit_slots(DspDev d) {
  const int a = blockIdx.y;
  DspAgent &s = d.ag[a];
  if (!s.ok) return;
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v == 0 && s.changed[DSP_ROUNDS - 1] != 0) s.err_unconverged += 1;
  if (v >= d.V) return;
  const size_t at0 = ((size_t)a * d.V + v) * DSP_SLOTS;
  const uint4  f4  = *reinterpret_cast<const uint4 *>(d.flag + at0);
  // bytes >= 5 are the transient codes F_DEPART / F_KILLED: (b + 3) & 8 is set exactly for 5, 6 (codes <= 6)
  const unsigned w4[4] = {f4.x, f4.y, f4.z, f4.w};
  unsigned       any = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) any |= (w4[k] + 0x03030303u) & 0x08080808u;
  if (!any) return;
#pragma unroll
  for (int p = 0; p < DSP_SLOTS; ++p) {
    const uint8_t c = (uint8_t)((w4[p >> 2] >> ((p & 3) * 8)) & 0xffu);
    if (c == F_DEPART) d.flag[at0 + p] = F_EMPTY;
    if (c == F_KILLED) {
      d.flag[at0 + p] = F_EMPTY;
      atomicAdd(&s.dbg_pyr_full, 1);
    }
  }
}
__global__ __launch_bounds__(256) void k_dsp_commit_movers(DspDev d) {
  const int a = blockIdx.y;
  DspAgent &s = d.ag[a];
  if (!s.ok) return;
  const int m = blockIdx.x * 256 + threadIdx.x;
  const int n = s.cand_cnt < d.cand_cap ? s.cand_cnt : d.cand_cap;
  if (m >= n) return;
  const size_t ci = (size_t)a * d.cand_cap + m;
  if (d.c_vnext[ci] == -2) return;  // stays live in place
  if (d.c_kill[ci]) {
    atomicAdd(&s.dbg_pyr_full, 1);
    return;
  }
  const int sl = d.c_assign[ci];
  if (sl < 0) {
    atomicAdd(&s.dbg_voxel_full, 1);
    return;
  }
  const size_t at  = ((size_t)a * d.V + d.c_dest[ci]) * DSP_SLOTS + sl;
  const float *pay = d.c_pay + ci * 6;
  d.flag[at]       = F_MOVED;
  for (int k = 0; k < 6; ++k) d.f[k][at] = pay[k];
}

// ---- update: mapUpdate (:750-849) --------------------------------------------------------------------
// C_k + kappa per stored observation: neighbour pyramids in table order, list entries in sweep order.
__global__ __launch_bounds__(256) void k_dsp_ck(DspDev d) {
  const int a = blockIdx.y;
  DspAgent &s = d.ag[a];
  if (!s.ok) return;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= s.n_obs) return;
  const int    e  = d.obs_list[(size_t)a * d.max_pts + k];
  const int    pi = e / d.OM;
  float       *o  = d.pc + ((size_t)a * d.NP * d.OM + e) * 5;
  const float  ox = o[0], oy = o[1], oz = o[2];
  float        ck = 0.f;
  const int   *nb = d.nbr + pi * 10;
  const size_t sb = (size_t)a * d.V * DSP_SLOTS;
  float        sig = d.sigma;
  for (int n = 0; n < nb[0]; ++n) {
    const int  q  = nb[n + 1];
    const int  cn = d.pyr_n[(size_t)a * d.NP + q];
    const int *li = d.pyr_list + ((size_t)a * d.NP + q) * d.SP;
    for (int i = 0; i < cn; ++i) {
      const size_t at = sb + li[i];
      const float  gk = query_pdf(d.pdf, d.f[2][at], ox, sig) * query_pdf(d.pdf, d.f[3][at], oy, sig) *
                       query_pdf(d.pdf, d.f[4][at], oz, sig);
      ck += d.Pd * d.f[5][at] * gk;
    }
  }
  ck += (s.enb + d.kappa);
  o[3] = ck;
}
// weight update of every particle listed in a pyramid
__global__ __launch_bounds__(256) void k_dsp_weight(DspDev d) {
  const int a = blockIdx.y;
  DspAgent &s = d.ag[a];
  if (!s.ok) return;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= d.NP * d.SP) return;
  const int i = gid / d.SP, q = gid % d.SP;
  if (q >= d.pyr_n[(size_t)a * d.NP + i]) return;
  const size_t at = (size_t)a * d.V * DSP_SLOTS + d.pyr_list[((size_t)a * d.NP + i) * d.SP + q];
  const float  px = d.f[2][at], py = d.f[3][at], pz = d.f[4][at];
  const float  len = sqrtf(px * px + py * py + pz * pz);
  const int    mb  = d.maxlen[(size_t)a * d.NP + i];
  const float  ml  = mb < 0 ? -1.f : __int_as_float(mb);
  if (ml > 0.f && len > ml + d.thick) return;  // occluded
  const int   *nb = d.nbr + i * 10;
  const float *pc = d.pc + (size_t)a * d.NP * d.OM * 5;
  float        sum = 0.f, sig = d.sigma;
  for (int n = 0; n < nb[0]; ++n) {
    const int ni = nb[n + 1];
    const int cn = d.nobs[(size_t)a * d.NP + ni];
    for (int z = 0; z < cn; ++z) {
      const float *o  = pc + ((size_t)ni * d.OM + z) * 5;
      const float  gk = query_pdf(d.pdf, px, o[0], sig) * query_pdf(d.pdf, py, o[1], sig) *
                       query_pdf(d.pdf, pz, o[2], sig);
      sum += d.Pd * gk / o[3];
    }
  }
  d.f[5][at] *= ((1 - d.Pd) + sum);
}

// ---- update: new-born particles (:852-990) --------------------------------------------------------------
__device__ inline int block_scan_excl(int v, int *s_tmp, int *total) {  // blockDim = 1024
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int       x    = v;
  for (int o = 1; o < 64; o <<= 1) {
    int y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  if (lane == 63) s_tmp[w] = x;
  __syncthreads();
  if (w == 0) {
    int t = lane < 16 ? s_tmp[lane] : 0;
    for (int o = 1; o < 16; o <<= 1) {
      int y = __shfl_up(t, o);
      if (lane >= o) t += y;
    }
    if (lane < 16) s_tmp[lane] = t;
  }
  __syncthreads();
  const int base = w ? s_tmp[w - 1] : 0;
  if (total) *total = s_tmp[15];
  __syncthreads();
  return base + x - v;
}

__global__ __launch_bounds__(1024) void k_dsp_newborn(DspDev d) {
  extern __shared__ float s_inv[];  // [max_pts] 1/C_k in (pyramid, j) order
  __shared__ int   s_tmp[16];
  __shared__ float s_wnew;
  const int a = blockIdx.x;
  DspAgent &s = d.ag[a];
  if (!s.ok) return;
  const int tid = threadIdx.x;
  // normalisation coefficient: sequential fp32 sum in (pyramid, j) order (:854-860)
  {
    const int per = (d.NP + 1023) / 1024;
    int       cnt = 0;
    for (int i = tid * per; i < (tid + 1) * per && i < d.NP; ++i) cnt += d.nobs[(size_t)a * d.NP + i];
    int       tot;
    int       off = block_scan_excl(cnt, s_tmp, &tot);
    const float *pc = d.pc + (size_t)a * d.NP * d.OM * 5;
    for (int i = tid * per; i < (tid + 1) * per && i < d.NP; ++i) {
      const int c = d.nobs[(size_t)a * d.NP + i];
      for (int j = 0; j < c; ++j) s_inv[off++] = 1.f / pc[((size_t)i * d.OM + j) * 5 + 3];
    }
    __syncthreads();
    if (tid == 0) {
      float norm = 0.f;
      for (int i = 0; i < tot; ++i) norm += s_inv[i];
      s_wnew  = d.w_nb * norm;
      s.w_new = s_wnew;
    }
    __syncthreads();
  }
  const int n = s.n_born, nb = d.nb;
  const int min_static = (int)((float)nb * 0.15f), model_gen = (int)((float)nb * 0.8f);
  const float *born = d.born + (size_t)a * d.max_pts * 7;
  int   *b_valid = d.b_valid + (size_t)a * d.max_pts, *b_nst = d.b_nstatic + (size_t)a * d.max_pts;
  float *b_c = d.b_c + (size_t)a * d.max_pts * 3;
  const size_t sb = (size_t)a * d.V * DSP_SLOTS;
  // per point: voxel, Dempster-Shafer split (:880-925)
  for (int k = tid; k < n; k += 1024) {
    const float cx = born[k * 7] - s.cur[0], cy = born[k * 7 + 1] - s.cur[1], cz = born[k * 7 + 2] - s.cur[2];
    b_c[k * 3] = cx;
    b_c[k * 3 + 1] = cy;
    b_c[k * 3 + 2] = cz;
    const int vi = voxel_index(d, cx, cy, cz);
    int       ns = 0;
    if (vi >= 0) {
      float ws = 0.f, wd = 0.f, wsd = 0.f;
      for (int kk = 0; kk < d.S; ++kk) {
        const size_t  at = sb + (size_t)vi * DSP_SLOTS + kk;
        const uint8_t c  = d.flag[at];
        if (c == F_VALID || c == F_MOVED) {
          const float va = fabsf(d.f[0][at]) + fabsf(d.f[1][at]) + fabsf(0.f);
          const float w  = d.f[5][at];
          if (va < 0.1f)
            ws += w;
          else if (va < 0.5f)
            wsd += w;
          else
            wd += w;
        }
      }
      const float tot = ws + wd + wsd;
      const float m_s = ws / tot, m_d = wd / tot, m_sd = wsd / tot;
      const float p_s = (m_s + m_s + m_sd) * 0.5f, p_d = (m_d + m_d + m_sd) * 0.5f;
      const float psn = p_s / (p_s + p_d);
      const float fs  = (float)model_gen * psn;
      ns              = (fs == fs) ? (int)fs : min_static;
      ns              = ns > min_static ? ns : min_static;
    }
    b_valid[k] = vi >= 0;
    b_nst[k]   = ns;
  }
  __syncthreads();
  // rank of each valid point (position-noise draws happen only for valid points)
  int n_valid;
  {
    const int per = (n + 1023) / 1024;
    int       cnt = 0;
    for (int k = tid * per; k < (tid + 1) * per && k < n; ++k) cnt += b_valid[k];
    int off = block_scan_excl(cnt, s_tmp, &n_valid);
    for (int k = tid * per; k < (tid + 1) * per && k < n; ++k) {
      const int v = b_valid[k];
      b_valid[k]  = v ? off : -1;  // becomes the rank
      off += v;
    }
  }
  __syncthreads();
  // per particle: position, class; counts of velocity / rand draws
  const int total = n * nb;
  const int per   = (total + 1023) / 1024;
  int8_t   *b_cls = d.b_cls + (size_t)a * d.max_pts * nb;
  int      *b_dest = d.b_dest + (size_t)a * d.max_pts * nb;
  float    *b_pay  = d.b_pay + (size_t)a * d.max_pts * nb * 5;
  int       cv = 0, cr = 0;
  for (int j = tid * per; j < (tid + 1) * per && j < total; ++j) {
    const int k = j / nb, p = j % nb;
    int8_t    cls = -1;
    const int rk  = b_valid[k];
    if (rk >= 0) {
      long long g0 = (long long)s.pseq + ((long long)rk * nb + p) * 3;
      const float qx = b_c[k * 3] + d.pg[(g0 + 0) % d.n_gauss];
      const float qy = b_c[k * 3 + 1] + d.pg[(g0 + 1) % d.n_gauss];
      const float qz = b_c[k * 3 + 2] + d.pg[(g0 + 2) % d.n_gauss];
      const int   qi = voxel_index(d, qx, qy, qz);
      if (qi >= 0) {
        const float nx = born[k * 7 + 3], inten = born[k * 7 + 6];
        if (p < b_nst[k])
          cls = 0;
        else if (nx > -100.f && p < model_gen)
          cls = inten > 0.01f ? 1 : 0;
        else
          cls = inten > 0.01f ? 2 : 0;
        b_dest[j]        = qi;
        b_pay[j * 5 + 0] = qx;
        b_pay[j * 5 + 1] = qy;
        b_pay[j * 5 + 2] = qz;
      }
    }
    b_cls[j] = cls;
    cv += cls == 1;
    cr += cls == 2;
  }
  int tv, tr;
  int ov = block_scan_excl(cv, s_tmp, &tv);
  int orr = block_scan_excl(cr, s_tmp, &tr);
  for (int j = tid * per; j < (tid + 1) * per && j < total; ++j) {
    const int8_t cls = b_cls[j];
    if (cls < 0) continue;
    const int k = j / nb;
    float     vx = 0.f, vy = 0.f;
    if (cls == 1) {  // estimated velocity + 4 sigma noise (:944-948); the vz draw is consumed
      const long long g = (long long)s.vseq + (long long)ov * 3;
      vx = born[k * 7 + 3] + 4 * d.vg[(g + 0) % d.n_gauss];
      vy = born[k * 7 + 4] + 4 * d.vg[(g + 1) % d.n_gauss];
      ++ov;
    } else if (cls == 2) {  // generateRandomFloat(-1.5, 1.5) x2, (-0.5, 0.5) consumed (:956-958,1682)
      const long long g = (long long)s.rseq + (long long)orr * 3;
      const float den = (float)((float)2147483647 / (1.5f - -1.5f));
      vx = -1.5f + (float)d.rnd[(g + 0) % d.n_rand] / den;
      vy = -1.5f + (float)d.rnd[(g + 1) % d.n_rand] / den;
      ++orr;
    }
    b_pay[j * 5 + 3] = vx;
    b_pay[j * 5 + 4] = vy;
    d.b_vnext[(size_t)a * d.max_pts * nb + j] = atomicExch(&d.vhead[(size_t)a * d.V + b_dest[j]], j);
  }
  __syncthreads();
  if (tid == 0) {
    s.pseq     = (int)(((long long)s.pseq + (long long)n_valid * nb * 3) % d.n_gauss);
    s.vseq     = (int)(((long long)s.vseq + (long long)tv * 3) % d.n_gauss);
    s.rseq     = (int)(((long long)s.rseq + (long long)tr * 3) % d.n_rand);
    s.born_cnt = total;
  }
}

// addAParticle in (point, particle) order (:1271-1290): one thread per destination voxel
__global__ __launch_bounds__(64) void k_dsp_place_born(DspDev d) {
  const int a = blockIdx.y;
  DspAgent &s = d.ag[a];
  if (!s.ok) return;
  const int B = blockIdx.x * 64 + threadIdx.x;
  if (B >= d.V) return;
  const int head = d.vhead[(size_t)a * d.V + B];
  if (head < 0) return;
  __shared__ int s_key[64][DSP_SLOTS];
  int *K = s_key[threadIdx.x];
  int  n = 0;
  const int    S = d.S;
  const size_t bb = (size_t)a * d.max_pts * d.nb;
  for (int m = head; m >= 0; m = d.b_vnext[bb + m]) {
    if (n == S && m > K[S - 1]) continue;
    int j = n < S ? n : S - 1;
    while (j > 0 && K[j - 1] > m) {
      K[j] = K[j - 1];
      --j;
    }
    K[j] = m;
    if (n < S) ++n;
  }
  uint8_t *fl  = d.flag + ((size_t)a * d.V + B) * DSP_SLOTS;
  unsigned occ = 0;
  for (int p = 0; p < S; ++p)
    if (fl[p] != F_EMPTY) occ |= 1u << p;
  const unsigned all = (1u << S) - 1u;
  const float    w   = s.w_new;
  for (int i = 0; i < n; ++i) {
    const unsigned fr = ~occ & all;
    if (!fr) break;
    const int sl = __ffs(fr) - 1;
    occ |= 1u << sl;
    const size_t at  = ((size_t)a * d.V + B) * DSP_SLOTS + sl;
    const float *pay = d.b_pay + (bb + K[i]) * 5;
    d.flag[at]       = F_NEW;
    d.f[0][at]       = pay[3];
    d.f[1][at]       = pay[4];
    d.f[2][at]       = pay[0];
    d.f[3][at]       = pay[1];
    d.f[4][at]       = pay[2];
    d.f[5][at]       = w;
  }
}

// ---- update: occupancy + resampling, one thread per voxel (:993-1130) -----------------------------------
__global__ __launch_bounds__(256) void k_dsp_occupancy(DspDev d) {
  const int a = blockIdx.y;
  DspAgent &s = d.ag[a];
  if (!s.ok) return;
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= d.V) return;
  const size_t at0 = ((size_t)a * d.V + v) * DSP_SLOTS;
  float       *occ = d.occ + (size_t)a * 4 * d.V;
  const uint4  f4  = *reinterpret_cast<const uint4 *>(d.flag + at0);
  if ((f4.x | f4.y | f4.z | f4.w) == 0u) {
    occ[v] = occ[d.V + v] = occ[2 * d.V + v] = occ[3 * d.V + v] = 0.f;
    return;
  }
  const int S = d.S;
  float     wsum = 0.f, vxs = 0.f, vys = 0.f, vzs = 0.f;
  int       n = 0, n_old = 0;
  unsigned  valid = 0;
  float    *fut = d.fut + (size_t)a * d.T * d.V;
  for (int p = 0; p < S; ++p) {
    const uint8_t c = d.flag[at0 + p];
    if (c == F_EMPTY) continue;
    const float w = d.f[5][at0 + p];
    if ((double)w < 1e-3) {
      d.flag[at0 + p] = F_EMPTY;
      continue;
    }
    if (c != F_NEW) {
      ++n_old;
      const float vx = d.f[0][at0 + p], vy = d.f[1][at0 + p];
      const float px = d.f[2][at0 + p], py = d.f[3][at0 + p], pz = d.f[4][at0 + p];
      vxs += vx;
      vys += vy;
      vzs += 0.f;
      for (int t = 0; t < d.T; ++t) {
        const float pt = d.pred_t[t];
        const int   pi = voxel_index(d, px + vx * pt, py + vy * pt, pz + 0.f * pt);
        if (pi >= 0) atomicAdd(&fut[(size_t)t * d.V + pi], w);
      }
    }
    d.flag[at0 + p] = F_VALID;
    valid |= 1u << p;
    ++n;
    wsum += w;
  }
  occ[v] = wsum;
  if (n_old > 0) {
    occ[d.V + v]     = vxs / (float)n_old;
    occ[2 * d.V + v] = vys / (float)n_old;
    occ[3 * d.V + v] = vzs / (float)n_old;
  } else {
    occ[d.V + v] = occ[2 * d.V + v] = occ[3 * d.V + v] = 0.f;
  }
  if (n < 5) return;
  const int   n_after = n > d.maxp ? d.maxp : n;
  const float w_after = wsum / (float)n_after;
  float       acc_ori = 0.f, acc_new = w_after * 0.5f;
  unsigned    used = valid;  // non-empty slots (copies included)
  for (int p = 0; p < S; ++p) {
    if (!((valid >> p) & 1u)) continue;
    float w = d.f[5][at0 + p];
    acc_ori += w;
    if (acc_ori > acc_new) {
      w = w_after;
      acc_new += w_after;
      int full = 0, p_i = 0;
      while (acc_ori > acc_new) {
        int found = 0;
        if (!full) {
          for (; p_i < S; ++p_i) {
            if (!((used >> p_i) & 1u)) {
              d.flag[at0 + p_i] = F_RESAMP;
              for (int k = 0; k < 5; ++k) d.f[k][at0 + p_i] = d.f[k][at0 + p];
              d.f[5][at0 + p_i] = w;
              used |= 1u << p_i;
              found = 1;
              break;
            }
          }
        }
        if (!found) {
          w += w_after;
          full = 1;
        }
        acc_new += w_after;
      }
      d.f[5][at0 + p] = w;
    } else {
      d.flag[at0 + p] = F_EMPTY;
      used &= ~(1u << p);
    }
  }
}

// ---- publish (:445-469 + risk_voxel.cpp:141-153) ---------------------------------------------------------
__global__ __launch_bounds__(256) void k_dsp_publish(DspDev d, void *__restrict__ grid, GridGeom gg, float thr,
                                                     float *__restrict__ poses, double *__restrict__ stamps) {
  const int half = gg.half;
  const int a = blockIdx.y;
  const int v = blockIdx.x * 256 + threadIdx.x;
  DspAgent &s = d.ag[a];
  if (v == 0) {
    poses[a * 3]     = s.cur[0];
    poses[a * 3 + 1] = s.cur[1];
    poses[a * 3 + 2] = s.cur[2];
    stamps[a]        = s.last_t;
  }
  if (v >= d.V) return;
  float *fut = d.fut + (size_t)a * d.T * d.V;
  char  *g   = reinterpret_cast<char *>(grid) + (size_t)a * d.T * d.V * (half ? 2 : 4);
  const int pv = gg.phys_of(v);  // (rows: v itself, the straight copy; tiles: the cell's place in the slice)
  for (int t = 0; t < d.T; ++t) {
    cell_st(g, (size_t)t * d.V + pv, fut[(size_t)t * d.V + v], half);
    fut[(size_t)t * d.V + v] = 0.f;
  }
  if (d.occ[(size_t)a * 4 * d.V + v] > thr) atomicAdd(&s.n_occupied, 1);
}
__global__ void k_dsp_publish_ego(DspDev d, void *__restrict__ grid, GridGeom gg, int inf_step,
                                  int32_t *__restrict__ out_n) {
  const int half = gg.half;
  const int a = blockIdx.x;
  const int w = 2 * inf_step + 1;
  char     *g = reinterpret_cast<char *>(grid) + (size_t)a * d.T * d.V * (half ? 2 : 4);
  for (int k = threadIdx.x; k < w * w * w; k += blockDim.x) {
    const int x = k / (w * w) - inf_step, y = (k / w) % w - inf_step, z = k % w - inf_step;
    const int idx = z * d.L * d.W + y * d.L + x;  // getVoxelIndex(Vector3i) of the OFFSET (map.h:176)
    if (idx < 0 || idx >= d.V) continue;          // negative index = UB in the reference: skipped
    for (int t = 0; t < 3 && t < d.T; ++t) cell_st(g, (size_t)t * d.V + gg.phys_of(idx), 0.f, half);
  }
  if (threadIdx.x == 0) {
    if (out_n) out_n[a] = d.ag[a].n_occupied;
    d.ag[a].n_occupied = 0;
  }
}

}  // namespace sogm

using namespace sogm;

struct sogm_dsp {
  sogm_ctx          *map;
  DspDev             d;
  SogmDspParams      P;
  std::vector<void *> allocs;
};

namespace {
template <typename T>
int dmalloc(sogm_dsp *h, T **out, size_t n) {
  void *p = nullptr;
  if (hipMalloc(&p, n * sizeof(T) ? n * sizeof(T) : 16) != hipSuccess) return -1;
  h->allocs.push_back(p);
  *out = (T *)p;
  return 0;
}
template <typename T>
int dupload(sogm_dsp *h, const T **out, const T *src, size_t n) {
  T *p;
  if (dmalloc(h, &p, n)) return -1;
  if (hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return -1;
  *out = p;
  return 0;
}
}  // namespace

extern "C" {

void sogm_dsp_destroy(sogm_dsp *h) {
  if (!h) return;
  (void)hipSetDevice(h->map->device);
  for (void *p : h->allocs) (void)hipFree(p);
  delete h;
}

int sogm_dsp_create(sogm_ctx *map, const SogmDspParams *P, const float *p_gauss, const float *v_gauss,
                    int n_gauss, const int32_t *rand_tab, int n_rand, int max_points, sogm_dsp **out) {
  if (!map || !P || !out || !p_gauss || !v_gauss || !rand_tab || n_gauss <= 0 || n_rand <= 0 ||
      max_points <= 0 || max_points > 8192)
    return SOGM_ERR_INVALID_ARG;
  const SogmSpec &sp = map->spec;
  const int       S  = P->max_particle_num_voxel * 2;
  const int       ar = P->angle_resolution;
  if (S > DSP_SLOTS || S < 1 || ar < 1 || sp.T > SOGM_DSP_MAX_T || P->newborn_num < 1 ||
      P->obs_max_per_pyramid < 2 || P->half_fov_h * 2 / ar > 180 || P->half_fov_v * 2 / ar > 180)
    return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(map->device));
  {
    // k_dsp_velocity keeps the cloud in LDS: 20 B per point of dynamic LDS on top of its static arrays; the request
    // must fit the workgroup limit of THIS device (160 KiB on gfx950: max_points <= ~7.4 k), checked here instead of
    // failing every later sogm_update_dsp(labels = NULL) launch
    hipFuncAttributes fa;
    int               lds_max = 0;
    SOGM_HIP_CHECK(hipFuncGetAttributes(&fa, (const void *)k_dsp_velocity));
    SOGM_HIP_CHECK(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, map->device));
    if ((size_t)max_points * 20 + fa.sharedSizeBytes > (size_t)lds_max) {
      set_error("sogm_dsp_create: max_points needs more LDS than a workgroup can have (k_dsp_velocity)", hipErrorInvalidValue);
      return SOGM_ERR_CAPACITY;
    }
  }
  sogm_dsp *h = new (std::nothrow) sogm_dsp;
  if (!h) return SOGM_ERR_HIP;
  h->map    = map;
  h->P      = *P;
  DspDev &d = h->d;
  std::memset(&d, 0, sizeof(d));
  d.A   = map->n_agents;
  d.L   = sp.L;
  d.W   = sp.W;
  d.H   = sp.H;
  d.T   = sp.T;
  d.V   = sp.L * sp.W * sp.H;
  d.S   = S;
  d.nph = P->half_fov_h * 2 / ar;
  d.npv = P->half_fov_v * 2 / ar;
  d.NP  = d.nph * d.npv;
  d.SP  = (int)(d.V * P->max_particle_num_voxel + 1e5) / (360 * 180 / ar / ar) * 2;  // SAFE_PARTICLE_NUM_PYRAMID
  if (d.SP < 1) {
    delete h;
    return SOGM_ERR_INVALID_ARG;
  }
  d.OM       = P->obs_max_per_pyramid;
  d.cand_cap = 2 * d.V;
  d.max_pts  = max_points;
  d.nb       = P->newborn_num;
  d.maxp     = P->max_particle_num_voxel;
  d.n_gauss  = n_gauss;
  d.n_rand   = n_rand;
  d.res      = sp.resolution;
  d.hx       = (d.res * (float)sp.L) * 0.5f;  // map_length_x_half (:572)
  d.hy       = (d.res * (float)sp.W) * 0.5f;
  d.hz       = (d.res * (float)sp.H) * 0.5f;
  d.sigma    = P->sigma_observation;
  d.Pd       = P->p_detection;
  d.kappa    = P->kappa;
  d.w_nb     = P->newborn_weight;
  d.thick    = P->obstacle_thickness;
  for (int t = 0; t < SOGM_DSP_MAX_T; ++t) d.pred_t[t] = P->prediction_times[t];

  // host-side constant tables (setInitParameters :566-632): boundary-plane normals, neighbour table,
  // Gaussian PDF lookup (calculateNormalPDFBuffer :1373-1378) — set-up, evaluated once with libm.
  const float        arr = (float)ar / 180.f * 3.14159265358979323846f;
  std::vector<float> bh((d.nph + 1) * 3), bv((d.npv + 1) * 3), pdf(20000);
  const int          he = P->half_fov_h / ar, ve = P->half_fov_v / ar;
  for (int i = -he; i <= he; i++) {
    bh[(i + he) * 3 + 0] = -sinf((float)i * arr);
    bh[(i + he) * 3 + 1] = cosf((float)i * arr);
    bh[(i + he) * 3 + 2] = 0.f;
  }
  for (int i = -ve; i <= ve; i++) {
    bv[(i + ve) * 3 + 0] = sinf((float)i * arr);
    bv[(i + ve) * 3 + 1] = 0.f;
    bv[(i + ve) * 3 + 2] = cosf((float)i * arr);
  }
  std::vector<int> nbr((size_t)d.NP * 10, 0);
  for (int i = 0; i < d.NP; i++) {
    int h0 = i / d.npv, v0 = i % d.npv, n = 0;
    for (int x = -1; x <= 1; ++x)
      for (int y = -1; y <= 1; ++y) {
        int hh = h0 + x, vv = v0 + y;
        if (hh >= 0 && hh < d.nph && vv >= 0 && vv < d.npv) nbr[i * 10 + 1 + n++] = hh * d.npv + vv;
      }
    nbr[i * 10] = n;
  }
  for (int i = 0; i < 20000; ++i) {
    float value = (float)(i - 10000) * 0.001f;
    pdf[i]      = (1.f / (sqrtf(2.f * 1.57079632679489661923f))) * expf(-powf(value, 2) / (2));
  }
  const size_t A = d.A, V = d.V, VS = V * DSP_SLOTS, NP = d.NP, MP = d.max_pts, NB = d.nb, CC = d.cand_cap;
  int          bad = 0;
  bad |= dmalloc(h, &d.ag, A);
  bad |= dmalloc(h, &d.flag, A * VS);
  for (int k = 0; k < 6; ++k) bad |= dmalloc(h, &d.f[k], A * VS);
  bad |= dmalloc(h, &d.fut, A * d.T * V);
  bad |= dmalloc(h, &d.occ, A * 4 * V);
  bad |= dmalloc(h, &d.pc, A * NP * d.OM * 5);
  bad |= dmalloc(h, &d.nobs, A * NP);
  bad |= dmalloc(h, &d.maxlen, A * NP);
  bad |= dmalloc(h, &d.obs_list, A * MP);
  bad |= dmalloc(h, &d.bp_h, A * (d.nph + 1) * 3);
  bad |= dmalloc(h, &d.bp_v, A * (d.npv + 1) * 3);
  bad |= dmalloc(h, &d.born, A * MP * 7);
  bad |= dmalloc(h, &d.vel_adj, A * MP * VEL_ADJ_CAP);
  bad |= dmalloc(h, &d.vel_deg, A * MP);
  bad |= dmalloc(h, &d.c_key, A * CC);
  bad |= dmalloc(h, &d.c_dest, A * CC);
  bad |= dmalloc(h, &d.c_pyr, A * CC);
  bad |= dmalloc(h, &d.c_assign, A * CC);
  bad |= dmalloc(h, &d.c_vnext, A * CC);
  bad |= dmalloc(h, &d.c_pnext, A * CC);
  bad |= dmalloc(h, &d.c_kill, A * CC);
  bad |= dmalloc(h, &d.c_pay, A * CC * 6);
  bad |= dmalloc(h, &d.vhead, A * V);
  bad |= dmalloc(h, &d.phead, A * NP);
  bad |= dmalloc(h, &d.pyr_list, A * NP * d.SP);
  bad |= dmalloc(h, &d.pyr_n, A * NP);
  bad |= dmalloc(h, &d.pyr_key, A * NP * d.SP);
  bad |= dmalloc(h, &d.pyr_id, A * NP * d.SP);
  bad |= dmalloc(h, &d.b_valid, A * MP);
  bad |= dmalloc(h, &d.b_nstatic, A * MP);
  bad |= dmalloc(h, &d.b_vi, A * MP);
  bad |= dmalloc(h, &d.b_c, A * MP * 3);
  bad |= dmalloc(h, &d.b_cls, A * MP * NB);
  bad |= dmalloc(h, &d.b_dest, A * MP * NB);
  bad |= dmalloc(h, &d.b_vnext, A * MP * NB);
  bad |= dmalloc(h, &d.b_pay, A * MP * NB * 5);
  bad |= dupload(h, &d.pg, p_gauss, (size_t)n_gauss);
  bad |= dupload(h, &d.vg, v_gauss, (size_t)n_gauss);
  bad |= dupload(h, &d.rnd, (const int *)rand_tab, (size_t)n_rand);
  bad |= dupload(h, &d.pdf, pdf.data(), pdf.size());
  bad |= dupload(h, &d.bp_ori_h, bh.data(), bh.size());
  bad |= dupload(h, &d.bp_ori_v, bv.data(), bv.size());
  bad |= dupload(h, &d.nbr, nbr.data(), nbr.size());
  if (bad) {
    set_error("sogm_dsp_create: hipMalloc", hipGetLastError());
    sogm_dsp_destroy(h);
    return SOGM_ERR_HIP;
  }
  std::vector<DspAgent> ag(A);
  std::memset(ag.data(), 0, A * sizeof(DspAgent));
  for (auto &x : ag) {
    x.first   = 1;
    x.quat[0] = 1.f;
  }
  hipError_t e = hipMemcpy(d.ag, ag.data(), A * sizeof(DspAgent), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(d.flag, 0, A * VS);
  if (e == hipSuccess) e = hipMemset(d.fut, 0, A * d.T * V * sizeof(float));
  if (e == hipSuccess) e = hipMemset(d.occ, 0, A * 4 * V * sizeof(float));
  if (e == hipSuccess) e = hipMemset(d.nobs, 0, A * NP * sizeof(int));
  if (e == hipSuccess) e = hipMemset(d.pyr_n, 0, A * NP * sizeof(int));
  if (e == hipSuccess) e = hipMemset(d.pc, 0, A * NP * d.OM * 5 * sizeof(float));
  if (e == hipSuccess) e = hipMemset(d.maxlen, 0xFF, A * NP * sizeof(int));
  for (int k = 0; k < 6 && e == hipSuccess; ++k) e = hipMemset(d.f[k], 0, A * VS * sizeof(float));
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);  // null-stream memsets: complete before the first update on another stream
  if (e != hipSuccess) {
    set_error("sogm_dsp_create: init", e);
    sogm_dsp_destroy(h);
    return SOGM_ERR_HIP;
  }
  // the observe / newborn kernels keep per-pyramid counters / the 1/C_k list in dynamic LDS (function attributes
  // are process-wide: keep the largest request of any handle)
  static size_t lds_obs = 0, lds_born = 0;
  if (2 * NP * sizeof(int) > lds_obs) {
    lds_obs = 2 * NP * sizeof(int);
    (void)hipFuncSetAttribute((const void *)k_dsp_observe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_obs);
  }
  if (MP * sizeof(float) > lds_born) {
    lds_born = MP * sizeof(float);
    (void)hipFuncSetAttribute((const void *)k_dsp_newborn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_born);
  }
  *out = h;
  return SOGM_OK;
}

int sogm_update_dsp(sogm_dsp *h, const float *points, const float *labels, const int32_t *cloud_range,
                    const float *sensor_pos, const float *sensor_quat, const double *stamps,
                    int32_t *out_ok, void *stream) {
  if (!h || !points || !cloud_range || !sensor_pos || !sensor_quat || !stamps) return SOGM_ERR_INVALID_ARG;
  DspDev     &d  = h->d;
  hipStream_t st = (hipStream_t)stream;
  SOGM_HIP_CHECK(hipSetDevice(h->map->device));
  const size_t A = d.A, V = d.V;
  const dim3   g_vox64((unsigned)((V + 63) / 64), (unsigned)A), g_vox256((unsigned)((V + 255) / 256), (unsigned)A);
  const dim3   g_pyr((unsigned)((d.NP + 63) / 64), (unsigned)A);
  hipLaunchKernelGGL(k_dsp_begin, dim3((unsigned)A), dim3(128), 0, st, d, sensor_pos, sensor_quat, stamps, out_ok);
  SOGM_HIP_CHECK(hipMemsetAsync(d.vhead, 0xFF, A * V * sizeof(int), st));
  SOGM_HIP_CHECK(hipMemsetAsync(d.phead, 0xFF, A * d.NP * sizeof(int), st));
  hipLaunchKernelGGL(k_dsp_observe, dim3((unsigned)A), dim3(256), 2 * d.NP * sizeof(int), st, d, points, labels,
                     cloud_range);
  if (!labels) {  // velocityEstimationThread (:305): labels and point order of the new-born list are computed here
    // the attribute is per function, not per sogm_dsp: raise it whenever a handle asks for more than any before
    // (size checked against the device limit in sogm_dsp_create)
    static size_t attr_lds = 0;
    const size_t  lds      = (size_t)d.max_pts * 20;
    if (lds > attr_lds) {
      SOGM_HIP_CHECK(hipFuncSetAttribute((const void *)k_dsp_velocity, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_lds = lds;
    }
    hipLaunchKernelGGL(k_dsp_velocity, dim3((unsigned)A), dim3(1024), lds, st, d, cloud_range, 0.15f);
  }
  hipLaunchKernelGGL(k_dsp_predict, g_vox256, dim3(256), 0, st, d);
  for (int r = 0; r < DSP_ROUNDS; ++r) {
    hipLaunchKernelGGL(k_dsp_place, g_vox64, dim3(64), 0, st, d, r);
    hipLaunchKernelGGL(k_dsp_pyramids, g_pyr, dim3(64), 0, st, d, r);
  }
  hipLaunchKernelGGL(k_dsp_commit_slots, g_vox256, dim3(256), 0, st, d);
  hipLaunchKernelGGL(k_dsp_commit_movers, dim3((unsigned)((d.cand_cap + 255) / 256), (unsigned)A), dim3(256), 0, st, d);
  hipLaunchKernelGGL(k_dsp_ck, dim3((unsigned)((d.max_pts + 255) / 256), (unsigned)A), dim3(256), 0, st, d);
  hipLaunchKernelGGL(k_dsp_weight, dim3((unsigned)((d.NP * d.SP + 255) / 256), (unsigned)A), dim3(256), 0, st, d);
  SOGM_HIP_CHECK(hipMemsetAsync(d.vhead, 0xFF, A * V * sizeof(int), st));
  hipLaunchKernelGGL(k_dsp_newborn, dim3((unsigned)A), dim3(1024), d.max_pts * sizeof(float), st, d);
  hipLaunchKernelGGL(k_dsp_place_born, g_vox64, dim3(64), 0, st, d);
  hipLaunchKernelGGL(k_dsp_occupancy, g_vox256, dim3(256), 0, st, d);
  SOGM_HIP_CHECK(hipGetLastError());
  return SOGM_OK;
}

int sogm_dsp_publish(sogm_dsp *h, int32_t *out_n_occupied, void *stream) {
  if (!h) return SOGM_ERR_INVALID_ARG;
  DspDev     &d  = h->d;
  sogm_ctx   *c  = h->map;
  hipStream_t st = (hipStream_t)stream;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  if (int rc = sogm::join_update(c, st)) return rc;
  if (c->precleared) {  // a pending side-stream clear must not race the copy
    int rc = sogm::adopt_preclear(c, st);
    if (rc) return rc;
  }
  c->tracked[sogm::cur_slot(c)] = 0;  // every cell is written: the next reset of this grid is the dense clear
  hipLaunchKernelGGL(k_dsp_publish, dim3((unsigned)((d.V + 255) / 256), (unsigned)d.A), dim3(256), 0, st, d,
                     (void *)c->d_grid, c->geom, c->geom.risk_threshold, c->d_poses, c->d_stamps);
  hipLaunchKernelGGL(k_dsp_publish_ego, dim3((unsigned)d.A), dim3(128), 0, st, d, (void *)c->d_grid, c->geom, c->geom.inf_step,
                     out_n_occupied);
  SOGM_HIP_CHECK(hipGetLastError());
  c->updated = 1;
  return SOGM_OK;
}

int sogm_dsp_download_state(sogm_dsp *h, int agent, float *store, float *objnum, int32_t *counters) {
  if (!h || agent < 0 || agent >= h->d.A) return SOGM_ERR_INVALID_ARG;
  DspDev &d = h->d;
  SOGM_HIP_CHECK(hipSetDevice(h->map->device));
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  const size_t V = d.V, VS = V * DSP_SLOTS;
  if (store) {
    std::vector<uint8_t> fl(VS);
    std::vector<float>   f[6];
    SOGM_HIP_CHECK(hipMemcpy(fl.data(), d.flag + (size_t)agent * VS, VS, hipMemcpyDeviceToHost));
    for (int k = 0; k < 6; ++k) {
      f[k].resize(VS);
      SOGM_HIP_CHECK(hipMemcpy(f[k].data(), d.f[k] + (size_t)agent * VS, VS * sizeof(float), hipMemcpyDeviceToHost));
    }
    static const float fv[7] = {0.f, 1.f, 0.6f, 7.f, 15.f, 0.f, 0.f};
    for (size_t v = 0; v < V; ++v)
      for (int p = 0; p < d.S; ++p) {
        float       *o  = store + (v * d.S + p) * 9;
        const size_t at = v * DSP_SLOTS + p;
        o[0]            = fv[fl[at] < 7 ? fl[at] : 0];
        o[1]            = f[0][at];
        o[2]            = f[1][at];
        o[3]            = 0.f;
        o[4]            = f[2][at];
        o[5]            = f[3][at];
        o[6]            = f[4][at];
        o[7]            = f[5][at];
        o[8]            = 0.f;
      }
  }
  if (objnum) {
    std::vector<float> occ(4 * V), fut((size_t)d.T * V);
    SOGM_HIP_CHECK(hipMemcpy(occ.data(), d.occ + (size_t)agent * 4 * V, occ.size() * sizeof(float), hipMemcpyDeviceToHost));
    SOGM_HIP_CHECK(hipMemcpy(fut.data(), d.fut + (size_t)agent * d.T * V, fut.size() * sizeof(float), hipMemcpyDeviceToHost));
    const int OD = 4 + d.T;
    for (size_t v = 0; v < V; ++v) {
      for (int k = 0; k < 4; ++k) objnum[v * OD + k] = occ[k * V + v];
      for (int t = 0; t < d.T; ++t) objnum[v * OD + 4 + t] = fut[(size_t)t * V + v];
    }
  }
  if (counters) {
    DspAgent s;
    SOGM_HIP_CHECK(hipMemcpy(&s, d.ag + agent, sizeof(s), hipMemcpyDeviceToHost));
    std::memset(counters, 0, 16 * sizeof(int32_t));
    counters[0]  = s.dbg_voxel_full;
    counters[1]  = s.dbg_pyr_full;
    counters[2]  = s.dbg_out;
    counters[3]  = s.cand_cnt;
    counters[4]  = s.pseq;
    counters[5]  = s.vseq;
    counters[6]  = s.rseq;
    counters[7]  = d.S;
    counters[8]  = d.SP;
    counters[9]  = d.NP;
    counters[10] = s.err_unconverged;
    counters[11] = s.err_pool;
    counters[12] = s.err_points;
    counters[13] = s.valid_points;
    counters[14] = s.n_obs;
    counters[15] = s.ok;
  }
  return SOGM_OK;
}

int sogm_dsp_download_observations(sogm_dsp *h, int agent, int32_t *nobs, float *pc, float *maxlen) {
  if (!h || agent < 0 || agent >= h->d.A) return SOGM_ERR_INVALID_ARG;
  DspDev &d = h->d;
  SOGM_HIP_CHECK(hipSetDevice(h->map->device));
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  const size_t NP = d.NP;
  if (nobs) SOGM_HIP_CHECK(hipMemcpy(nobs, d.nobs + agent * NP, NP * sizeof(int), hipMemcpyDeviceToHost));
  if (pc)
    SOGM_HIP_CHECK(hipMemcpy(pc, d.pc + (size_t)agent * NP * d.OM * 5, NP * d.OM * 5 * sizeof(float), hipMemcpyDeviceToHost));
  if (maxlen) {
    std::vector<int> b(NP);
    SOGM_HIP_CHECK(hipMemcpy(b.data(), d.maxlen + agent * NP, NP * sizeof(int), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < NP; ++i) {
      float f;
      std::memcpy(&f, &b[i], 4);
      maxlen[i] = b[i] < 0 ? -1.f : f;
    }
  }
  return SOGM_OK;
}

// Parity I/O (synchronous): input_cloud_with_velocity of the last update — rows {x, y, z, vx, vy, vz, intensity} in
// the order new-born particles are created — and counters {clusters, possibly dynamic, matched, error code}.
int sogm_dsp_download_born(sogm_dsp *h, int agent, float *born_host, int cap, int32_t *n_born, int32_t *counters4) {
  if (!h || agent < 0 || agent >= h->d.A || cap < 0) return SOGM_ERR_INVALID_ARG;
  DspDev &d = h->d;
  SOGM_HIP_CHECK(hipSetDevice(h->map->device));
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  DspAgent ag;
  SOGM_HIP_CHECK(hipMemcpy(&ag, d.ag + agent, sizeof(ag), hipMemcpyDeviceToHost));
  if (n_born) *n_born = ag.n_born;
  if (counters4) {
    counters4[0] = ag.vel_clusters;
    counters4[1] = ag.vel_dynamic;
    counters4[2] = ag.vel_matched;
    counters4[3] = ag.vel_err;
  }
  if (born_host) {
    const size_t n = (size_t)(ag.n_born < cap ? ag.n_born : cap);
    SOGM_HIP_CHECK(hipMemcpy(born_host, d.born + (size_t)agent * d.max_pts * 7, n * 7 * sizeof(float), hipMemcpyDeviceToHost));
  }
  return SOGM_OK;
}

}  // extern "C"
