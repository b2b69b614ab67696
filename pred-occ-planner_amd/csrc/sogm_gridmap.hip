// sogm_gridmap.hip — batched depth-image front end (GridMap::projectDepthImage / raycastProcess /
// clearAndInflateLocalMap, plan_env/src/grid_map.cpp:210-583; RayCaster, plan_env/src/raycast.cpp) for
// gfx950, behind sogm_gridmap_* in include/sogm_abi.h.  SURVEY section 8 row f1.
//
// One lane per sampled depth pixel.  The reference walks the rays one after the other and lets a ray
// stop at the first voxel that an EARLIER ray of the same frame already traversed (flag_traverse_), and
// skips rays whose end voxel was already a ray end (flag_rayend_): hit/miss counts therefore depend on
// the pixel order.  Reproduced exactly without serialising the rays:
//   * ray index = sample ordinal (row-major), so "earlier" is a comparison of lane ids;
//   * ray-end de-duplication: per-voxel atomic max of an epoch-tagged key picks the smallest ray index;
//   * traversal: owner(u) = smallest ray index that ARRIVES at voxel u.  A ray arrives at the voxels of
//     its path up to and including the first one owned by an earlier ray.  Starting from "nobody owns
//     anything", arrival sets and owners are recomputed until the per-ray stop positions repeat — ray 0
//     is exact at once, and exactness propagates in index order, so the fixed point is the sequential
//     result (rounds after convergence exit immediately; non-convergence raises an error counter);
//   * counts are then accumulated with atomics (order-free) and fused per touched voxel.
// Epoch-tagged 64-bit keys ((frame, round) << 32 | ~ray) make "clear the per-frame flags" free.
// All traffic is scattered 4/8-byte atomics over an L2-resident neighbourhood of the camera: latency /
// atomic-throughput bound, no dense contraction (no MFMA).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/sogm_detmath.h"
#include "sogm_device.hpp"

namespace sogm {

#ifndef GM_ROUNDS
#define GM_ROUNDS 64
#endif

struct GmAgent {
  unsigned long long bb_min[3], bb_max[3];  // order-preserving keys of fp64 min / max of the ray ends
  double cam[3], R[9];
  int    lb_min[3], lb_max[3], upd_min[3], upd_max[3];
  int    in_map, do_rays, has_first_depth, raycast_num, dedup;
  int    n_touched, n_active, n_valid;
  int    converged, final_round, changed[GM_ROUNDS];
  int    err_unconverged, err_touched, err_paths;
  int    local_updated;
};

struct GmDev {
  int    A, nv[3], N, rows, cols, n_u, n_v, n_samples, u0, v0;
  int    use_filter, margin, skip, local_margin, inf_step, ceil_id, has_ceil;
  double origin[3], bmin[3], bmax[3], res, res_inv;
  double fx, fy, cx, cy, inv_factor, scale, maxdist, mindist, max_ray, ground, range[3];
  double hit_log, miss_log, cmin_log, cmax_log, occ_log, unk;
  GmAgent *ag;
  double  *occ;       // [A][N]
  int8_t  *inflate;   // [A][N]
  int     *cnt_hm, *cnt_hit;  // [A][N]
  unsigned long long *rayend, *own[2];  // [A][N] epoch-tagged keys
  int     *touched;   // [A][N]
  unsigned long long *first_trav;  // [A][N] epoch-tagged: the earliest TRAVERSAL touch of the address this frame — (ray << 12 |
                                   // step << 1 | its id had a component outside the map) — k_gm_count / k_gm_fuse
  double  *pt;        // [A][n_samples][3] ray end (world)
  int     *end_vox;   // [A][n_samples]  (-1: sample dropped)
  int     *stop[2];   // [A][n_samples] steps walked in the previous / current round
  uint8_t *active;    // [A][n_samples]
  // de-duplicated (active) rays: compact list + their voxel paths, written once per frame
  int      act_cap, path_max;
  int     *act_ray;   // [A][act_cap] sample index of each active ray
  int     *path;      // [A][path_max][act_cap] voxel address of step k (-1 outside the arrays), slot-minor
  int     *plen;      // [A][act_cap] steps of the full path
  int     *astop[2];  // [A][act_cap] arrivals of the previous / current round
};

__device__ inline unsigned long long d2key(double d) {
  unsigned long long u = (unsigned long long)__double_as_longlong(d);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ inline double key2d(unsigned long long k) {
  return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}
__device__ inline int gm_addr(const GmDev &d, int x, int y, int z) { return x * d.nv[1] * d.nv[2] + y * d.nv[2] + z; }
__device__ inline void gm_pos_to_index(const GmDev &d, const double p[3], int id[3]) {
  for (int i = 0; i < 3; ++i) id[i] = (int)floor((p[i] - d.origin[i]) * d.res_inv);
}
__device__ inline int gm_bound(int v, int n) { return max(min(v, n - 1), 0); }
__device__ inline bool gm_in_map(const GmDev &d, const double p[3]) {
  for (int i = 0; i < 3; ++i)
    if (p[i] < d.bmin[i] + 1e-4) return false;
  for (int i = 0; i < 3; ++i)
    if (p[i] > d.bmax[i] - 1e-4) return false;
  return true;
}
__device__ inline unsigned long long gm_key(int frame, int round, int ray) {
  return ((unsigned long long)(unsigned)(frame * (GM_ROUNDS + 2) + round) << 32) | (unsigned long long)(0xffffffffu - (unsigned)ray);
}
__device__ inline int gm_key_ray(unsigned long long k, int frame, int round) {  // -1: nothing this epoch
  if ((unsigned)(k >> 32) != (unsigned)(frame * (GM_ROUNDS + 2) + round)) return -1;
  return (int)(0xffffffffu - (unsigned)(k & 0xffffffffu));
}

// setCacheOccupancy (grid_map.cpp:192-208): count, remember first touch; returns the voxel address
__device__ inline int gm_touch(const GmDev &d, GmAgent &s, int a, const double p[3], int occ) {
  int id[3];
  gm_pos_to_index(d, p, id);
  const int ad = gm_addr(d, id[0], id[1], id[2]);
  if (ad < 0 || ad >= d.N) return -1;  // out of bounds (UB in the reference): not emulated
  const size_t at = (size_t)a * d.N + ad;
  if (atomicAdd(&d.cnt_hm[at], 1) == 0) {
    const int k = atomicAdd(&s.n_touched, 1);
    if (k < d.N)
      d.touched[(size_t)a * d.N + k] = ad;
    else
      atomicAdd(&s.err_touched, 1);
  }
  if (occ == 1) atomicAdd(&d.cnt_hit[at], 1);
  return ad;
}

// RayCaster (raycast.cpp:17-30,242-335)
__device__ inline int    gm_signum(int x) { return x == 0 ? 0 : x < 0 ? -1 : 1; }
__device__ inline double gm_mod(double v, double m) { return fmod(fmod(v, m) + m, m); }
__device__ inline double gm_intbound(double s, double ds) {
  if (ds < 0) {
    s  = -s;
    ds = -ds;
  }
  s = gm_mod(s, 1);
  return (1 - s) / ds;
}
struct GmRay {
  int    x, y, z, ex, ey, ez, sx, sy, sz;
  double tMaxX, tMaxY, tMaxZ, tDX, tDY, tDZ;
  __device__ inline void set(const double s[3], const double e[3]) {
    x  = (int)floor(s[0]);
    y  = (int)floor(s[1]);
    z  = (int)floor(s[2]);
    ex = (int)floor(e[0]);
    ey = (int)floor(e[1]);
    ez = (int)floor(e[2]);
    const double dx = ex - x, dy = ey - y, dz = ez - z;
    sx    = gm_signum((int)dx);
    sy    = gm_signum((int)dy);
    sz    = gm_signum((int)dz);
    tMaxX = gm_intbound(s[0], dx);
    tMaxY = gm_intbound(s[1], dy);
    tMaxZ = gm_intbound(s[2], dz);
    tDX   = ((double)sx) / dx;
    tDY   = ((double)sy) / dy;
    tDZ   = ((double)sz) / dz;
  }
  __device__ inline bool step(int out[3]) {
    out[0] = x;
    out[1] = y;
    out[2] = z;
    if (x == ex && y == ey && z == ez) return false;
    if (tMaxX < tMaxY) {
      if (tMaxX < tMaxZ) {
        x += sx;
        tMaxX += tDX;
      } else {
        z += sz;
        tMaxZ += tDZ;
      }
    } else {
      if (tMaxY < tMaxZ) {
        y += sy;
        tMaxY += tDY;
      } else {
        z += sz;
        tMaxZ += tDZ;
      }
    }
    return true;
  }
};
// voxel address of the centre of RayCaster cell (cx, cy, cz)  (grid_map.cpp:381-385)
__device__ inline int gm_cell_addr(const GmDev &d, const int c[3]) {
  const double tmp[3] = {(c[0] + 0.5) * d.res, (c[1] + 0.5) * d.res, (c[2] + 0.5) * d.res};
  int          id[3];
  gm_pos_to_index(d, tmp, id);
  const int ad = gm_addr(d, id[0], id[1], id[2]);
  return (ad < 0 || ad >= d.N) ? -1 : ad;
}

// ---- frame begin: depthPoseCallback (:636-665) + the bookkeeping of projectDepthImage / raycastProcess ---
__global__ void k_gm_begin(GmDev d, const double *__restrict__ cam_pos, const double *__restrict__ cam_rot,
                           int32_t *__restrict__ out_updated) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= d.A) return;
  GmAgent &s = d.ag[a];
  for (int k = 0; k < 3; ++k) s.cam[k] = cam_pos[a * 3 + k];
  for (int k = 0; k < 9; ++k) s.R[k] = cam_rot[a * 9 + k];
  s.in_map  = gm_in_map(d, s.cam) ? 1 : 0;
  s.do_rays = 0;
  if (s.in_map) {
    if (d.use_filter && !s.has_first_depth) {
      s.has_first_depth = 1;  // first filtered frame projects nothing (:247-249)
    } else {
      s.do_rays = d.n_samples > 0;
    }
  }
  if (s.do_rays) {
    s.raycast_num += 1;
    s.dedup = s.raycast_num <= 127;  // char flags never equal an int frame counter > 127
  }
  s.n_touched = s.n_active = s.n_valid = 0;
  s.converged                          = 0;
  s.final_round                        = -1;
  for (int r = 0; r < GM_ROUNDS; ++r) s.changed[r] = 0;
  for (int k = 0; k < 3; ++k) {
    s.bb_min[k] = d2key(d.bmax[k]);  // min_x starts at map_max_boundary (:324-330)
    s.bb_max[k] = d2key(d.bmin[k]);
  }
  s.local_updated = 0;
  if (out_updated) out_updated[a] = 0;
}

// ---- projectDepthImage (:210-311) + the per-point head of raycastProcess (:337-366) -------------------------
__global__ __launch_bounds__(256) void k_gm_project(GmDev d, const uint16_t *__restrict__ depth) {
  const int a = blockIdx.y;
  GmAgent  &s = d.ag[a];
  if (!s.do_rays) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= d.n_samples) return;
  const size_t   si  = (size_t)a * d.n_samples + i;
  const uint16_t *img = depth + (size_t)a * d.rows * d.cols;
  const int      vi = i / d.n_u, ui = i % d.n_u;
  const int      v = d.v0 + vi * d.skip, u = d.u0 + ui * d.skip;
  double         dep;
  bool           valid = true;
  if (!d.use_filter) {
    dep = img[(size_t)v * d.cols + ui] / d.scale;  // row_ptr++ walks consecutive pixels (:225-231)
  } else {
    dep = img[(size_t)v * d.cols + u] * d.inv_factor;
    const size_t   nxt = (size_t)v * d.cols + u + d.skip;  // the zero test reads the NEXT sample (:262-267)
    const uint16_t nv  = nxt < (size_t)d.rows * d.cols ? img[nxt] : (uint16_t)1;
    if (nv == 0) {
      dep = d.max_ray + 0.1;
    } else if (dep < d.mindist) {
      valid = false;
    } else if (dep > d.maxdist) {
      dep = d.max_ray + 0.1;
    }
  }
  d.end_vox[si] = -1;
  d.active[si]  = 0;
  if (!valid) return;
  const double c[3] = {(u - d.cx) * dep / d.fx, (v - d.cy) * dep / d.fy, dep};
  double       pt[3];
  for (int k = 0; k < 3; ++k) pt[k] = ((s.R[k * 3] * c[0] + s.R[k * 3 + 1] * c[1]) + s.R[k * 3 + 2] * c[2]) + s.cam[k];
  int occ = 1;
  if (!gm_in_map(d, pt)) {  // closetPointInMap (:447-467)
    double diff[3], min_t = 1000000;
    for (int k = 0; k < 3; ++k) diff[k] = pt[k] - s.cam[k];
    for (int k = 0; k < 3; ++k) {
      if (fabs(diff[k]) > 0) {
        const double t1 = (d.bmax[k] - s.cam[k]) / diff[k];
        if (t1 > 0 && t1 < min_t) min_t = t1;
        const double t2 = (d.bmin[k] - s.cam[k]) / diff[k];
        if (t2 > 0 && t2 < min_t) min_t = t2;
      }
    }
    for (int k = 0; k < 3; ++k) pt[k] = s.cam[k] + (min_t - 1e-3) * diff[k];
    const double d0 = pt[0] - s.cam[0], d1 = pt[1] - s.cam[1], d2 = pt[2] - s.cam[2];
    const double len = sogm_det::sqrt_rn((d0 * d0 + d1 * d1) + d2 * d2);
    if (len > d.max_ray)
      for (int k = 0; k < 3; ++k) pt[k] = (pt[k] - s.cam[k]) / len * d.max_ray + s.cam[k];
    occ = 0;
  } else {
    const double d0 = pt[0] - s.cam[0], d1 = pt[1] - s.cam[1], d2 = pt[2] - s.cam[2];
    const double len = sogm_det::sqrt_rn((d0 * d0 + d1 * d1) + d2 * d2);
    if (len > d.max_ray) {
      for (int k = 0; k < 3; ++k) pt[k] = (pt[k] - s.cam[k]) / len * d.max_ray + s.cam[k];
      occ = 0;
    }
  }
  const int e = gm_touch(d, s, a, pt, occ);
  for (int k = 0; k < 3; ++k) {
    d.pt[si * 3 + k] = pt[k];
    atomicMin(&s.bb_min[k], d2key(pt[k]));
    atomicMax(&s.bb_max[k], d2key(pt[k]));
  }
  d.end_vox[si] = e >= 0 ? e : -2;  // -2: valid ray whose end is outside the arrays (never de-duplicated)
  if (e >= 0) atomicMax(&d.rayend[(size_t)a * d.N + e], gm_key(s.raycast_num, 0, i));
  atomicAdd(&s.n_valid, 1);
}

// flag_rayend_ (:370-376): the first ray ending in a voxel traverses, later ones skip
__global__ __launch_bounds__(256) void k_gm_select(GmDev d) {
  const int a = blockIdx.y;
  GmAgent  &s = d.ag[a];
  if (!s.do_rays) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= d.n_samples) return;
  const size_t si = (size_t)a * d.n_samples + i;
  const int    e  = d.end_vox[si];
  if (e == -1) return;
  bool act = true;
  if (s.dedup && e >= 0) act = gm_key_ray(d.rayend[(size_t)a * d.N + e], s.raycast_num, 0) == i;
  d.active[si]  = act ? 1 : 0;
  d.stop[0][si] = -1;
  d.stop[1][si] = -1;
  if (act) {
    const int slot = atomicAdd(&s.n_active, 1);
    if (s.dedup) {  // the fixed-point rounds work on the compact list (order irrelevant: ray ids carry the order)
      if (slot < d.act_cap) {
        d.act_ray[(size_t)a * d.act_cap + slot]  = i;
        d.astop[0][(size_t)a * d.act_cap + slot] = -1;
        d.astop[1][(size_t)a * d.act_cap + slot] = -1;
      } else {
        atomicAdd(&s.err_paths, 1);
      }
    }
  }
}

// voxel path of every active ray, once per frame: the DDA does not depend on the owners
__global__ __launch_bounds__(256) void k_gm_paths(GmDev d) {
  const int a = blockIdx.y;
  GmAgent  &s = d.ag[a];
  if (!s.do_rays || !s.dedup) return;
  const int slot = blockIdx.x * 256 + threadIdx.x;
  const int na   = s.n_active < d.act_cap ? s.n_active : d.act_cap;
  if (slot >= na) return;
  const int    i  = d.act_ray[(size_t)a * d.act_cap + slot];
  const size_t si = (size_t)a * d.n_samples + i;
  const double st[3] = {d.pt[si * 3] / d.res, d.pt[si * 3 + 1] / d.res, d.pt[si * 3 + 2] / d.res};
  const double en[3] = {s.cam[0] / d.res, s.cam[1] / d.res, s.cam[2] / d.res};
  GmRay        rc;
  rc.set(st, en);
  int *path = d.path + (size_t)a * d.path_max * d.act_cap + slot;
  int  c[3], k = 0;
  while (rc.step(c)) {
    if (k < d.path_max) path[(size_t)k * d.act_cap] = gm_cell_addr(d, c);
    ++k;
  }
  if (k > d.path_max) {
    atomicAdd(&s.err_paths, 1);
    k = d.path_max;
  }
  d.plen[(size_t)a * d.act_cap + slot] = k;
}

// One fixed-point round over the compact list.  A ray's path is fixed, so a round is: fetch the previous
// round's owners of 16 path voxels at a time (independent loads, staged through LDS: "ray segments"), find
// the first voxel owned by an earlier ray, then publish this round's arrivals with fire-and-forget atomics.
// Two memory round trips per segment instead of one per voxel.
#define GM_SEG 16
__global__ __launch_bounds__(256) void k_gm_walk(GmDev d, int round) {
  const int a = blockIdx.y;
  GmAgent  &s = d.ag[a];
  if (!s.do_rays || !s.dedup) return;
  // converged = an earlier round r >= 1 changed no stop position (changed[] of rounds that never ran stays 0,
  // so "the previous round changed nothing" identifies every round after the fixed point as well)
  if (round >= 2 && s.changed[round - 1] == 0) return;
  const int slot = blockIdx.x * 256 + threadIdx.x;
  const int na   = s.n_active < d.act_cap ? s.n_active : d.act_cap;
  if (blockIdx.x * 256 >= na) return;
  __shared__ int                s_addr[GM_SEG][256];
  __shared__ unsigned long long s_own[GM_SEG][256];
  const bool on = slot < na;
  const int  i    = on ? d.act_ray[(size_t)a * d.act_cap + slot] : 0;
  const int  len  = on ? d.plen[(size_t)a * d.act_cap + slot] : 0;
  const int *path = d.path + (size_t)a * d.path_max * d.act_cap + slot;
  const unsigned long long *prev = d.own[(round + 1) & 1] + (size_t)a * d.N;
  unsigned long long       *cur  = d.own[round & 1] + (size_t)a * d.N;
  const unsigned long long  key  = gm_key(s.raycast_num, round + 1, i);
  int arrivals = len;  // steps the ray arrives at (the stopping voxel included)
  for (int base = 0; base < len; base += GM_SEG) {
#pragma unroll
    for (int q = 0; q < GM_SEG; ++q) s_addr[q][threadIdx.x] = base + q < len ? path[(size_t)(base + q) * d.act_cap] : -1;
#pragma unroll
    for (int q = 0; q < GM_SEG; ++q) {
      const int ad          = s_addr[q][threadIdx.x];
      s_own[q][threadIdx.x] = (round > 0 && ad >= 0) ? prev[ad] : 0ull;
    }
    int stop = -1;
#pragma unroll
    for (int q = 0; q < GM_SEG; ++q) {
      if (stop < 0 && base + q < len && s_addr[q][threadIdx.x] >= 0) {
        const int o = gm_key_ray(s_own[q][threadIdx.x], s.raycast_num, round);
        if (round > 0 && o >= 0 && o < i) stop = q;
      }
    }
    const int last = stop >= 0 ? stop : GM_SEG - 1;
#pragma unroll
    for (int q = 0; q < GM_SEG; ++q)
      if (q <= last && base + q < len && s_addr[q][threadIdx.x] >= 0) atomicMax(&cur[s_addr[q][threadIdx.x]], key);
    if (stop >= 0) {
      arrivals = base + stop + 1;
      break;
    }
  }
  if (on) {
    if (arrivals != d.astop[(round + 1) & 1][(size_t)a * d.act_cap + slot]) atomicAdd(&s.changed[round], 1);
    d.astop[round & 1][(size_t)a * d.act_cap + slot] = arrivals;
  }
}
// first round r >= 1 that changed nothing = the fixed point (its owners equal the previous round's)
__device__ inline int gm_final_round(const GmAgent &s) {
  for (int r = 1; r < GM_ROUNDS; ++r)
    if (s.changed[r] == 0) return r;
  return GM_ROUNDS - 1;
}

// final walk: count the arrivals (setCacheOccupancy(tmp, 0), :383) with the converged owners
__global__ __launch_bounds__(256) void k_gm_count(GmDev d) {
  const int a = blockIdx.y;
  GmAgent  &s = d.ag[a];
  if (!s.do_rays) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= d.n_samples) return;
  const size_t si = (size_t)a * d.n_samples + i;
  if (!d.active[si]) return;
  const double st[3] = {d.pt[si * 3] / d.res, d.pt[si * 3 + 1] / d.res, d.pt[si * 3 + 2] / d.res};
  const double en[3] = {s.cam[0] / d.res, s.cam[1] / d.res, s.cam[2] / d.res};
  GmRay        rc;
  rc.set(st, en);
  const int                 fr  = s.dedup ? gm_final_round(s) : 0;
  const unsigned long long *own = d.own[fr & 1] + (size_t)a * d.N;
  int c[3], step = 0;
  while (rc.step(c)) {
    const double tmp[3] = {(c[0] + 0.5) * d.res, (c[1] + 0.5) * d.res, (c[2] + 0.5) * d.res};
    const int    ad     = gm_touch(d, s, a, tmp, 0);
    if (ad >= 0) {
      // The reference queues the Vector3i ID of a voxel's first touch of the frame and judges `in_local` on that id (:199-201,
      // :430-433).  An id with a component outside the map — a ray voxel below the ground plane has z = -1 — reaches, through
      // the unchecked flat address, a cell of the neighbouring row: if that touch is the address's FIRST of the frame, the cell
      // counts as "outside the local range".  Sequential order = (ray, step); kept as an epoch-tagged maximum of its complement.
      int id[3];
      gm_pos_to_index(d, tmp, id);
      const unsigned wrapped = (id[0] < 0 || id[0] >= d.nv[0] || id[1] < 0 || id[1] >= d.nv[1] || id[2] < 0 || id[2] >= d.nv[2]) ? 1u : 0u;
      const unsigned v = ((unsigned)i << 12) | ((unsigned)(step < 2047 ? step : 2047) << 1) | wrapped;
      atomicMax(&d.first_trav[(size_t)a * d.N + ad], ((unsigned long long)(unsigned)s.raycast_num << 32) | (unsigned long long)(0xffffffffu - v));
    }
    ++step;
    if (ad < 0 || !s.dedup) continue;
    const int o = gm_key_ray(own[ad], s.raycast_num, fr + 1);
    if (o >= 0 && o < i) break;
  }
}

// local bounds (:397-420)
__global__ void k_gm_bounds(GmDev d, int32_t *__restrict__ out_updated) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= d.A) return;
  GmAgent &s = d.ag[a];
  if (!s.do_rays) return;
  if (s.n_valid == 0) {  // proj_points_cnt == 0: raycastProcess returns before counting the frame (:315-320)
    s.raycast_num -= 1;
    return;
  }
  if (s.dedup) {
    s.final_round = gm_final_round(s);
    if (s.changed[s.final_round] != 0) s.err_unconverged += 1;
  }
  double mn[3], mx[3];
  for (int k = 0; k < 3; ++k) {
    mn[k] = fmin(key2d(s.bb_min[k]), s.cam[k]);
    mx[k] = fmax(key2d(s.bb_max[k]), s.cam[k]);
  }
  mx[2] = fmax(mx[2], d.ground);
  int lo[3], hi[3];
  gm_pos_to_index(d, mx, hi);
  gm_pos_to_index(d, mn, lo);
  double rl[3], rh[3];
  for (int k = 0; k < 3; ++k) {
    s.lb_min[k] = gm_bound(lo[k], d.nv[k]);
    s.lb_max[k] = gm_bound(hi[k], d.nv[k]);
    rl[k]       = s.cam[k] - d.range[k];
    rh[k]       = s.cam[k] + d.range[k];
  }
  gm_pos_to_index(d, rl, lo);
  gm_pos_to_index(d, rh, hi);
  for (int k = 0; k < 3; ++k) {
    s.upd_min[k] = gm_bound(lo[k], d.nv[k]);
    s.upd_max[k] = gm_bound(hi[k], d.nv[k]);
  }
  s.local_updated = 1;
  if (out_updated) out_updated[a] = 1;
}

// hit/miss fusion of every touched voxel (:424-444)
__global__ __launch_bounds__(256) void k_gm_fuse(GmDev d) {
  const int a = blockIdx.y;
  GmAgent  &s = d.ag[a];
  if (!s.local_updated) return;
  const int n = s.n_touched < d.N ? s.n_touched : d.N;
  for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
    const int    ad = d.touched[(size_t)a * d.N + k];
    const size_t at = (size_t)a * d.N + ad;
    const short  hm = (short)d.cnt_hm[at], hit = (short)d.cnt_hit[at];  // 16-bit counters (grid_map.h)
    d.cnt_hm[at]  = 0;
    d.cnt_hit[at] = 0;
    const double upd = (int)hit >= (int)hm - (int)hit ? d.hit_log : d.miss_log;
    double       o   = d.occ[at];
    if (upd >= 0 && o >= d.cmax_log) continue;
    if (upd <= 0 && o <= d.cmin_log) {
      d.occ[at] = d.cmin_log;
      continue;
    }
    const int ix = ad / (d.nv[1] * d.nv[2]), iy = (ad / d.nv[2]) % d.nv[1], iz = ad % d.nv[2];
    // whose id was queued for this address: the first touch of the frame in (ray, step) order.  Ray-END touches are of points
    // inside the map (their ids are the cell's own) and a ray's end touch precedes its traversal; a traversal touch may have
    // come through an id outside the map (k_gm_count): such an id is never inside [upd_min, upd_max]
    bool first_is_wrapped = false;
    {
      const unsigned long long kt = d.first_trav[at];
      if ((unsigned)(kt >> 32) == (unsigned)s.raycast_num) {
        const unsigned v = 0xffffffffu - (unsigned)(kt & 0xffffffffu);
        const int      j_end = gm_key_ray(d.rayend[at], s.raycast_num, 0);
        first_is_wrapped = (v & 1u) != 0 && !(j_end >= 0 && j_end <= (int)(v >> 12));
      }
    }
    const bool in_local = !first_is_wrapped && ix >= s.upd_min[0] && ix <= s.upd_max[0] && iy >= s.upd_min[1] &&
                          iy <= s.upd_max[1] && iz >= s.upd_min[2] && iz <= s.upd_max[2];
    if (!in_local) o = d.cmin_log;
    d.occ[at] = fmin(fmax(o + upd, d.cmin_log), d.cmax_log);
  }
}

// clearAndInflateLocalMap (:469-583), three passes over index boxes
__device__ inline void gm_cut_boxes(const GmDev &d, const GmAgent &s, int mc[3], int xc[3], int mm[3], int xm[3]) {
  for (int k = 0; k < 3; ++k) {
    mc[k] = gm_bound(s.lb_min[k] - d.local_margin, d.nv[k]);
    xc[k] = gm_bound(s.lb_max[k] + d.local_margin, d.nv[k]);
    mm[k] = gm_bound(mc[k] - 5, d.nv[k]);
    xm[k] = gm_bound(xc[k] + 5, d.nv[k]);
  }
}
__global__ __launch_bounds__(256) void k_gm_clear_shell(GmDev d) {
  const int a = blockIdx.y;
  GmAgent  &s = d.ag[a];
  if (!s.local_updated) return;
  int mc[3], xc[3], mm[3], xm[3];
  gm_cut_boxes(d, s, mc, xc, mm, xm);
  const int       sx = xm[0] - mm[0] + 1, sy = xm[1] - mm[1] + 1, sz = xm[2] - mm[2] + 1;
  const long long tot = (long long)sx * sy * sz;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < tot; k += (long long)gridDim.x * 256) {
    const int x = mm[0] + (int)(k / ((long long)sy * sz)), y = mm[1] + (int)((k / sz) % sy), z = mm[2] + (int)(k % sz);
    const bool inside = x >= mc[0] && x <= xc[0] && y >= mc[1] && y <= xc[1] && z >= mc[2] && z <= xc[2];
    if (!inside) d.occ[(size_t)a * d.N + gm_addr(d, x, y, z)] = d.unk;
  }
}
__global__ __launch_bounds__(256) void k_gm_inflate(GmDev d, int pass) {
  const int a = blockIdx.y;
  GmAgent  &s = d.ag[a];
  if (!s.local_updated) return;
  const int       sx = s.lb_max[0] - s.lb_min[0] + 1, sy = s.lb_max[1] - s.lb_min[1] + 1, sz = s.lb_max[2] - s.lb_min[2] + 1;
  const long long tot = (long long)sx * sy * sz;
  int8_t         *inf = d.inflate + (size_t)a * d.N;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < tot; k += (long long)gridDim.x * 256) {
    const int x = s.lb_min[0] + (int)(k / ((long long)sy * sz)), y = s.lb_min[1] + (int)((k / sz) % sy),
              z = s.lb_min[2] + (int)(k % sz);
    const int ad = gm_addr(d, x, y, z);
    if (pass == 0) {
      inf[ad] = 0;
    } else if (pass == 1) {
      if (d.occ[(size_t)a * d.N + ad] > d.occ_log) {
        for (int p = -d.inf_step; p <= d.inf_step; ++p)
          for (int q = -d.inf_step; q <= d.inf_step; ++q)
            for (int r = -d.inf_step; r <= d.inf_step; ++r) {
              const int idx = gm_addr(d, x + p, y + q, z + r);  // wraps across rows like the reference (:549-556)
              if (idx >= 0 && idx < d.N) inf[idx] = 1;
            }
      }
    } else if (d.has_ceil && z == s.lb_min[2]) {  // virtual ceiling (:574-581), once per (x, y)
      const int idx = gm_addr(d, x, y, d.ceil_id);
      if (idx >= 0 && idx < d.N) inf[idx] = 1;
    }
  }
}

// getInflateOccupancy (grid_map.h:342-349)
__global__ void k_gm_query(GmDev d, const int32_t *__restrict__ agent, const double *__restrict__ pos, int n,
                           int8_t *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double p[3] = {pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2]};
  if (!gm_in_map(d, p)) {
    out[i] = -1;
    return;
  }
  int id[3];
  gm_pos_to_index(d, p, id);
  out[i] = d.inflate[(size_t)agent[i] * d.N + gm_addr(d, id[0], id[1], id[2])];
}

__global__ void k_gm_fill_f64(double *p, size_t n, double v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_gm_set_frame(GmDev d, int frame) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a < d.A) d.ag[a].raycast_num = frame;
}

}  // namespace sogm

using namespace sogm;

struct sogm_gridmap {
  GmDev               d;
  int                 device;
  std::vector<void *> allocs;
};

extern "C" {

void sogm_gridmap_destroy(sogm_gridmap *g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  (void)hipDeviceSynchronize();
  for (void *p : g->allocs) (void)hipFree(p);
  delete g;
}

int sogm_gridmap_create(const SogmGridMapParams *P, int n_agents, int device, sogm_gridmap **out) {
  if (!P || !out || n_agents <= 0 || !(P->resolution > 0) || P->skip_pixel < 1 || P->rows < 1 || P->cols < 1 ||
      !(P->fx > 0) || !(P->fy > 0) || !(P->k_depth_scaling_factor > 0))
    return SOGM_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SOGM_ERR_NO_DEVICE;
  SOGM_HIP_CHECK(hipSetDevice(device));
  sogm_gridmap *g = new (std::nothrow) sogm_gridmap;
  if (!g) return SOGM_ERR_HIP;
  g->device = device;
  GmDev &d  = g->d;
  std::memset(&d, 0, sizeof(d));
  SogmGridMapParams p = *P;
  if (p.virtual_ceil_height - p.ground_height > p.map_size[2]) p.virtual_ceil_height = p.ground_height + p.map_size[2];
  d.A       = n_agents;
  d.res     = p.resolution;
  d.res_inv = 1 / p.resolution;
  d.origin[0] = -p.map_size[0] / 2.0;
  d.origin[1] = -p.map_size[1] / 2.0;
  d.origin[2] = p.ground_height;
  for (int i = 0; i < 3; ++i) {
    d.nv[i]   = (int)ceil(p.map_size[i] / p.resolution);
    d.bmin[i] = d.origin[i];
    d.bmax[i] = d.origin[i] + p.map_size[i];
    d.range[i] = p.local_update_range[i];
  }
  d.N = d.nv[0] * d.nv[1] * d.nv[2];
  auto logit = [](double x) { return log(x / (1 - x)); };  // set-up, host libm like the oracle
  d.hit_log  = logit(p.p_hit);
  d.miss_log = logit(p.p_miss);
  d.cmin_log = logit(p.p_min);
  d.cmax_log = logit(p.p_max);
  d.occ_log  = logit(p.p_occ);
  d.unk      = d.cmin_log - 0.01;
  d.rows = p.rows;
  d.cols = p.cols;
  d.use_filter = p.use_depth_filter ? 1 : 0;
  d.margin = p.depth_filter_margin;
  d.skip   = p.skip_pixel;
  d.u0 = d.use_filter ? d.margin : 0;
  d.v0 = d.u0;
  const int u1 = d.use_filter ? d.cols - d.margin : d.cols, v1 = d.use_filter ? d.rows - d.margin : d.rows;
  d.n_u = u1 > d.u0 ? (u1 - d.u0 + d.skip - 1) / d.skip : 0;
  d.n_v = v1 > d.v0 ? (v1 - d.v0 + d.skip - 1) / d.skip : 0;
  d.n_samples = d.n_u * d.n_v;
  d.fx = p.fx;
  d.fy = p.fy;
  d.cx = p.cx;
  d.cy = p.cy;
  d.scale      = p.k_depth_scaling_factor;
  d.inv_factor = 1.0 / p.k_depth_scaling_factor;
  d.maxdist = p.depth_filter_maxdist;
  d.mindist = p.depth_filter_mindist;
  d.max_ray = p.max_ray_length;
  d.ground  = p.ground_height;
  d.local_margin = p.local_map_margin;
  d.inf_step = (int)ceil(p.obstacles_inflation / p.resolution);
  d.has_ceil = p.virtual_ceil_height > -0.5;
  d.ceil_id  = (int)floor((p.virtual_ceil_height - d.origin[2]) * d.res_inv) - 1;
  const size_t A = n_agents, N = d.N, NS = d.n_samples > 0 ? d.n_samples : 1;
  auto alloc = [&](void **ptr, size_t bytes) {
    if (hipMalloc(ptr, bytes ? bytes : 16) != hipSuccess) return -1;
    g->allocs.push_back(*ptr);
    return 0;
  };
  int bad = 0;
  bad |= alloc((void **)&d.ag, A * sizeof(GmAgent));
  bad |= alloc((void **)&d.occ, A * N * sizeof(double));
  bad |= alloc((void **)&d.inflate, A * N);
  bad |= alloc((void **)&d.cnt_hm, A * N * sizeof(int));
  bad |= alloc((void **)&d.cnt_hit, A * N * sizeof(int));
  bad |= alloc((void **)&d.rayend, A * N * 8);
  bad |= alloc((void **)&d.own[0], A * N * 8);
  bad |= alloc((void **)&d.own[1], A * N * 8);
  bad |= alloc((void **)&d.touched, A * N * sizeof(int));
  bad |= alloc((void **)&d.first_trav, A * N * 8);
  bad |= alloc((void **)&d.pt, A * NS * 3 * sizeof(double));
  bad |= alloc((void **)&d.end_vox, A * NS * sizeof(int));
  bad |= alloc((void **)&d.stop[0], A * NS * sizeof(int));
  bad |= alloc((void **)&d.stop[1], A * NS * sizeof(int));
  bad |= alloc((void **)&d.active, A * NS);
  d.act_cap  = (int)(NS / 4 > 4096 ? NS / 4 : 4096);  // rays left after the ray-end de-duplication
  d.path_max = 192;                                   // DDA steps of the longest ray (max_ray_length / resolution * 3)
  {
    const double steps = 3.0 * p.max_ray_length / p.resolution + 8.0;
    if (steps > d.path_max) d.path_max = (int)steps;
  }
  const size_t AC = d.act_cap;
  bad |= alloc((void **)&d.act_ray, A * AC * sizeof(int));
  bad |= alloc((void **)&d.path, A * AC * (size_t)d.path_max * sizeof(int));
  bad |= alloc((void **)&d.plen, A * AC * sizeof(int));
  bad |= alloc((void **)&d.astop[0], A * AC * sizeof(int));
  bad |= alloc((void **)&d.astop[1], A * AC * sizeof(int));
  if (bad) {
    set_error("sogm_gridmap_create: hipMalloc", hipGetLastError());
    sogm_gridmap_destroy(g);
    return SOGM_ERR_HIP;
  }
  std::vector<GmAgent> ag(A);
  std::memset(ag.data(), 0, A * sizeof(GmAgent));
  for (auto &x : ag)
    for (int k = 0; k < 3; ++k) {  // resetBuffer (:166-176)
      x.lb_min[k] = 0;
      x.lb_max[k] = d.nv[k] - 1;
    }
  hipError_t e = hipMemcpy(d.ag, ag.data(), A * sizeof(GmAgent), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(d.inflate, 0, A * N);
  if (e == hipSuccess) e = hipMemset(d.cnt_hm, 0, A * N * sizeof(int));
  if (e == hipSuccess) e = hipMemset(d.cnt_hit, 0, A * N * sizeof(int));
  if (e == hipSuccess) e = hipMemset(d.rayend, 0, A * N * 8);
  if (e == hipSuccess) e = hipMemset(d.own[0], 0, A * N * 8);
  if (e == hipSuccess) e = hipMemset(d.own[1], 0, A * N * 8);
  if (e == hipSuccess) e = hipMemset(d.first_trav, 0, A * N * 8);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_gm_fill_f64, dim3(2048), dim3(256), 0, 0, d.occ, A * N, d.unk);
    e = hipDeviceSynchronize();
  }
  if (e != hipSuccess) {
    set_error("sogm_gridmap_create: init", e);
    sogm_gridmap_destroy(g);
    return SOGM_ERR_HIP;
  }
  *out = g;
  return SOGM_OK;
}

int sogm_gridmap_update(sogm_gridmap *g, const uint16_t *depth, const double *cam_pos, const double *cam_rot,
                        int32_t *out_updated, void *stream) {
  if (!g || !depth || !cam_pos || !cam_rot) return SOGM_ERR_INVALID_ARG;
  GmDev      &d  = g->d;
  hipStream_t st = (hipStream_t)stream;
  SOGM_HIP_CHECK(hipSetDevice(g->device));
  const unsigned A = d.A;
  const dim3     gs((d.n_samples + 255) / 256 > 0 ? (d.n_samples + 255) / 256 : 1, A), ga((A + 63) / 64);
  hipLaunchKernelGGL(k_gm_begin, ga, dim3(64), 0, st, d, cam_pos, cam_rot, out_updated);
  hipLaunchKernelGGL(k_gm_project, gs, dim3(256), 0, st, d, depth);
  hipLaunchKernelGGL(k_gm_select, gs, dim3(256), 0, st, d);
  const dim3 gact((d.act_cap + 255) / 256, A);
  hipLaunchKernelGGL(k_gm_paths, gact, dim3(256), 0, st, d);
  for (int r = 0; r < GM_ROUNDS; ++r) {
    hipLaunchKernelGGL(k_gm_walk, gact, dim3(256), 0, st, d, r);
  }
  hipLaunchKernelGGL(k_gm_count, gs, dim3(256), 0, st, d);
  hipLaunchKernelGGL(k_gm_bounds, ga, dim3(64), 0, st, d, out_updated);
  hipLaunchKernelGGL(k_gm_fuse, dim3(1024, A), dim3(256), 0, st, d);
  hipLaunchKernelGGL(k_gm_clear_shell, dim3(1024, A), dim3(256), 0, st, d);
  for (int pass = 0; pass < 3; ++pass) hipLaunchKernelGGL(k_gm_inflate, dim3(1024, A), dim3(256), 0, st, d, pass);
  SOGM_HIP_CHECK(hipGetLastError());
  return SOGM_OK;
}

int sogm_gridmap_query_inflate(sogm_gridmap *g, const int32_t *agent_idx, const double *pos, int n, int8_t *out,
                               void *stream) {
  if (!g || !agent_idx || !pos || !out || n < 0) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(g->device));
  if (n > 0)
    hipLaunchKernelGGL(k_gm_query, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, g->d, agent_idx, pos, n, out);
  SOGM_HIP_CHECK(hipGetLastError());
  return SOGM_OK;
}

int sogm_gridmap_download(sogm_gridmap *g, int agent, double *occ, int8_t *inflate, int32_t *bounds,
                          int32_t *counters) {
  if (!g || agent < 0 || agent >= g->d.A) return SOGM_ERR_INVALID_ARG;
  GmDev &d = g->d;
  SOGM_HIP_CHECK(hipSetDevice(g->device));
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  const size_t N = d.N;
  if (occ) SOGM_HIP_CHECK(hipMemcpy(occ, d.occ + agent * N, N * sizeof(double), hipMemcpyDeviceToHost));
  if (inflate) SOGM_HIP_CHECK(hipMemcpy(inflate, d.inflate + agent * N, N, hipMemcpyDeviceToHost));
  if (bounds || counters) {
    GmAgent s;
    SOGM_HIP_CHECK(hipMemcpy(&s, d.ag + agent, sizeof(s), hipMemcpyDeviceToHost));
    if (bounds)
      for (int k = 0; k < 3; ++k) {
        bounds[k]     = s.lb_min[k];
        bounds[3 + k] = s.lb_max[k];
      }
    if (counters) {
      counters[0] = s.n_valid;
      counters[1] = s.n_active;
      counters[2] = s.final_round + 1;
      counters[3] = s.err_unconverged + s.err_touched + s.err_paths;
    }
  }
  return SOGM_OK;
}

int sogm_gridmap_force_frame(sogm_gridmap *g, int raycast_num) {
  if (!g) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(g->device));
  hipLaunchKernelGGL(k_gm_set_frame, dim3((g->d.A + 63) / 64), dim3(64), 0, 0, g->d, raycast_num);
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  return SOGM_OK;
}

}  // extern "C"
