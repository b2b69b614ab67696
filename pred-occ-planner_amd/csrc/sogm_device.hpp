// sogm_device.hpp — device-side grid geometry, context layout and small helpers shared by the
// HIP translation units.  gfx950 only; compiled with -ffp-contract=off so that every fp32/fp64
// expression is evaluated exactly as written (voxel indices must be bit-exact).
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <new>
#include <stdint.h>

#include "../../include/sogm_abi_debug.h"  // (includes sogm_abi.h: the library defines both headers' entry points)
#include "../../include/sogm_detmath.h"

namespace sogm {

// Geometry of one agent's SOGM.  Device layout is time-major slabs:
//     grid[agent][t][z][y][x]      (fp32)
// so one (agent, t) slice is V contiguous floats and x is the fastest axis — the reference's
// risk_maps_[V][T] (plan_env/include/plan_env/map.h:52) is voxel-major.
struct GridGeom {
  int   L, W, H, T;
  int   V;            // L*W*H
  float res;          // VOXEL_RESOLUTION
  float rx, ry, rz;   // local_update_range_* = (N/2) * res in fp32 (risk_base.cpp:26-28)
  float dt;           // map/time_resolution
  float risk_threshold;
  float clearance;
  float ground, ceiling;
  float thr_region, decay_region, decay_voxel;
  int   inf_step;     // (int)(clearance / res) in fp32 (risk_base.cpp:31)
  int   map_kind;
  int   half;         // 1: occupancy stored as __half (SOGM_STORE_F16), arithmetic stays fp32
  int   tile;         // 1: the cells of a slice are stored in 2 x 2 x 2 tiles (SOGM_LAYOUT_TILED), see phys()
  // resample branch of ParticleATC::getParticlesWithRisk (sogm_set_resample; rate 0 = off, the shipped configurations)
  float        rs_rate;  // swarm/replan_risk_rate
  int          rs_n;     // swarm/num_resample
  const float *rs_z;     // injected standard-normal table, device
  int          rs_nz;
  // a / res, correctly rounded.  The compiler's IEEE fp32 division is a dozen instructions (scaling, reciprocal iterations,
  // fix-up, two mode switches) and the stamp computes two or three per mark, the search three per query.  For the one
  // resolution the reference ships (VOXEL_RESOLUTION 0.15, a compile-time constant of map_parameters.h) the three-instruction
  // sequence q0 = a * RN(1/res), e = fma(-res, q0, a) (exact), q = fma(e, RN(1/res), q0) gives RN(a / res) for EVERY
  // float a with |a| in [2^-20, 64) (checked for EVERY such float on the device, tests/test_fast_division.py::test_every_float_on_the_device
  // — sogm_debug_div_check —, and on the CPU for a sample + every float near a voxel boundary; below that both truncate to 0):
  // used for that resolution and that range only, the true division everywhere else.
  float inv_res;
  int   fast_div;
  // largest |offset| of the body particles per axis (sogm_set_body_particles; +inf until they are set): the overlay skips
  // a neighbour whose centre is farther outside the map than that — all of its particles are (splat_item)
  float body_ext[3];
  __host__ __device__ inline float div_res(float a) const {
    if (fast_div && fabsf(a) < 64.0F) {
      const float q0 = a * inv_res;
      const float e  = fmaf(-res, q0, a);
      return fmaf(e, inv_res, q0);
    }
    return a / res;
  }

  // map.h:153-157 — strict inequalities
  __host__ __device__ inline bool in_range(float x, float y, float z) const {
    return x > -rx && x < rx && y > -ry && y < ry && z > -rz && z < rz;
  }
  // map.h:159-162
  __host__ __device__ inline bool in_range(int x, int y, int z) const {
    return x >= 0 && x < L && y >= 0 && y < W && z >= 0 && z < H;
  }
  // map.h:169-174 — fp32 add, fp32 divide, truncation
  __host__ __device__ inline int voxel_of(float x, float y, float z) const {
    const int ix = (int)div_res(x + rx);
    const int iy = (int)div_res(y + ry);
    const int iz = (int)div_res(z + rz);
    return iz * L * W + iy * L + ix;
  }
  // Where cell (x, y, z) of a slice lives.  Rows (tile = 0): z L W + y L + x, x fastest — a 32-byte sector is 8 cells of
  // one x-row.  Tiles (tile = 1; L, W, H even): 2 x 2 x 2 cells = 8 fp32 cells = one sector, tiles in row order, inside a
  // tile z, y, x.  Obstacle surfaces are thin shells in xy that run along z: an x-row sector catches 2.3 of a stamp's
  // marks on average, a tile 4-5, so the stamp writes (and the reset zeroes) about half the sectors; the 5 x 5 window of
  // the collision query at one height is nine 16-byte loads (3 x 3 tiles) instead of five to ten row loads.
  __host__ __device__ inline int phys(int x, int y, int z) const {
    if (!tile) return z * L * W + y * L + x;
    return ((((z >> 1) * (W >> 1) + (y >> 1)) * (L >> 1) + (x >> 1)) << 3) | ((z & 1) << 2) | ((y & 1) << 1) | (x & 1);
  }
  // where the point (x, y, z) of the map frame (in range) is stored: getVoxelIndex's cell (map.h:169-174), or V when that
  // index leaves the array (the reference's out-of-bounds case: a coordinate one ulp below +range).  No integer division
  // on the common path.
  __host__ __device__ inline int cell_of(float x, float y, float z) const {
    return cell_of_xy(x, y, (int)div_res(z + rz));
  }
  // ... with the z index computed by the caller (the stamp's future marks keep their voxel's z: once per voxel)
  __host__ __device__ inline int cell_of_xy(float x, float y, int iz) const {
    const int ix = (int)div_res(x + rx);
    const int iy = (int)div_res(y + ry);
    if (tile && ix < L && iy < W && iz < H) return phys(ix, iy, iz);
    const int v = iz * L * W + iy * L + ix;  // (rows; or an index component equal to its axis size: wraps like the reference's)
    return v < V ? phys_of(v) : V;
  }
  // ... of a LOGICAL cell index (voxel_of's value — which may have wrapped into the next row / layer exactly as the
  // reference's does — decomposed the way corner_of decomposes it)
  __host__ __device__ inline int phys_of(int v) const {
    if (!tile) return v;
    return phys(v % L, (v / L) % W, v / (L * W));
  }
  // ... and back: the logical index of physical cell p
  __host__ __device__ inline int logical_of(int p) const {
    if (!tile) return p;
    const int t = p >> 3, lt = L >> 1, wt = W >> 1;
    const int x = ((t % lt) << 1) | (p & 1), y = (((t / lt) % wt) << 1) | ((p >> 1) & 1), z = ((t / (lt * wt)) << 1) | ((p >> 2) & 1);
    return z * L * W + y * L + x;
  }
  // map.h:186-194 — voxel corner in the world frame
  __host__ __device__ inline void corner_of(int index, const float *pose, float &ox, float &oy,
                                            float &oz) const {
    const int x = index % L;
    const int y = (index / L) % W;
    const int z = index / (L * W);
    ox          = ((float)x * res - rx) + pose[0];
    oy          = ((float)y * res - ry) + pose[1];
    oz          = ((float)z * res - rz) + pose[2];
  }
};

inline GridGeom make_geom(const SogmSpec &s) {
  GridGeom g;
  g.L              = s.L;
  g.W              = s.W;
  g.H              = s.H;
  g.T              = s.T;
  g.V              = s.L * s.W * s.H;
  g.res            = s.resolution;
  g.rx             = (float)(s.L / 2) * s.resolution;
  g.ry             = (float)(s.W / 2) * s.resolution;
  g.rz             = (float)(s.H / 2) * s.resolution;
  g.dt             = s.time_resolution;
  g.risk_threshold = s.risk_threshold;
  g.clearance      = s.clearance;
  g.ground         = s.ground_height;
  g.ceiling        = s.ceiling_height;
  g.thr_region     = s.risk_threshold_region;
  g.decay_region   = s.risk_thres_reg_decay;
  g.decay_voxel    = s.risk_thres_vox_decay;
  g.inf_step       = (int)(s.clearance / s.resolution);
  g.map_kind       = s.map_kind;
  g.half           = (s.storage & 1) == SOGM_STORE_F16 ? 1 : 0;
  g.tile           = (s.storage & SOGM_LAYOUT_TILED) ? 1 : 0;
  g.rs_rate        = 0.0f;
  g.rs_n           = 0;
  g.rs_z           = nullptr;
  g.rs_nz          = 0;
  g.body_ext[0] = g.body_ext[1] = g.body_ext[2] = INFINITY;
  g.inv_res        = (float)(1.0 / (double)s.resolution);
  g.fast_div       = s.resolution == 0.15F && g.inv_res == 6.666666507720947F ? 1 : 0;
  // RiskVoxel::getClearOcccupancy (risk_voxel.cpp:399-423) compares the K-cell sum with the fixed
  // map/risk_threshold_astar: the RiskBase rule with risk_threshold_region = that value and no decay.
  // It inherits MapBase::getObstaclePoints (map.cpp:480-518): fixed risk_threshold as well.
  if (s.map_kind == SOGM_MAP_RISKVOXEL) g.decay_region = g.decay_voxel = 0.0f;
  return g;
}

// Occupancy cells are fp32 or fp16 in HBM (GridGeom::half); every access goes through these.
__device__ inline float cell_ld(const void *slab, size_t i, int half) {
  return half ? __half2float(reinterpret_cast<const __half *>(slab)[i]) : reinterpret_cast<const float *>(slab)[i];
}
__device__ inline void cell_st(void *slab, size_t i, float v, int half) {
  if (half)
    reinterpret_cast<__half *>(slab)[i] = __float2half(v);
  else
    reinterpret_cast<float *>(slab)[i] = v;
}
// += inc: hardware fp32 atomic, or a CAS on the 32-bit word holding the fp16 cell
__device__ inline void cell_add(void *slab, size_t i, float inc, int half) {
  if (!half) {
    __hip_atomic_fetch_add(reinterpret_cast<float *>(slab) + i, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // word and half selected from the cell's ABSOLUTE address: a slab of an odd number of fp16 cells starts on a
  // 2-byte boundary (odd V), where "i & 1" would pick the neighbour's half
  const uintptr_t cp = reinterpret_cast<uintptr_t>(slab) + i * 2;
  unsigned *w  = reinterpret_cast<unsigned *>(cp & ~(uintptr_t)3);
  const int hi = (int)((cp >> 1) & 1);
  unsigned  old = *w, assumed;
  do {
    assumed = old;
    const unsigned short hb = (unsigned short)((assumed >> (16 * hi)) & 0xffffu);
    const float          f  = __half2float(__ushort_as_half(hb)) + inc;
    const unsigned       nb = (unsigned)__half_as_ushort(__float2half(f));
    const unsigned       nw = hi ? ((assumed & 0x0000ffffu) | (nb << 16)) : ((assumed & 0xffff0000u) | nb);
    old = atomicCAS(w, assumed, nw);
  } while (old != assumed);
}

// Read-only view of the batched map handed to every kernel that queries it.
struct MapView {
  GridGeom      g;
  const char   *grid;    // [A][T][V] cells of 4 (fp32) or 2 (fp16) bytes
  const float  *poses;   // [A][3]
  const double *stamps;  // [A]
  int           n_agents;

  __device__ inline const void *slab(int agent, int t) const {
    return grid + ((size_t)agent * g.T + t) * (size_t)g.V * (g.half ? 2 : 4);
  }
};

// The (2S+1)^2 window of getClearOcccupancy at one height (fkpcp map: the z loop degenerates): every gather is
// issued first so the loads overlap, then the reference's running sum is replayed in kernel order (x outer, y inner)
// WITHOUT divergent control flow — a cell outside the grid is read at a clamped address and masked to +0.0f
// (x + 0 = x exactly), and "return 1 at the first prefix sum above the threshold" is the OR of the prefix
// comparisons (49 nested early exits cost ~30 instructions and two spilled exec masks per cell).  The window size is
// a template parameter and the clamps / range tests are per row and per column, ~4 instructions per cell: with a
// run-time window (k / w, k % w, six range compares and a 64-bit address per cell) this one query was 5.5 of the
// 7 us of an A* child evaluation — instruction-bound, not memory-bound.  (ix, iy, iz) is inside the grid.
template <int S>
__device__ inline void window_gather(const char *base, const GridGeom &g, int ix, int iy, int iz,
                                     float (&v)[2 * S + 1][2 * S + 1]) {  // v[x][y]
  constexpr int W  = 2 * S + 1;
  const int     sh = g.half ? 1 : 2;
  unsigned      col[W], colok[W];
#pragma unroll
  for (int a = 0; a < W; ++a) {
    const int qx = ix + a - S;
    colok[a]     = (unsigned)qx < (unsigned)g.L ? 0xFFFFFFFFu : 0u;
    col[a]       = (unsigned)min(max(qx, 0), g.L - 1) << sh;
  }
  if (!g.half && ix - S >= 0 && ix + S < g.L) {
    // fp32 cells, window inside the grid in x: a row's W cells are one run of dwords (4-byte aligned), fetched with
    // one or two wide loads instead of W
    struct __attribute__((packed, aligned(4))) Row {
      float c[W];
    };
    const unsigned x0 = (unsigned)(ix - S) << 2;
#pragma unroll
    for (int b = 0; b < W; ++b) {
      const int      qy    = iy + b - S;
      const unsigned rowok = (unsigned)qy < (unsigned)g.W ? 0xFFFFFFFFu : 0u;
      const unsigned row   = ((unsigned)(iz * g.W + min(max(qy, 0), g.W - 1)) * (unsigned)g.L) << 2;
      const Row      r     = *reinterpret_cast<const Row *>(base + (row + x0));
#pragma unroll
      for (int a = 0; a < W; ++a) v[a][b] = __uint_as_float(__float_as_uint(r.c[a]) & rowok);
    }
  } else {
#pragma unroll
    for (int b = 0; b < W; ++b) {
      const int      qy    = iy + b - S;
      const unsigned rowok = (unsigned)qy < (unsigned)g.W ? 0xFFFFFFFFu : 0u;
      // byte offset of the row inside the slab: V * 4 < 2^32 is checked where the map is created
      const unsigned row = ((unsigned)(iz * g.W + min(max(qy, 0), g.W - 1)) * (unsigned)g.L) << sh;
#pragma unroll
      for (int a = 0; a < W; ++a) {
        const char *p   = base + (row + col[a]);
        const float val = g.half ? __half2float(*reinterpret_cast<const __half *>(p)) : *reinterpret_cast<const float *>(p);
        v[a][b]         = __uint_as_float(__float_as_uint(val) & (rowok & colok[a]));
      }
    }
  }
}
// The same window from a tiled slice (GridGeom::phys): the (2S+1)^2 cells at height iz lie in (S+1)^2 tiles at most; one
// 16-byte load (8 bytes for fp16 cells) fetches a tile's four cells of that height, a tile is inside or outside the grid
// as a whole, and a window cell picks its value with compile-time tile / cell indices selected by the parities of the
// window's corner (no dynamically indexed registers).
template <int S>
__device__ inline void window_gather_tiled(const char *base, const GridGeom &g, int ix, int iy, int iz,
                                           float (&v)[2 * S + 1][2 * S + 1]) {  // v[x][y]
  constexpr int W = 2 * S + 1, NT = S + 1;
  const int     x0 = ix - S, y0 = iy - S;
  const bool    px = (x0 & 1) != 0, py = (y0 & 1) != 0;
  const int     tx0 = x0 >> 1, ty0 = y0 >> 1;  // (arithmetic shift: floors for negative corners)
  const int     lt = g.L >> 1, wt = g.W >> 1;
  const unsigned zrow = (unsigned)(iz >> 1) * (unsigned)wt, zoff = (unsigned)(iz & 1) << 2;
  float          t[NT][NT][4];  // [tile y][tile x][(y & 1) * 2 + (x & 1)]
#pragma unroll
  for (int tb = 0; tb < NT; ++tb) {
    const int      ty   = ty0 + tb;
    const unsigned tyok = (unsigned)ty < (unsigned)wt ? 0xFFFFFFFFu : 0u;
    const unsigned row  = (zrow + (unsigned)min(max(ty, 0), wt - 1)) * (unsigned)lt;
#pragma unroll
    for (int ta = 0; ta < NT; ++ta) {
      const int      tx  = tx0 + ta;
      const unsigned ok  = tyok & ((unsigned)tx < (unsigned)lt ? 0xFFFFFFFFu : 0u);
      const unsigned off = ((row + (unsigned)min(max(tx, 0), lt - 1)) << 3) + zoff;  // cells
      if (g.half) {
        const uint2 h = *reinterpret_cast<const uint2 *>(base + ((size_t)off << 1));
        const __half2 a = *reinterpret_cast<const __half2 *>(&h.x), b = *reinterpret_cast<const __half2 *>(&h.y);
        t[tb][ta][0] = __uint_as_float(__float_as_uint(__low2float(a)) & ok);
        t[tb][ta][1] = __uint_as_float(__float_as_uint(__high2float(a)) & ok);
        t[tb][ta][2] = __uint_as_float(__float_as_uint(__low2float(b)) & ok);
        t[tb][ta][3] = __uint_as_float(__float_as_uint(__high2float(b)) & ok);
      } else {
        const uint4 q = *reinterpret_cast<const uint4 *>(base + ((size_t)off << 2));
        t[tb][ta][0] = __uint_as_float(q.x & ok);
        t[tb][ta][1] = __uint_as_float(q.y & ok);
        t[tb][ta][2] = __uint_as_float(q.z & ok);
        t[tb][ta][3] = __uint_as_float(q.w & ok);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < W; ++a)
#pragma unroll
    for (int b = 0; b < W; ++b) {
      // window cell (a, b) is cell (a + px, b + py) counted from the corner tile's origin
      const float c00 = t[b >> 1][a >> 1][((b & 1) << 1) | (a & 1)];
      const float c10 = t[b >> 1][(a + 1) >> 1][((b & 1) << 1) | ((a + 1) & 1)];
      const float c01 = t[(b + 1) >> 1][a >> 1][(((b + 1) & 1) << 1) | (a & 1)];
      const float c11 = t[(b + 1) >> 1][(a + 1) >> 1][(((b + 1) & 1) << 1) | ((a + 1) & 1)];
      v[a][b]         = py ? (px ? c11 : c01) : (px ? c10 : c00);
    }
}
template <int S>
__device__ inline int window_replay(const float (&v)[2 * S + 1][2 * S + 1], float thr) {
  constexpr int W   = 2 * S + 1;
  float         sum = 0.0F;
  bool          hit = false;
#pragma unroll
  for (int a = 0; a < W; ++a)
#pragma unroll
    for (int b = 0; b < W; ++b) {
      sum += v[a][b];
      hit = hit || sum > thr;
    }
  return hit ? 1 : 0;
}
template <int S>
__device__ inline int window_sum_hits(const char *base, const GridGeom &g, int ix, int iy, int iz, float thr) {
  float v[2 * S + 1][2 * S + 1];
  if (g.tile)
    window_gather_tiled<S>(base, g, ix, iy, iz, v);
  else
    window_gather<S>(base, g, ix, iy, iz, v);
  return window_replay<S>(v, thr);
}

// getClearOcccupancy(pos, int t): fake_particle_risk_voxel.cpp:309-331 / risk_base.cpp:228-251.
// Summation order = inflate-kernel build order (x, y, z nested) so the fp32 running sum and its
// early exit are identical to the reference.
__device__ inline int query_clear_idx(const MapView &m, int agent, double px, double py, double pz,
                                      int t) {
  const GridGeom &g = m.g;
  if (g.map_kind == SOGM_MAP_FAKE) {
    if (pz < (double)g.ground || pz > (double)g.ceiling) return -1;
  } else {
    if (pz < (double)g.ground) return 1;
    if (pz > (double)g.ceiling) return 1;
  }
  const float *pose = m.poses + agent * 3;
  const float  fx = (float)px - pose[0], fy = (float)py - pose[1], fz = (float)pz - pose[2];
  const int    ix = (int)g.div_res(fx + g.rx);
  const int    iy = (int)g.div_res(fy + g.ry);
  const int    iz = (int)g.div_res(fz + g.rz);
  if (!g.in_range(ix, iy, iz)) return -1;
  const void  *sl  = m.slab(agent, t);
  const int    s   = g.inf_step;
  const int    zs  = g.map_kind == SOGM_MAP_FAKE ? 0 : s;  // fake map: z loop degenerates (:41)
  const float  thr = g.map_kind == SOGM_MAP_FAKE ? g.risk_threshold
                                                 : g.thr_region - (float)t * g.decay_region;
  float        sum = 0.0F;
  if (zs == 0 && s <= 3) {
    // Common case (fkpcp map, K = (2s+1)^2 <= 49): see window_sum_hits
    const char *base = reinterpret_cast<const char *>(sl);
    switch (s) {
      case 0: return window_sum_hits<0>(base, g, ix, iy, iz, thr);
      case 1: return window_sum_hits<1>(base, g, ix, iy, iz, thr);
      case 2: return window_sum_hits<2>(base, g, ix, iy, iz, thr);
      default: return window_sum_hits<3>(base, g, ix, iy, iz, thr);
    }
  }
  for (int x = -s; x <= s; ++x) {
    const int qx = ix + x;
    for (int y = -s; y <= s; ++y) {
      const int qy = iy + y;
      for (int z = -zs; z <= zs; ++z) {
        const int qz = iz + z;
        if (!g.in_range(qx, qy, qz)) continue;
        sum += cell_ld(sl, (size_t)g.phys(qx, qy, qz), g.half);
        if (sum > thr) return 1;
      }
    }
  }
  return 0;
}

// getClearOcccupancy(pos, double dt): slice = floor(dt / time_resolution) clamped to T-1
// (fake_particle_risk_voxel.cpp:341-346).  Negative slices (UB in the reference) clamp to 0.
__device__ inline int query_clear_time(const MapView &m, int agent, double px, double py,
                                       double pz, double dt) {
  int tf = (int)floor(dt / (double)m.g.dt);
  tf     = tf > (m.g.T - 1) ? m.g.T - 1 : tf;
  tf     = tf < 0 ? 0 : tf;
  return query_clear_idx(m, agent, px, py, pz, tf);
}

// Bezier::getPos (traj_utils/src/bernstein.cpp:25-34, bernstein.hpp:164-178):
// p(t) = C^T * A4 * [1, s, s^2, s^3, s^4],  s = (t - t0) / (tf - t0), evaluated left to right.
__device__ inline void bezier_pos(const SogmTrajRecord &r, double t, double out[3]) {
  int    piece = r.n_pieces - 1;
  double tt    = t;
  for (int i = 0; i < r.n_pieces; ++i) {
    tt -= r.duration[i];
    if (tt < 0) {
      piece = i;
      break;
    }
  }
  double t0 = 0;
  for (int k = 0; k < piece; ++k) t0 += r.duration[k];
  const double tf  = t0 + r.duration[piece];
  const double dur = tf - t0;
  const double s   = (t - t0) / dur;
  double       S[5];
  S[0] = 1.0;
  S[1] = s;
  S[2] = s * s;
  S[3] = s * s * s;
  S[4] = (s * s) * (s * s);
  const double A[5][5] = {{1, -4, 6, -4, 1},
                          {0, 4, -12, 12, -4},
                          {0, 0, 6, -12, 6},
                          {0, 0, 0, 4, -4},
                          {0, 0, 0, 0, 1}};
  const double *c = r.cpts + piece * 15;
  for (int d = 0; d < 3; ++d) {
    double acc = 0.0;
    for (int j = 0; j < 5; ++j) {
      double b = 0.0;
      for (int i = 0; i < 5; ++i) b += c[i * 3 + d] * A[i][j];
      acc += b * S[j];
    }
    out[d] = acc;
  }
}

}  // namespace sogm

// ---- tuning knobs (sogm_set_tuning / sogm_get_tuning): one table per context, defaults below, no environment ----
// X(id, key, default, smallest, largest value sogm_set_tuning accepts)
#define SOGM_TUNING_TABLE(X)                                                                                             \
  X(GROUPS, "groups", 2, 1, 64)                   /* [planner create] agent groups of the grouped-stream replan            */  \
  X(SPEC_ASTAR, "spec_astar", 1, 0, 1)           /* [planner create] speculative second search beside the first          */  \
  X(CLEAR_GATE_FRAC, "clear_gate_frac", 1, 0, 1) /* [planner create] fraction of agents with final corridors that opens the wide clear */ \
  X(QP_WGS, "qp_wgs", 0, 0, 1024)                   /* persistent QP workgroups of the dataflow replan; 0 = half the CUs    */  \
  X(QP_ABLATE, "qp_ablate", 0, 0, 255)             /* phase ablation mask (only in -DSOGM_QP_ABLATE_BUILD libraries)       */  \
  X(CLEAR_WGS, "clear_wgs", 0, 0, 65536)             /* dense clear: workgroups; 0 = 64 / 80 polite, 2048 alone              */  \
  X(CLEAR_THROTTLE, "clear_throttle", 0, 0, 64)   /* dense clear: stores in flight per wave when clear_wgs is set         */  \
  X(CLEAR_NT, "clear_nt", 1, 0, 1)               /* dense clear: non-temporal stores                                      */  \
  X(CLEAR_WIDE_WGS, "clear_wide_wgs", 256, 0, 65536) /* dense clear: workgroups of the gated wide launch; 0 = fixed width    */  \
  X(CLEAR_WIDE_BOUND, "clear_wide_bound", 0, 0, 64) /* dense clear: stores in flight per wave of the wide launch; 0 = unbounded */ \
  X(CLEAR_HEAD_GB, "clear_head_gb", 1.0e9, 0, 1.0e12) /* dense clear: GB of the narrow head before a full-width rest          */  \
  X(CLEAR_EARLY, "clear_early", 0, 0, 1)         /* dense clear: queue the swapped-out grid's clear under the stamp      */  \
  X(CLEAR_RETIRE_AT_END, "clear_retire_at_end", 0, 0, 1) /* dense clear: the wide launch retires when the replan ends    */  \
  X(RESET_WGS, "reset_wgs", 32, 0, 4096)            /* sparse reset: workgroups per agent                                    */  \
  X(RESET_LANES, "reset_lanes", 0, 0, 4)         /* sparse reset: 2 / 4 lanes per entry; 0 = 2 under the replan, 4 alone  */  \
  X(RESET_UNROLL, "reset_unroll", 0, 0, 8)       /* sparse reset: 1 / 8 entries per trip; 0 = 1 under the replan, 8 alone */  \
  X(RESET_LATE, "reset_late", 1, 0, 1)           /* sparse reset: held back until every agent's corridors are final       */  \
  X(STAMP_WGS, "stamp_wgs", 256, 0, 4096)           /* stamp: one-wave workgroups per agent                                  */  \
  X(STAMP_BITS_WGS, "stamp_bits_wgs", 0, 0, 4096)   /* stamp: one-wave workgroups per agent of the occupancy-bits pass; 0 = stamp_wgs */  \
  X(STAMP_LDS_LOG, "stamp_lds_log", 1, 0, 1)        /* stamp: the marks' log pass reads the sectors back from LDS instead of recomputing them */  \
  X(STAMP_CACHED, "stamp_cached", 0, 0, 1)          /* stamp: the marks' register-cached single-pass slice loops (the persistent kernels' form) */  \
  X(STAMP_LDS_KB, "stamp_lds_kb", 0, 0, 64)              /* stamp: unused dynamic LDS per marks workgroup (bounds waves per CU) */  \
  X(SPLAT_WGS, "splat_wgs", 256, 0, 65536)           /* overlay launched under a pre-stamp's tail: workgroups                 */  \
  X(SPLAT_OVERLAP, "splat_overlap", 1, 0, 1)     /* 0: sogm_replan joins the pre-stamp's end itself                       */  \
  X(PRESTAMP_BITS, "prestamp_bits", 32, 0, 1024)    /* pre-stamp: one-wave tickets per agent, occupancy bits pass            */  \
  X(PRESTAMP_MARKS, "prestamp_marks", 64, 0, 1024)  /* pre-stamp: one-wave tickets per agent, marks pass                     */  \
  X(PRESTAMP_WGS, "prestamp_wgs", 0, 0, 65536)       /* pre-stamp: one-wave workgroups; 0 = 8 per CU                          */  \
  X(PRESTAMP_GATE_FRAC, "prestamp_gate_frac", 0.9, 0, 1) /* pre-stamp: fraction of the agents whose corridors must be final before it starts */ \
  X(PRESTAMP_STREAM, "prestamp_stream", 1, 0, 1) /* pre-stamp on a stream of its own behind its target grid's reset EVENT; 0 = on the resets' stream */ \
  X(PRESTAMP_LATE_AGENTS, "prestamp_late_agents", 8, 0, 65536)  /* the last agents to be published get finer tickets ...     */  \
  X(PRESTAMP_LATE_BITS, "prestamp_late_bits", 128, 0, 4096)    /* ... this many for the bits pass                           */  \
  X(PRESTAMP_LATE_MARKS, "prestamp_late_marks", 256, 0, 4096)  /* ... and for the marks pass                                */  \
  X(FLIGHT_QP_UNITS, "flight_qp_units", 4, 1, 13)          /* flight: 16-CU units of the QP kernel                            */  \
  X(FLIGHT_SEARCH_UNITS, "flight_search_units", 2, 1, 13)  /* flight: ... of the search kernel                                */  \
  X(FLIGHT_MAP_UNITS, "flight_map_units", 4, 1, 13)        /* flight: ... of the map kernel (corridor + finish: the rest)     */  \
  X(FLIGHT_MASKS, "flight_masks", 1, 0, 1)                 /* flight: streams with compute-unit masks                         */  \
  X(FLIGHT_SPEC, "flight_spec", 1, 0, 1)                   /* flight: both search attempts side by side                       */  \
  X(FLIGHT_RESET, "flight_reset", 8, 1, 256)               /* flight: one-wave tickets per agent-tick, sparse reset           */  \
  X(FLIGHT_BITS, "flight_bits", 16, 1, 256)                /* flight: ... occupancy bits                                      */  \
  X(FLIGHT_MARKS, "flight_marks", 32, 1, 256)              /* flight: ... marks                                               */  \
  X(FLIGHT_SPLAT, "flight_splat", 4, 1, 64)                /* flight: ... neighbour overlay                                   */  \
  X(FLIGHT_ADMIT, "flight_admit", 48, 1, 65536)            /* flight: agents whose map may be under construction at once      */  \
  X(FLIGHT_PACE_US, "flight_pace_us", 20, 0, 100000)       /* flight: microseconds between two admissions to the map stage    */  \
  X(FLIGHT_HEADS, "flight_heads", 32, 1, 4096)             /* flight: admitting waves of the map kernel                       */  \
  X(FLIGHT_URGENT, "flight_urgent", 8, 0, 65536)           /* flight: the last n finishers of a tick build their next map in the urgent lane; 0 = none */ \
  X(FLIGHT_URGENT_WAVES, "flight_urgent_waves", 4096, 1, 4096) /* flight: map workers that look at the urgent queue first (all of them by default) */ \
  X(FLIGHT_URGENT_FINE, "flight_urgent_fine", 4, 1, 16)    /* flight: the urgent lane's maps in this many times more tickets  */  \
  X(FLIGHT_GATE_PACE_US, "flight_gate_pace_us", 20, 0, 100000) /* flight: microseconds between two overlays a gate releases     */  \
  X(FLIGHT_NEIGHBOUR_LAG, "flight_neighbour_lag", 2, 1, 2) /* flight: tick k reads the neighbours' records of tick k - this (1 = the reference's staleness) */ \
  X(FLIGHT_ENGINES, "flight_engines", 4, 1, 4)             /* flight: shader engines (of every XCD) the four kernels share ... */ \
  X(FLIGHT_ENGINE_FIRST, "flight_engine_first", 0, 0, 3)   /* flight: ... starting with this one (two flights side by side on one device: tests) */ \
  X(FLIGHT_EXCHANGE_UNITS, "flight_exchange_units", 0, 0, 4) /* flight: 16-CU units left to NO kernel (room for the collective's kernels beside a multi-rank flight) */ \
  X(FLIGHT_LIGHT_PER_CU, "flight_light_per_cu", 4, 1, 4)   /* flight: corridor + finish waves per compute unit of their partition (4 = every SIMD) */ \
  X(FLIGHT_MAP_PER_CU, "flight_map_per_cu", 8, 1, 8)       /* flight: map waves per compute unit of their partition            */ \
  X(UPDATE_FLOW, "update_flow", 0, 0, 1)                   /* sogm_update_world builds the maps agent by agent on a stream of its own; sogm_replan's searches start per agent */ \
  X(UPDATE_BITS, "update_bits", 16, 1, 256)                /* update flow: one-wave tickets per agent, occupancy bits         */  \
  X(UPDATE_MARKS, "update_marks", 64, 1, 256)              /* update flow: ... marks                                          */  \
  X(UPDATE_SPLAT, "update_splat", 40, 1, 64)               /* update flow: ... neighbour overlay                              */  \
  X(UPDATE_WGS, "update_wgs", 0, 0, 65536)                 /* update flow: one-wave workgroups; 0 = 16 per CU                 */  \
  X(UPDATE_CHUNK, "update_chunk", 1, 1, 64)                /* update flow: consecutive tickets per claim                      */  \
  X(UPDATE_CACHED, "update_cached", 0, 0, 1)               /* update flow: the marks' register-cached single-pass form        */  \
  X(UPDATE_ORDER, "update_order", 1, 0, 1)                 /* update flow: agents in the order of their previous chain's length, longest first */
enum {
#define X(id, name, dflt, lo, hi) SOGM_TUNE_##id,
  SOGM_TUNING_TABLE(X)
#undef X
  SOGM_TUNE_N
};

// ---- context (opaque in the C ABI) ----
struct sogm_ctx {
  SogmSpec       spec;
  sogm::GridGeom geom;
  int            n_agents;
  int            device;
  float         *d_grid;    // [A][T][V] cells (fp32, or __half when geom.half)
  size_t         cell_bytes() const { return geom.half ? 2 : 4; }
  float         *d_poses;   // [A][3]
  double        *d_stamps;  // [A]
  double        *d_body;    // [n_body][3]
  int            n_body;
  int            updated;
  float         *d_scratch_vt;  // [V][T] staging for download / upload
  int            overlap;     // tick pipelining: 0 off, 1 pre-clear in place, 2 / 3 pre-clear of spare grids
  // modes 2 and 3: d_grid rotates through a pool of 2 / 3 grids.  `ready` = spares whose clear has been queued on
  // the side stream (FIFO; the next update adopts the front one after waiting for its event), `dirty` = spares
  // that still hold an old map (the next sogm_replan queues their clear).  Mode 1 uses ev_cleared only.
  float         *pool[3];
  hipEvent_t     pool_ev[3];
  int            n_pool, cur_idx;
  int            ready[2], n_ready;
  int            dirty[2], n_dirty;
  int            precleared;  // the next update finds a (being-)cleared grid: mode 1 in place, modes 2 / 3 n_ready > 0
  hipStream_t    side;
  hipStream_t    pstream;     // the pre-stamp's stream (tuning key prestamp_stream; the resets stay on `side`)
  hipEvent_t     ev_gate_frac;  // (the same for prestamp_gate_frac < 1: recorded behind a gate kernel with that share of the agents as its target)
  int            gate_frac_valid, gate_frac_agents;  // gate_frac_agents: set per replan before queue_spare_clears (0: no such gate)
  hipEvent_t     ev_gate_open;  // recorded on `side` behind the reset's gate kernel of the replan being queued ("every
  int            gate_open_valid;  // agent's corridors are final"): the pre-stamp's stream waits for it instead of spinning
  hipEvent_t     ev_grid_free, ev_cleared;
  // width-adaptive clear (dataflow replan, modes 2 / 3): the side-stream clear starts narrow; a second, wide launch
  // on side2 joins it once the word *clear_gate reaches clear_gate_target (the planner's "corridors final" counter:
  // from then on the tick's remaining kernels iterate in LDS); clear_cursor is the chunk counter the two launches
  // share.  clear_gate == nullptr: one launch of fixed width.
  // The planner registers the gate at creation (dataflow replan).  *clear_epoch_word names the replan in flight: a
  // replan writes a fresh epoch (clear_epoch) after resetting its counters, the next update writes 0.  The gate only
  // trusts the counter while the word equals its epoch (a later one: that replan is over, open), and the wide
  // launch takes chunks only while it does — it retires when the next tick begins, the narrow launch goes on.
  const int          *clear_gate;
  const int          *clear_gate_err;
  int                 clear_gate_target;
  int                *clear_epoch_word;
  int                 clear_epoch, clear_epoch_ahead;
  int                 wide_clear_pending;  // a dense adaptive clear was queued since the last update wrote epoch 0
  unsigned long long *clear_cursor;  // [2], used in turn (clear_seq)
  unsigned            clear_seq;
  hipStream_t         side2;  // the wide part's stream
  hipEvent_t          ev_side2_go, ev_side2_done;
  float         *d_filter_cells;   // filterPointCloud leaf accumulators [A][max_cells][4] (lazy)
  void          *d_filter_box;
  int           *d_filter_blocks;
  int            filter_max_cells;
  unsigned      *d_stamp_bits;     // [A][ceil(V / 32)] occupancy bits of slice 0 between k_stamp_bits and k_stamp_marks (lazy)
  // Sparse reset.  The reference rebuilds the map from zero at every update (fake_particle_risk_voxel.cpp:107-108:
  // a fill over all V x T cells); here every mark written into a grid since its last reset is logged as the index
  // of its 32-byte sector (per agent), and the reset zeroes exactly those sectors — the cells of the rebuilt map
  // are the same, the 640 MB per agent of zero stores are not issued.  tracked[s] = every non-zero cell of slot s
  // is covered by its log (false after dense writers — sogm_set_future_risk, sogm_dsp_publish, sogm_grid_ptr —
  // and for a fresh allocation: the next reset of that slot is the dense clear).  A log that overflows makes the
  // reset kernel zero that agent's whole grid.
  // pre-stamp (sogm_planner_set_prestamp): the replan builds the next tick's map into pool slot prestamp_slot (-1:
  // none) with the next map centres / stamps in d_poses_next / d_stamps_next; sogm_update_prestamped adopts both
  float         *d_poses_next;
  double        *d_stamps_next;
  int            prestamp_slot;
  int            sparse;           // feature switch (sogm_set_sparse_reset; default on, SOGM_SPARSE_RESET=0 turns it off)
  int            log_cap;          // entries per agent
  unsigned      *d_log[3];         // [A][log_cap] per pool slot (slot 0 = the only grid without a pool), lazy
  unsigned      *d_log_n[3];       // [A] entries appended since the slot's last reset (beyond log_cap: overflow)
  int            tracked[3];
  unsigned long long *d_reset_stat;  // [8]: {entries read, launches, bytes zeroed, -} of k_reset_sectors since the last
                                     // state query; {marks written, entries logged, -, -} of the stamp (sogm_map_traffic)
  long long      n_stamps;           // stamps launched since the last sogm_map_traffic reset (host count)
  long long     *h_tick_clock;       // pinned, device-visible [4]: wall_clock64 (100 MHz) of {the last update's first kernel,
                                     // the last replan's report, the last sogm_device_clock kernel, -} (sogm_tick_clock)
  // history of each pool slot since the pool was (re)built: resets through its log, dense clears (host-side launch
  // counts), and whether the CURRENT grid was built by a replan's pre-stamp (sogm_grid_history: lets a parity test
  // assert that the grid it compares went through k_reset_sectors and k_prestamp_flow)
  int            hist_sparse[3], hist_dense[3];
  int            cur_prestamped;
  void          *d_cand;           // [A][1024] candidate cylinders of the stamp (k_cull_cylinders)
  int           *d_ncand;          // [A]
  int           *d_blk_list;       // [A][blk_cap] block ids of each agent's crop of a SogmWorld cloud (lazy)
  int           *d_blk_n;          // [A]
  int            blk_cap;
  // trajectory exchange (sogm_traj_allgather): its own stream, ordered against producers / consumers by events
  hipStream_t    xstream;
  hipEvent_t     ev_xin, ev_xdone;
  int            exchange_pending;
  // set by a publishing sogm_replan: the event after which `records_final_ptr` (the host's own records) and every
  // reader of the swarm table inside that replan are done — the finishing kernel's end.  sogm_traj_allgather of exactly
  // those records starts from it instead of from the caller's stream position, i.e. under the pre-stamp's tail.
  hipEvent_t            ev_records_final;
  const SogmTrajRecord *records_final_ptr;
  int                   records_final_valid;
  // set by a pre-stamping sogm_replan: the caller's stream is NOT joined to the pre-stamp's end inside that call — the
  // next update's overlay starts behind the replan's fan-in and waits per agent for ps_stage[agent] == FLOW_PS_DONE,
  // under the pre-stamp's tail — but by the first later call that needs it (join_prestamp)
  hipEvent_t            ev_pdone;
  int                   pdone_pending;
  const int            *ps_stage;  // the planner's per-agent pre-stamp progress words, ps_err its error word
  int                  *ps_err;
  // the pre-stamping planner's pinned failure words {code, failed ticks} and the count when the pre-stamp was queued:
  // a pre-stamped grid whose replan failed since is refused by sogm_update_prestamped (the host falls back to
  // sogm_update_gt*, which discards the grid and clears the stamp's bitmask)
  volatile int         *ps_fail_host;
  int                   ps_fail_seen;
  // update flow (tuning key update_flow): sogm_update_world leaves the caller's stream after the cull and builds the
  // maps agent by agent on `ustream` (k_update_flow: tickets, per agent bits -> marks -> overlay); the agent's last ticket
  // stores map_epoch into d_map_ready[agent].  sogm_replan's searches are launched beside it and wait per agent; every
  // other reader or writer of the current grid joins the flow's end first (join_update).
  hipStream_t    ustream;
  hipEvent_t     ev_uin, ev_udone;
  int            update_pending;
  int            map_epoch;
  int           *d_map_ready;      // [A]
  int           *d_update_ctl;     // [8 + A]: {ticket, error, -...} + per-agent progress counters
  long long     *d_update_ts;      // [A][4] diagnostics (sogm_debug_update_flow_times)
  int           *d_update_order;   // [A] agents in the order the flow takes them (identity until a replan ranks them)
  double         tune[SOGM_TUNE_N];  // sogm_set_tuning; defaults from SOGM_TUNING_TABLE at sogm_create
  int            tune_i(int k) const { return (int)tune[k]; }
  int            profiling;   // bit k: slot k is timed (sogm_set_profiling: all, sogm_set_profiling_slots: a choice)
  // per-slot ring of HIP event pairs: every launch of a profiled kernel since profiling was enabled keeps its own
  // pair, so a run can be timed launch by launch WITHOUT synchronising between launches (sogm_profile_read_all)
  hipEvent_t    *ring[SOGM_PROF_N];    // [SOGM_PROF_RING][2], created lazily
  long long      ring_n[SOGM_PROF_N];  // launches recorded since sogm_set_profiling(1)
};
#define SOGM_PROF_RING 1024

namespace sogm {
// Internal streams.  (CU-masked variants — the clear or the QP stage on CUs of their own — were measured in rounds 3
// and 4 and lost: the latency-bound planner waves need the whole machine, profiles/EXPERIMENTS.md.)
inline hipError_t create_stream_partitioned(hipStream_t *st, int /*role*/) {
  return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}

// RAII-free helper: record the begin/end events of profiling slot `slot` on `st`.
inline hipEvent_t *prof_pair(sogm_ctx *c, int slot, long long n) {
  if (!c->ring[slot]) {
    c->ring[slot] = new (std::nothrow) hipEvent_t[2 * SOGM_PROF_RING]();
    if (!c->ring[slot]) return nullptr;
  }
  hipEvent_t *p = c->ring[slot] + 2 * (n % SOGM_PROF_RING);
  if (!p[0] && (hipEventCreate(&p[0]) != hipSuccess || hipEventCreate(&p[1]) != hipSuccess)) return nullptr;
  return p;
}
inline void prof_begin(sogm_ctx *c, int slot, hipStream_t st) {
  if (!((c->profiling >> slot) & 1)) return;
  if (hipEvent_t *p = prof_pair(c, slot, c->ring_n[slot])) (void)hipEventRecord(p[0], st);
}
inline void prof_end(sogm_ctx *c, int slot, hipStream_t st) {
  if (!((c->profiling >> slot) & 1)) return;
  if (hipEvent_t *p = prof_pair(c, slot, c->ring_n[slot])) {
    (void)hipEventRecord(p[1], st);
    c->ring_n[slot]++;
  }
}
}  // namespace sogm

namespace sogm {
inline MapView view_of(const sogm_ctx *c) {
  MapView m;
  m.g        = c->geom;
  m.grid     = reinterpret_cast<const char *>(c->d_grid);
  m.poses    = c->d_poses;
  m.stamps   = c->d_stamps;
  m.n_agents = c->n_agents;
  return m;
}
void set_error(const char *what, hipError_t e);
void set_error_text(const char *text);
// readers of the swarm's records wait (on their own stream) for an all-gather still in flight
// the caller's stream waits for the end of the last replan's pre-stamp (and its report), if it has not yet
inline int join_prestamp(sogm_ctx *c, hipStream_t st) {
  if (c->pdone_pending) {
    if (hipStreamWaitEvent(st, c->ev_pdone, 0) != hipSuccess) return SOGM_ERR_HIP;
    c->pdone_pending = 0;
  }
  c->ps_stage = nullptr;
  return SOGM_OK;
}
// the caller's stream waits for the end of an update flow still building the current grid
inline int join_update(sogm_ctx *c, hipStream_t st) {
  if (c->update_pending) {
    if (hipStreamWaitEvent(st, c->ev_udone, 0) != hipSuccess) return SOGM_ERR_HIP;
    c->update_pending = 0;
  }
  return SOGM_OK;
}
// csrc/sogm_exchange.hip: the context's exchange stream (created on first use), a raw all-gather queued on it in ITS order (no
// event from a caller's stream: the caller has queued whatever the collective must wait for on that stream itself), and the
// completion mark consumers join through join_exchange
int exchange_stream(sogm_ctx *c, hipStream_t *out);
int exchange_allgather_raw(sogm_ctx *c, void *nccl_comm, const void *send, void *recv, size_t bytes_per_rank);
int exchange_mark_pending(sogm_ctx *c);
inline int join_exchange(sogm_ctx *c, hipStream_t st) {
  if (c->exchange_pending && hipStreamWaitEvent(st, c->ev_xdone, 0) != hipSuccess) return SOGM_ERR_HIP;
  return SOGM_OK;
}
int  launch_clear(sogm_ctx *c, hipStream_t st, float *grid = nullptr, bool polite = false, int part = 0,
                  size_t split = 0);
size_t clear_vec4_total(const sogm_ctx *c);
// next update's grid becomes current (modes 2 / 3) and the stream waits for its pre-clear
int  adopt_preclear(sogm_ctx *c, hipStream_t st, bool join = true);
int  retire_wide_clear(sogm_ctx *c, hipStream_t st);
int  announce_clear_epoch(sogm_ctx *c, hipStream_t st);
int  next_clear_epoch(sogm_ctx *c);  // the epoch a replan writes itself (k_flow_reset) instead of a launch of its own
int  queue_spare_clears(sogm_ctx *c, hipEvent_t after);
// device view of a SogmWorld cloud (blocks of consecutive points with xy bounds) + the per-agent crop lists the stamp builds
struct CloudBlocks {
  const float *bounds;  // [n_blocks][4] {xmin, xmax, ymin, ymax}; null = the caller's per-agent {begin, end} ranges are used
  int          n_blocks, block_points, n_points;
  int         *list;    // [A][row] ids of the blocks that intersect the agent's window, ascending
  int         *n_list;  // [A]
  int          row;     // row length of `list` (>= n_blocks).  NOT n_blocks itself: a flight's agents are on different ticks at
                        // once, their frames may hold different block counts, and they share the lists — a row stride that
                        // followed the frame would let one agent's head overwrite a list another agent's bits tickets read
};
// the context's crop lists sized for `w` (grown on demand; stream-ordered)
int world_blocks(sogm_ctx *c, const SogmWorld *w, CloudBlocks *out);
// sparse reset (sogm_map.hip): the mark log of a pool slot as the writers see it (null entries = not logging)
struct MarkLog {
  unsigned *entries;  // [A][cap]
  unsigned *n;        // [A]
  int       cap;
  unsigned long long *stat;  // {marks written by the stamp, entries it logged} (sogm_map_traffic), or null
};
inline int cur_slot(const sogm_ctx *c) { return c->n_pool ? c->cur_idx : 0; }
MarkLog    mark_log(sogm_ctx *c, int slot);
// zero slot `slot`'s grid on `st`: the logged sectors when the slot is tracked, the dense clear otherwise
int  reset_slot(sogm_ctx *c, hipStream_t st, int slot, float *grid, bool polite);
struct PrestampDev;
// the map's part of the pre-stamp arguments (buffers allocated on first use)
int  prestamp_buffers(sogm_ctx *c, PrestampDev *d);
}  // namespace sogm

#define SOGM_HIP_CHECK(expr)                    \
  do {                                          \
    hipError_t _e = (expr);                     \
    if (_e != hipSuccess) {                     \
      sogm::set_error(#expr, _e);               \
      return SOGM_ERR_HIP;                      \
    }                                           \
  } while (0)
