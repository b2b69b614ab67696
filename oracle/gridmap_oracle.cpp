// gridmap_oracle.cpp — CPU restatement of the depth-image front end (row f1).  TEST INFRASTRUCTURE ONLY.
//
// GridMap::projectDepthImage / raycastProcess / clearAndInflateLocalMap
// (plan_env/src/grid_map.cpp:210-583) with RayCaster (plan_env/src/raycast.cpp:17-30,242-335) and the
// inline helpers of plan_env/include/plan_env/grid_map.h:261-421, followed line by line: per-pixel
// back-projection, 3-D DDA from the ray end towards the camera with the per-frame
// flag_rayend_/flag_traverse_ de-duplication, hit/miss log-odds fusion, local-map clearing and
// inflation.  Quirks kept: the zero-depth test of the filtered path reads the NEXT sampled pixel
// (:262-267); the de-duplication flags are `char`s compared with the int frame counter, so they stop
// matching after frame 127 (:371-373,391-393 with grid_map.h flag types); hit/miss counters are
// 16-bit (:110-111); inflation addresses wrap across rows (:549-556).
// Not emulated: out-of-bounds writes (camera outside the map is rejected by the callbacks, :656-662)
// and the double queue push when one voxel collects more than 65536 touches in a frame.
// Parity: UNPINNED (no reference test or fixture; no grid_map YAML ships in the tree).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <vector>

#include "oracle.h"

namespace {

struct GM {
  SogmGridMapParams P;
  int    nv[3];
  double origin[3], bmin[3], bmax[3], res, res_inv;
  double hit_log, miss_log, cmin_log, cmax_log, occ_log, unknown_flag;
  std::vector<double>  occ;
  std::vector<char>    inflate;
  std::vector<int32_t> cnt_hm, cnt_hit;  // read back as the reference's 16-bit counters at fusion time
  std::vector<signed char> flag_rayend, flag_traverse;
  int    raycast_num = 0;
  bool   has_first_depth = false, local_updated = false;
  int    lb_min[3], lb_max[3];
  std::vector<double> proj;  // xyz
  struct Queued { int a, id[3]; };
  std::queue<Queued>  cache;  // the reference queues the Vector3i id of a voxel's FIRST touch of the frame (:199-201): kept with
                              // its address — an id with a component outside the map (a ray voxel below the ground plane:
                              // z = -1) reaches, through the unchecked flat address, the top cell of the neighbouring row, and
                              // the fusion judges `in_local` on that ID (:430-433), not on the cell the address decodes to
  int N() const { return nv[0] * nv[1] * nv[2]; }
};

inline double logit(double x) { return log(x / (1 - x)); }
inline int    addr(const GM &g, int x, int y, int z) { return x * g.nv[1] * g.nv[2] + y * g.nv[2] + z; }
inline void   posToIndex(const GM &g, const double p[3], int id[3]) {
  for (int i = 0; i < 3; ++i) id[i] = (int)floor((p[i] - g.origin[i]) * g.res_inv);
}
inline void boundIndex(const GM &g, int id[3]) {
  for (int i = 0; i < 3; ++i) id[i] = std::max(std::min(id[i], g.nv[i] - 1), 0);
}
inline bool isInMap(const GM &g, const double p[3]) {
  for (int i = 0; i < 3; ++i)
    if (p[i] < g.bmin[i] + 1e-4) return false;
  for (int i = 0; i < 3; ++i)
    if (p[i] > g.bmax[i] - 1e-4) return false;
  return true;
}
// grid_map.cpp:192-208
int setCacheOccupancy(GM &g, const double p[3], int occ) {
  int id[3];
  posToIndex(g, p, id);
  const int a = addr(g, id[0], id[1], id[2]);
  if (a < 0 || a >= g.N()) return -1;  // out of bounds = UB in the reference; not emulated
  g.cnt_hm[a] += 1;
  if (g.cnt_hm[a] == 1) g.cache.push(GM::Queued{a, {id[0], id[1], id[2]}});
  if (occ == 1) g.cnt_hit[a] += 1;
  return a;
}
// grid_map.cpp:447-467
void closetPointInMap(const GM &g, const double pt[3], const double cam[3], double out[3]) {
  double diff[3], max_tc[3], min_tc[3];
  for (int i = 0; i < 3; ++i) {
    diff[i]   = pt[i] - cam[i];
    max_tc[i] = g.bmax[i] - cam[i];
    min_tc[i] = g.bmin[i] - cam[i];
  }
  double min_t = 1000000;
  for (int i = 0; i < 3; ++i) {
    if (fabs(diff[i]) > 0) {
      double t1 = max_tc[i] / diff[i];
      if (t1 > 0 && t1 < min_t) min_t = t1;
      double t2 = min_tc[i] / diff[i];
      if (t2 > 0 && t2 < min_t) min_t = t2;
    }
  }
  for (int i = 0; i < 3; ++i) out[i] = cam[i] + (min_t - 1e-3) * diff[i];
}

// raycast.cpp:17-30
int    signum(int x) { return x == 0 ? 0 : x < 0 ? -1 : 1; }
double mod(double value, double modulus) { return fmod(fmod(value, modulus) + modulus, modulus); }
double intbound(double s, double ds) {
  if (ds < 0) return intbound(-s, -ds);
  s = mod(s, 1);
  return (1 - s) / ds;
}
struct RayCaster {  // raycast.cpp:242-335
  int    x, y, z, ex, ey, ez, sx, sy, sz;
  double tMaxX, tMaxY, tMaxZ, tDX, tDY, tDZ;
  bool   setInput(const double s[3], const double e[3]) {
    x  = (int)std::floor(s[0]);
    y  = (int)std::floor(s[1]);
    z  = (int)std::floor(s[2]);
    ex = (int)std::floor(e[0]);
    ey = (int)std::floor(e[1]);
    ez = (int)std::floor(e[2]);
    const double dx = ex - x, dy = ey - y, dz = ez - z;
    sx    = signum((int)dx);
    sy    = signum((int)dy);
    sz    = signum((int)dz);
    tMaxX = intbound(s[0], dx);
    tMaxY = intbound(s[1], dy);
    tMaxZ = intbound(s[2], dz);
    tDX   = ((double)sx) / dx;
    tDY   = ((double)sy) / dy;
    tDZ   = ((double)sz) / dz;
    return !(sx == 0 && sy == 0 && sz == 0);
  }
  bool step(double out[3]) {
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

// grid_map.cpp:210-311
void projectDepthImage(GM &g, const uint16_t *img, const double cam[3], const double R[9]) {
  const SogmGridMapParams &P = g.P;
  g.proj.clear();
  const int cols = P.cols, rows = P.rows;
  auto      emit = [&](double u, double v, double depth) {
    const double c[3] = {(u - P.cx) * depth / P.fx, (v - P.cy) * depth / P.fy, depth};
    for (int i = 0; i < 3; ++i) g.proj.push_back(((R[i * 3] * c[0] + R[i * 3 + 1] * c[1]) + R[i * 3 + 2] * c[2]) + cam[i]);
  };
  if (!P.use_depth_filter) {
    for (int v = 0; v < rows; v += P.skip_pixel) {
      int k = 0;  // row_ptr++ walks consecutive pixels although u advances by skip_pixel (:225-231)
      for (int u = 0; u < cols; u += P.skip_pixel) {
        const double depth = img[(size_t)v * cols + k++] / P.k_depth_scaling_factor;
        emit(u, v, depth);
      }
    }
  } else {
    if (!g.has_first_depth) {
      g.has_first_depth = true;
    } else {
      const double inv_factor = 1.0 / P.k_depth_scaling_factor;
      const size_t total      = (size_t)rows * cols;
      for (int v = P.depth_filter_margin; v < rows - P.depth_filter_margin; v += P.skip_pixel) {
        for (int u = P.depth_filter_margin; u < cols - P.depth_filter_margin; u += P.skip_pixel) {
          double       depth = img[(size_t)v * cols + u] * inv_factor;
          const size_t nxt   = (size_t)v * cols + u + P.skip_pixel;  // the zero test reads the NEXT sample
          const uint16_t nv_ = nxt < total ? img[nxt] : 1;
          if (nv_ == 0) {
            depth = P.max_ray_length + 0.1;
          } else if (depth < P.depth_filter_mindist) {
            continue;
          } else if (depth > P.depth_filter_maxdist) {
            depth = P.max_ray_length + 0.1;
          }
          emit(u, v, depth);
        }
      }
    }
  }
}

// grid_map.cpp:313-445
void raycastProcess(GM &g, const double cam[3]) {
  const int n = (int)g.proj.size() / 3;
  if (n == 0) return;
  g.raycast_num += 1;
  const SogmGridMapParams &P = g.P;
  double mn[3] = {g.bmax[0], g.bmax[1], g.bmax[2]}, mx[3] = {g.bmin[0], g.bmin[1], g.bmin[2]};
  RayCaster rc;
  for (int i = 0; i < n; ++i) {
    double pt[3] = {g.proj[i * 3], g.proj[i * 3 + 1], g.proj[i * 3 + 2]};
    int    vox;
    auto   norm_to = [&](const double *p) {
      const double d0 = p[0] - cam[0], d1 = p[1] - cam[1], d2 = p[2] - cam[2];
      return sqrt((d0 * d0 + d1 * d1) + d2 * d2);
    };
    if (!isInMap(g, pt)) {
      double c[3];
      closetPointInMap(g, pt, cam, c);
      memcpy(pt, c, sizeof(pt));
      double length = norm_to(pt);
      if (length > P.max_ray_length)
        for (int k = 0; k < 3; ++k) pt[k] = (pt[k] - cam[k]) / length * P.max_ray_length + cam[k];
      vox = setCacheOccupancy(g, pt, 0);
    } else {
      double length = norm_to(pt);
      if (length > P.max_ray_length) {
        for (int k = 0; k < 3; ++k) pt[k] = (pt[k] - cam[k]) / length * P.max_ray_length + cam[k];
        vox = setCacheOccupancy(g, pt, 0);
      } else {
        vox = setCacheOccupancy(g, pt, 1);
      }
    }
    for (int k = 0; k < 3; ++k) {
      mx[k] = std::max(mx[k], pt[k]);
      mn[k] = std::min(mn[k], pt[k]);
    }
    if (vox != -1) {
      if ((int)g.flag_rayend[vox] == g.raycast_num) continue;
      g.flag_rayend[vox] = (signed char)g.raycast_num;
    }
    const double s[3] = {pt[0] / g.res, pt[1] / g.res, pt[2] / g.res};
    const double e[3] = {cam[0] / g.res, cam[1] / g.res, cam[2] / g.res};
    rc.setInput(s, e);
    double rp[3];
    while (rc.step(rp)) {
      const double tmp[3] = {(rp[0] + 0.5) * g.res, (rp[1] + 0.5) * g.res, (rp[2] + 0.5) * g.res};
      vox = setCacheOccupancy(g, tmp, 0);
      if (vox != -1) {
        if ((int)g.flag_traverse[vox] == g.raycast_num) break;
        g.flag_traverse[vox] = (signed char)g.raycast_num;
      }
    }
  }
  for (int k = 0; k < 3; ++k) {
    mn[k] = std::min(mn[k], cam[k]);
    mx[k] = std::max(mx[k], cam[k]);
  }
  mx[2] = std::max(mx[2], P.ground_height);
  posToIndex(g, mx, g.lb_max);
  posToIndex(g, mn, g.lb_min);
  boundIndex(g, g.lb_min);
  boundIndex(g, g.lb_max);
  g.local_updated = true;
  double lo[3], hi[3];
  for (int k = 0; k < 3; ++k) {
    lo[k] = cam[k] - P.local_update_range[k];
    hi[k] = cam[k] + P.local_update_range[k];
  }
  int min_id[3], max_id[3];
  posToIndex(g, lo, min_id);
  posToIndex(g, hi, max_id);
  boundIndex(g, min_id);
  boundIndex(g, max_id);
  while (!g.cache.empty()) {
    const GM::Queued q = g.cache.front();
    g.cache.pop();
    const int a = q.a, ix = q.id[0], iy = q.id[1], iz = q.id[2];
    const int    hm = (int16_t)g.cnt_hm[a], hit = (int16_t)g.cnt_hit[a];  // `short` counters (grid_map.h)
    const double upd = hit >= hm - hit ? g.hit_log : g.miss_log;
    g.cnt_hit[a] = g.cnt_hm[a] = 0;
    if (upd >= 0 && g.occ[a] >= g.cmax_log) {
      continue;
    } else if (upd <= 0 && g.occ[a] <= g.cmin_log) {
      g.occ[a] = g.cmin_log;
      continue;
    }
    const bool in_local = ix >= min_id[0] && ix <= max_id[0] && iy >= min_id[1] && iy <= max_id[1] &&
                          iz >= min_id[2] && iz <= max_id[2];
    if (!in_local) g.occ[a] = g.cmin_log;
    g.occ[a] = std::min(std::max(g.occ[a] + upd, g.cmin_log), g.cmax_log);
  }
}

// grid_map.cpp:469-583
void clearAndInflate(GM &g) {
  const SogmGridMapParams &P = g.P;
  const int vm = 5;
  int min_cut[3], max_cut[3], min_cut_m[3], max_cut_m[3];
  for (int k = 0; k < 3; ++k) {
    min_cut[k] = g.lb_min[k] - P.local_map_margin;
    max_cut[k] = g.lb_max[k] + P.local_map_margin;
  }
  boundIndex(g, min_cut);
  boundIndex(g, max_cut);
  for (int k = 0; k < 3; ++k) {
    min_cut_m[k] = min_cut[k] - vm;
    max_cut_m[k] = max_cut[k] + vm;
  }
  boundIndex(g, min_cut_m);
  boundIndex(g, max_cut_m);
  const double unk = g.cmin_log - g.unknown_flag;
  for (int x = min_cut_m[0]; x <= max_cut_m[0]; ++x)
    for (int y = min_cut_m[1]; y <= max_cut_m[1]; ++y) {
      for (int z = min_cut_m[2]; z < min_cut[2]; ++z) g.occ[addr(g, x, y, z)] = unk;
      for (int z = max_cut[2] + 1; z <= max_cut_m[2]; ++z) g.occ[addr(g, x, y, z)] = unk;
    }
  for (int z = min_cut_m[2]; z <= max_cut_m[2]; ++z)
    for (int x = min_cut_m[0]; x <= max_cut_m[0]; ++x) {
      for (int y = min_cut_m[1]; y < min_cut[1]; ++y) g.occ[addr(g, x, y, z)] = unk;
      for (int y = max_cut[1] + 1; y <= max_cut_m[1]; ++y) g.occ[addr(g, x, y, z)] = unk;
    }
  for (int y = min_cut_m[1]; y <= max_cut_m[1]; ++y)
    for (int z = min_cut_m[2]; z <= max_cut_m[2]; ++z) {
      for (int x = min_cut_m[0]; x < min_cut[0]; ++x) g.occ[addr(g, x, y, z)] = unk;
      for (int x = max_cut[0] + 1; x <= max_cut_m[0]; ++x) g.occ[addr(g, x, y, z)] = unk;
    }
  const int inf_step = (int)ceil(P.obstacles_inflation / g.res);
  for (int x = g.lb_min[0]; x <= g.lb_max[0]; ++x)
    for (int y = g.lb_min[1]; y <= g.lb_max[1]; ++y)
      for (int z = g.lb_min[2]; z <= g.lb_max[2]; ++z) g.inflate[addr(g, x, y, z)] = 0;
  const int N = g.N();
  for (int x = g.lb_min[0]; x <= g.lb_max[0]; ++x)
    for (int y = g.lb_min[1]; y <= g.lb_max[1]; ++y)
      for (int z = g.lb_min[2]; z <= g.lb_max[2]; ++z) {
        if (g.occ[addr(g, x, y, z)] > g.occ_log) {
          for (int a = -inf_step; a <= inf_step; ++a)
            for (int b = -inf_step; b <= inf_step; ++b)
              for (int c = -inf_step; c <= inf_step; ++c) {
                const int idx = addr(g, x + a, y + b, z + c);  // wraps across rows like the reference
                if (idx < 0 || idx >= N) continue;
                g.inflate[idx] = 1;
              }
        }
      }
  if (P.virtual_ceil_height > -0.5) {
    const int ceil_id = (int)floor((P.virtual_ceil_height - g.origin[2]) * g.res_inv) - 1;
    for (int x = g.lb_min[0]; x <= g.lb_max[0]; ++x)
      for (int y = g.lb_min[1]; y <= g.lb_max[1]; ++y) {
        const int idx = addr(g, x, y, ceil_id);
        if (idx >= 0 && idx < N) g.inflate[idx] = 1;
      }
  }
}

}  // namespace

extern "C" {

// GridMap::initMap (grid_map.cpp:15-165)
void *orc_gridmap_create(const SogmGridMapParams *P) {
  GM *g = new GM;
  g->P  = *P;
  SogmGridMapParams &p = g->P;
  if (p.virtual_ceil_height - p.ground_height > p.map_size[2]) p.virtual_ceil_height = p.ground_height + p.map_size[2];
  g->res       = p.resolution;
  g->res_inv   = 1 / p.resolution;
  g->origin[0] = -p.map_size[0] / 2.0;
  g->origin[1] = -p.map_size[1] / 2.0;
  g->origin[2] = p.ground_height;
  g->hit_log   = logit(p.p_hit);
  g->miss_log  = logit(p.p_miss);
  g->cmin_log  = logit(p.p_min);
  g->cmax_log  = logit(p.p_max);
  g->occ_log   = logit(p.p_occ);
  g->unknown_flag = 0.01;
  for (int i = 0; i < 3; ++i) {
    g->nv[i]   = (int)ceil(p.map_size[i] / p.resolution);
    g->bmin[i] = g->origin[i];
    g->bmax[i] = g->origin[i] + p.map_size[i];
  }
  const int N = g->N();
  g->occ.assign(N, g->cmin_log - g->unknown_flag);
  g->inflate.assign(N, 0);
  g->cnt_hm.assign(N, 0);
  g->cnt_hit.assign(N, 0);
  g->flag_rayend.assign(N, -1);
  g->flag_traverse.assign(N, -1);
  for (int k = 0; k < 3; ++k) {  // resetBuffer (:166-176)
    g->lb_min[k] = 0;
    g->lb_max[k] = g->nv[k] - 1;
  }
  return g;
}
void orc_gridmap_destroy(void *h) { delete (GM *)h; }
void orc_gridmap_dims(void *h, int nv[3]) {
  for (int k = 0; k < 3; ++k) nv[k] = ((GM *)h)->nv[k];
}

// depthPoseCallback + updateOccupancyCallback (:585-633,636-665): returns 1 if the map was updated
int orc_gridmap_update(void *h, const uint16_t *depth, const double cam[3], const double R[9]) {
  GM &g = *(GM *)h;
  if (!isInMap(g, cam)) return 0;
  projectDepthImage(g, depth, cam, R);
  raycastProcess(g, cam);
  const int upd = g.local_updated ? 1 : 0;
  if (g.local_updated) clearAndInflate(g);
  g.local_updated = false;
  return upd;
}
void orc_gridmap_force_frame(void *h, int raycast_num) { ((GM *)h)->raycast_num = raycast_num; }

void orc_gridmap_state(void *h, double *occ, int8_t *inflate, int bounds[6]) {
  GM &g = *(GM *)h;
  if (occ) memcpy(occ, g.occ.data(), g.occ.size() * sizeof(double));
  if (inflate) memcpy(inflate, g.inflate.data(), g.inflate.size());
  if (bounds)
    for (int k = 0; k < 3; ++k) {
      bounds[k]     = g.lb_min[k];
      bounds[3 + k] = g.lb_max[k];
    }
}
// getInflateOccupancy (grid_map.h:342-349)
int orc_gridmap_inflate_occupancy(void *h, const double pos[3]) {
  GM &g = *(GM *)h;
  if (!isInMap(g, pos)) return -1;
  int id[3];
  posToIndex(g, pos, id);
  return (int)g.inflate[addr(g, id[0], id[1], id[2])];
}
}
