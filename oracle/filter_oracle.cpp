// filter_oracle.cpp — CPU restatement of MapBase::filterPointCloud (row a7).  TEST INFRASTRUCTURE ONLY.
//
// plan_env/src/map.cpp:107-132: pcl::VoxelGrid<PointXYZ> at filter_res, then per output point the
// camera->body swap (x = z, y = -x, z = -y), isInRange (map.h:153-157) and the 5000-point cap.
// pcl::VoxelGrid is third-party (PCL, system package, version unpinned: plan_env/CMakeLists.txt:24) and
// absent here; its published algorithm (voxel_grid.hpp, applyFilter) is restated:
//   inverse_leaf = 1/leaf; min_b = floor(min * inverse_leaf), div_b = max_b - min_b + 1;
//   idx = (floor(p * inverse_leaf) - min_b) . (1, div_b.x, div_b.x * div_b.y);
//   sort by idx; one centroid (fp32 sum / count) per run of equal idx, in ascending idx.
// PCL's std::sort is unstable, so the order of the fp32 additions inside a leaf is unspecified; here
// points are added in input order.  Parity: UNPINNED (no reference test covers this function).
#include <algorithm>
#include <cmath>
#include <vector>

#include "oracle.h"

extern "C" int orc_filter_point_cloud(const SogmSpec *s, const float *raw, int n, float leaf, int cap,
                                      float *out) {
  const float inv = 1.0f / leaf;
  float       mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  int         n_fin = 0;
  for (int i = 0; i < n; ++i) {
    const float *p = raw + (size_t)i * 3;
    if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))) continue;
    ++n_fin;
    for (int k = 0; k < 3; ++k) {
      mn[k] = std::min(mn[k], p[k]);
      mx[k] = std::max(mx[k], p[k]);
    }
  }
  if (!n_fin) return 0;
  int       min_b[3], div[3];
  for (int k = 0; k < 3; ++k) {
    min_b[k] = (int)std::floor(mn[k] * inv);
    div[k]   = (int)std::floor(mx[k] * inv) - min_b[k] + 1;
  }
  std::vector<std::pair<long long, int>> order;
  order.reserve(n_fin);
  for (int i = 0; i < n; ++i) {
    const float *p = raw + (size_t)i * 3;
    if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))) continue;
    const long long ix = (long long)((int)std::floor(p[0] * inv) - min_b[0]);
    const long long iy = (long long)((int)std::floor(p[1] * inv) - min_b[1]);
    const long long iz = (long long)((int)std::floor(p[2] * inv) - min_b[2]);
    order.emplace_back(ix + iy * div[0] + iz * (long long)div[0] * div[1], i);
  }
  std::sort(order.begin(), order.end());
  const float rx = (float)(s->L / 2) * s->resolution, ry = (float)(s->W / 2) * s->resolution,
              rz = (float)(s->H / 2) * s->resolution;
  int    count = 0;
  size_t i     = 0;
  while (i < order.size()) {
    size_t j = i;
    float  sx = 0.f, sy = 0.f, sz = 0.f, c = 0.f;
    while (j < order.size() && order[j].first == order[i].first) {
      const float *p = raw + (size_t)order[j].second * 3;
      sx += p[0];
      sy += p[1];
      sz += p[2];
      c += 1.0f;
      ++j;
    }
    const float cx = sx / c, cy = sy / c, cz = sz / c;
    const float x = cz, y = -cx, z = -cy;  // map.cpp:118-120
    if (x > -rx && x < rx && y > -ry && y < ry && z > -rz && z < rz) {
      out[count * 3 + 0] = x;
      out[count * 3 + 1] = y;
      out[count * 3 + 2] = z;
      ++count;
      if (count >= cap) break;  // map.cpp:126-128
    }
    i = j;
  }
  return count;
}
