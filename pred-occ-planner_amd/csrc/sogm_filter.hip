// sogm_filter.hip — MapBase::filterPointCloud (plan_env/src/map.cpp:107-132) for a batch of agents:
// PCL VoxelGrid centroid filter at filter_res, camera->body axis swap, isInRange, cap (5000).
//
// pcl::VoxelGrid<PointXYZ>::applyFilter (PCL >= 1.8, third-party, not in the reference tree) sorts the
// points by leaf index  idx = (floor(p * inv_leaf) - min_b) . (1, dx, dx*dy)  and emits one centroid
// (fp32 sum / count) per non-empty leaf in ascending idx order.  Here the leaves of the cloud's
// bounding box are a dense array of {count, sx, sy, sz} accumulators filled with atomics (HBM-bound
// scatter), and the ordered output is an ordered compaction of that array (two-level scan) — no sort.
// PCL's std::sort is unstable, so the fp32 summation order inside a leaf is unspecified in the
// reference too: centroids agree with any CPU evaluation to fp32 rounding, not bit for bit.
#include <hip/hip_runtime.h>

#include <cmath>

#include "sogm_device.hpp"

namespace sogm {

struct FilterBox {  // per agent, device
  unsigned key_min[3], key_max[3];  // order-preserving keys of the fp32 min / max
  int      min_b[3], div[3];
  int      n_cells;  // -1: more leaves than the accumulator array holds
  int      total;
};

__device__ inline unsigned f2key(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float key2f(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void k_filter_init(FilterBox *box) {
  FilterBox &b = box[blockIdx.x];
  if (threadIdx.x < 3) {
    b.key_min[threadIdx.x] = 0xffffffffu;
    b.key_max[threadIdx.x] = 0u;
  }
  if (threadIdx.x == 0) b.total = 0;
}

// getMinMax3D over the finite points
__global__ __launch_bounds__(256) void k_filter_minmax(const float *__restrict__ raw,
                                                       const int32_t *__restrict__ range, FilterBox *box) {
  const int a     = blockIdx.y;
  const int begin = range[a * 2], n = range[a * 2 + 1] - begin;
  unsigned  mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float *p = raw + (size_t)(begin + i) * 3;
    if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
    for (int k = 0; k < 3; ++k) {
      const unsigned key = f2key(p[k]);
      mn[k] = key < mn[k] ? key : mn[k];
      mx[k] = key > mx[k] ? key : mx[k];
    }
  }
  for (int k = 0; k < 3; ++k) {
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned a_ = __shfl_xor(mn[k], o), b_ = __shfl_xor(mx[k], o);
      mn[k] = a_ < mn[k] ? a_ : mn[k];
      mx[k] = b_ > mx[k] ? b_ : mx[k];
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&box[a].key_min[k], mn[k]);
      atomicMax(&box[a].key_max[k], mx[k]);
    }
  }
}

// min_b / max_b / div_b (voxel_grid.hpp: floor(min * inverse_leaf_size))
__global__ void k_filter_box(FilterBox *box, float inv_leaf, int max_cells) {
  FilterBox &b = box[blockIdx.x];
  if (threadIdx.x != 0) return;
  if (b.key_max[0] == 0u) {  // no finite point
    b.n_cells = 0;
    return;
  }
  long long cells = 1;
  for (int k = 0; k < 3; ++k) {
    const int lo = (int)floorf(key2f(b.key_min[k]) * inv_leaf);
    const int hi = (int)floorf(key2f(b.key_max[k]) * inv_leaf);
    b.min_b[k]   = lo;
    b.div[k]     = hi - lo + 1;
    cells *= (long long)b.div[k];
  }
  b.n_cells = cells > (long long)max_cells ? -1 : (int)cells;
}

// Depth images are spatially coherent: consecutive points mostly fall into the same leaf.  Each wave
// first sums runs of equal leaf index with a segmented shuffle scan and only the last lane of a run
// touches HBM, which removes most same-address atomic traffic (a leaf 1.6 m away spans ~36 pixels).
__global__ __launch_bounds__(256) void k_filter_accumulate(const float *__restrict__ raw,
                                                           const int32_t *__restrict__ range,
                                                           const FilterBox *__restrict__ box, float inv_leaf,
                                                           float *__restrict__ cells, int max_cells) {
  const int        a = blockIdx.y;
  const FilterBox &b = box[a];
  if (b.n_cells <= 0) return;
  const int begin = range[a * 2], n = range[a * 2 + 1] - begin;
  float    *c     = cells + (size_t)a * max_cells * 4;
  const int lane  = threadIdx.x & 63;
  const int n_pad = (n + 63) & ~63;  // whole waves take part in the shuffles
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_pad; i += gridDim.x * 256) {
    float x = 0.f, y = 0.f, z = 0.f, cnt = 0.f;
    int   idx = -1 - lane;  // distinct "no leaf" ids
    if (i < n) {
      const float *p = raw + (size_t)(begin + i) * 3;
      x = p[0];
      y = p[1];
      z = p[2];
      if (isfinite(x) && isfinite(y) && isfinite(z)) {
        const int ix = (int)floorf(x * inv_leaf) - b.min_b[0];
        const int iy = (int)floorf(y * inv_leaf) - b.min_b[1];
        const int iz = (int)floorf(z * inv_leaf) - b.min_b[2];
        idx = ix + iy * b.div[0] + iz * b.div[0] * b.div[1];
        cnt = 1.0f;
      }
    }
    const int                prev  = __shfl_up(idx, 1);
    const bool               head  = lane == 0 || prev != idx;
    const unsigned long long heads = __ballot(head);
    const int start = 63 - __clzll(heads & ((2ull << lane) - 1ull));  // first lane of this run
    for (int o = 1; o < 64; o <<= 1) {
      const float xs = __shfl_up(x, o), ys = __shfl_up(y, o), zs = __shfl_up(z, o), cs = __shfl_up(cnt, o);
      if (lane - o >= start) {
        x += xs;
        y += ys;
        z += zs;
        cnt += cs;
      }
    }
    const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    if (tail && idx >= 0) {
      atomicAdd(&c[(size_t)idx * 4 + 0], cnt);
      atomicAdd(&c[(size_t)idx * 4 + 1], x);
      atomicAdd(&c[(size_t)idx * 4 + 2], y);
      atomicAdd(&c[(size_t)idx * 4 + 3], z);
    }
  }
}

// centroid -> body frame (map.cpp:118-121) -> isInRange (map.h:153-157)
__device__ inline bool leaf_output(const GridGeom &g, const float *c, float out[3]) {
  if (c[0] <= 0.f) return false;
  const float cx = c[1] / c[0], cy = c[2] / c[0], cz = c[3] / c[0];
  out[0] = cz;
  out[1] = -cx;
  out[2] = -cy;
  return g.in_range(out[0], out[1], out[2]);
}

__global__ __launch_bounds__(1024) void k_filter_count(GridGeom g, const FilterBox *__restrict__ box,
                                                       const float *__restrict__ cells, int max_cells,
                                                       int *__restrict__ block_cnt, int n_blocks) {
  const int        a = blockIdx.y;
  const FilterBox &b = box[a];
  const int        i = blockIdx.x * 1024 + threadIdx.x;
  int              v = 0;
  float            o[3];
  if (i < b.n_cells) v = leaf_output(g, cells + ((size_t)a * max_cells + i) * 4, o);
  const unsigned long long m = __ballot(v);
  __shared__ int           s_w[16];
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < 16; ++w) t += s_w[w];
    block_cnt[(size_t)a * n_blocks + blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(1024) void k_filter_scan_blocks(int *__restrict__ block_cnt, int n_blocks,
                                                             FilterBox *box) {
  const int a   = blockIdx.x;
  int      *c   = block_cnt + (size_t)a * n_blocks;
  __shared__ int s_part[1024];
  const int per = (n_blocks + 1023) / 1024;
  int       sum = 0;
  for (int i = threadIdx.x * per; i < (threadIdx.x + 1) * per && i < n_blocks; ++i) sum += c[i];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < 1024; ++i) {
      const int t = s_part[i];
      s_part[i]   = run;
      run += t;
    }
    box[a].total = run;
  }
  __syncthreads();
  int run = s_part[threadIdx.x];
  for (int i = threadIdx.x * per; i < (threadIdx.x + 1) * per && i < n_blocks; ++i) {
    const int t = c[i];
    c[i]        = run;
    run += t;
  }
}

__global__ __launch_bounds__(1024) void k_filter_emit(GridGeom g, const FilterBox *__restrict__ box,
                                                      float *__restrict__ cells, int max_cells,
                                                      const int *__restrict__ block_off, int n_blocks, int cap,
                                                      float *__restrict__ out, int32_t *__restrict__ out_count) {
  const int        a = blockIdx.y;
  const FilterBox &b = box[a];
  const int        i = blockIdx.x * 1024 + threadIdx.x;
  int              v = 0;
  float            o[3] = {0.f, 0.f, 0.f};
  if (i < b.n_cells) {
    float *c = cells + ((size_t)a * max_cells + i) * 4;
    v        = leaf_output(g, c, o);
    c[0] = c[1] = c[2] = c[3] = 0.f;  // leave the accumulators clean for the next call
  }
  const unsigned long long m    = __ballot(v);
  const int                lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __shared__ int           s_w[16];
  if (lane == 0) s_w[w] = __popcll(m);
  __syncthreads();
  int base = block_off[(size_t)a * n_blocks + blockIdx.x];
  for (int k = 0; k < w; ++k) base += s_w[k];
  const int rank = base + __popcll(m & ((1ull << lane) - 1ull));
  if (v && rank < cap) {  // valid_clouds_num >= 5000 -> break (map.cpp:126-128)
    float *q = out + ((size_t)a * cap + rank) * 3;
    q[0]     = o[0];
    q[1]     = o[1];
    q[2]     = o[2];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    out_count[a] = b.n_cells < 0 ? -1 : (b.total < cap ? b.total : cap);
}

}  // namespace sogm

using namespace sogm;

extern "C" int sogm_filter_point_cloud(sogm_ctx *c, const float *raw_xyz, const int32_t *raw_range,
                                       float filter_res, int cap, float *out_xyz, int32_t *out_count,
                                       void *stream) {
  if (!c || !raw_xyz || !raw_range || !out_xyz || !out_count || !(filter_res > 0.f) || cap <= 0)
    return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  const int   A  = c->n_agents;
  if (!c->d_filter_cells) {
    const int    max_cells = c->filter_max_cells > 0 ? c->filter_max_cells : (1 << 20);
    const size_t nb        = (size_t)(max_cells + 1023) / 1024;
    SOGM_HIP_CHECK(hipMalloc((void **)&c->d_filter_cells, (size_t)A * max_cells * 4 * sizeof(float)));
    SOGM_HIP_CHECK(hipMalloc((void **)&c->d_filter_box, (size_t)A * sizeof(FilterBox)));
    SOGM_HIP_CHECK(hipMalloc((void **)&c->d_filter_blocks, (size_t)A * nb * sizeof(int)));
    SOGM_HIP_CHECK(hipMemsetAsync(c->d_filter_cells, 0, (size_t)A * max_cells * 4 * sizeof(float), st));
    c->filter_max_cells = max_cells;
  }
  const int   max_cells = c->filter_max_cells;
  const int   n_blocks  = (max_cells + 1023) / 1024;
  FilterBox  *box       = (FilterBox *)c->d_filter_box;
  const float inv_leaf  = 1.0f / filter_res;  // inverse_leaf_size_ = 1 / leaf_size_
  hipLaunchKernelGGL(k_filter_init, dim3(A), dim3(64), 0, st, box);
  hipLaunchKernelGGL(k_filter_minmax, dim3(64, A), dim3(256), 0, st, raw_xyz, raw_range, box);
  hipLaunchKernelGGL(k_filter_box, dim3(A), dim3(64), 0, st, box, inv_leaf, max_cells);
  hipLaunchKernelGGL(k_filter_accumulate, dim3(128, A), dim3(256), 0, st, raw_xyz, raw_range, box, inv_leaf,
                     c->d_filter_cells, max_cells);
  hipLaunchKernelGGL(k_filter_count, dim3(n_blocks, A), dim3(1024), 0, st, c->geom, box, c->d_filter_cells,
                     max_cells, c->d_filter_blocks, n_blocks);
  hipLaunchKernelGGL(k_filter_scan_blocks, dim3(A), dim3(1024), 0, st, c->d_filter_blocks, n_blocks, box);
  hipLaunchKernelGGL(k_filter_emit, dim3(n_blocks, A), dim3(1024), 0, st, c->geom, box, c->d_filter_cells,
                     max_cells, c->d_filter_blocks, n_blocks, cap, out_xyz, out_count);
  SOGM_HIP_CHECK(hipGetLastError());
  return SOGM_OK;
}

extern "C" int sogm_filter_reserve(sogm_ctx *c, int max_cells_per_agent) {
  if (!c || max_cells_per_agent <= 0 || c->d_filter_cells) return SOGM_ERR_INVALID_ARG;
  c->filter_max_cells = max_cells_per_agent;
  return SOGM_OK;
}
