// sogm_map.hip — batched SOGM build + queries for gfx950 (MI355X), behind include/sogm_abi.h.
//
// Kernels (all HBM-bound integer / fp32 work — no MFMA anywhere on this path):
//   k_clear_slabs        zero sogm[A][T][V]: 16-B coalesced streaming stores, grid-stride.
//                        This IS the voxel-update roofline kernel: B = V*T*4 bytes per agent-update.
//   k_stamp_cloud        per (agent, cloud point): crop, slice-0 mark, GT-velocity lookup (cylinders
//                        staged in LDS), T-1 advected marks.   fake_particle_risk_voxel.cpp:88-161
//   k_splat_neighbours   per (agent, record, slice): Bezier sample (fp64) + body particles,
//                        float atomic add.                      risk_base.cpp:136-168,199-208
//   k_query_clear        batched getClearOcccupancy.            fake_particle_risk_voxel.cpp:309-346
//   k_obstacle_points    one workgroup per corridor box, order-preserving block-scan compaction.
//                                                               map.cpp:480-518 / risk_base.cpp:295-337
//   k_slabs_to_vt / k_vt_to_slabs   layout converters for parity I/O and futureRiskCallback.
#include <hip/hip_runtime.h>

#include <utility>
#include <vector>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "sogm_device.hpp"
#include "sogm_planner.hpp"  // FlowCtl: the pre-stamp consumes the dataflow replan's list of finished agents

namespace sogm {

static thread_local char g_err[512] = {0};
void set_error(const char *what, hipError_t e) {
  std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}
void set_error_text(const char *text) { std::snprintf(g_err, sizeof(g_err), "%s", text); }

// ------------------------------------------------------------------------------------------------
// clear
// ------------------------------------------------------------------------------------------------
// Pure streaming store.  One float4 (16 B) per lane per iteration -> 1 KiB per wave-instruction,
// fully coalesced; 4 independent stores in flight per lane per trip.  The grid is sized to
// ~8 workgroups per CU and strides over the buffer.
typedef float vfloat4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ inline void clear_store(vfloat4 *p) {
  const vfloat4 z = {0.f, 0.f, 0.f, 0.f};
  if (NT)
    __builtin_nontemporal_store(z, p);
  else
    *p = z;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_clear_slabs(vfloat4 *__restrict__ p, size_t n_vec4,
                                                     float *__restrict__ tail, int n_tail, int throttle) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t       i      = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n_vec4; i += 4 * stride) {
    clear_store<NT>(p + i);
    clear_store<NT>(p + i + stride);
    clear_store<NT>(p + i + 2 * stride);
    clear_store<NT>(p + i + 3 * stride);
    // tuning aid: bound the stores a wave keeps in flight (vmcnt <= 4 / 8 / 12)
    if (throttle == 4) __builtin_amdgcn_s_waitcnt(0x0F74);
    else if (throttle == 8) __builtin_amdgcn_s_waitcnt(0x0F78);
    else if (throttle == 12) __builtin_amdgcn_s_waitcnt(0x0F7C);
    else if (throttle == 1) __builtin_amdgcn_s_waitcnt(0x0F70);
  }
  for (; i < n_vec4; i += stride) clear_store<NT>(p + i);
  if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) tail[threadIdx.x] = 0.f;
}

// The same stream of stores with a width that follows the tick (side-stream clear of the dataflow replan).  The
// grid is cut into 4 MiB chunks handed out by an atomic cursor shared by TWO launches: a narrow one (64 workgroups,
// <= 4 stores in flight per wave: what the latency-bound planner kernels tolerate beside them) that starts with the
// tick, and a wide, unbounded one on a second stream behind k_clear_gate, which returns once *gate >= gate_target —
// every agent's corridors are final, what is left of the tick iterates in LDS (QP) — or the tick failed, or no
// chunk is left.  (Gating at launch granularity matters: workgroups that merely SLEEP on a CU hold a wave slot per
// SIMD, and a QP workgroup — 2 x 256 registers per SIMD — cannot be placed beside them.)  A workgroup asks for its
// next chunk before it stores the current one, so the cursor's round trip hides under the stores.
#define CLEAR_CHUNK_V4 (size_t)(4u << 20 >> 4)  // 16-byte elements per chunk
__global__ void k_clear_gate(const unsigned long long *__restrict__ cursor, size_t nchunks,
                             const int *__restrict__ gate, const int *__restrict__ gate_err, int gate_target,
                             const int *__restrict__ epoch_word, int epoch) {
  if (threadIdx.x != 0) return;
  // bounded like every device-side wait of the tick (0.5 s of the 100 MHz clock): opening the wide launch early is
  // harmless, a gate that never returns is not (e.g. under a profiler that serialises kernels and runs this one
  // before the narrow launch it watches)
  const long long t0 = wall_clock64();
  for (;;) {
    if (wall_clock64() - t0 > 50000000LL) break;
    if (__hip_atomic_load(cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nchunks) break;
    // the replan this clear runs under writes `epoch` after resetting its counters; a later epoch = it is over
    const int e = __hip_atomic_load(epoch_word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (e != 0 && e - epoch > 0) break;
    if (e == epoch && (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gate_target ||
                       __hip_atomic_load(gate_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
      break;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_s_sleep(127);  // ~14 us between polls
  }
}
__global__ void k_set_word(int *p, int v) { *p = v; }
// (the mark log's counters are zeroed by a kernel, not a memset node: measured with four hardware queues, a 24-byte
//  hipMemsetAsync ran AFTER work another stream had ordered behind an event recorded after it)
__global__ void k_zero_words(unsigned *p, int n) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < n) p[i] = 0u;
}

// Sparse reset: zero the 32-byte sectors named by an agent's mark log (duplicates and ~0 place-holders included; the
// sector of a logged cell holds nothing but marks of the same log or zeros).  An overflowed log (n > cap) makes the
// agent's workgroups zero its whole grid instead.  Launched (blocks, A); the counts are reset by a kernel behind it.
// LANES adjacent lanes zero one entry with one 16-byte store each, so an entry is ONE write request of 16*LANES bytes
// at the L2 instead of two of 16; LANES = 4 zeroes the aligned 64-byte pair of sectors (everything
// non-zero in a tracked grid is in the log, so the neighbour sector holds marks of the same log or zeros as well).  An
// entry equal to the one before it in the wave is skipped: neighbouring marks of a stamp row log the same sector.
// Measured alone on 80.6 M entries (cfg2, 128 agents): one lane per entry with two stores 1.10 ms; 2 lanes 0.91;
// 4 lanes 0.87; 4 lanes x 8 entries per trip 0.745; 8 lanes (128-byte lines) 1.08-1.2.  reset_slot picks per use.
template <int LANES, int UNROLL>
__global__ __launch_bounds__(256) void k_reset_sectors(char *__restrict__ grid, size_t agent_bytes,
                                                       const unsigned *__restrict__ entries,
                                                       const unsigned *__restrict__ counts, int cap,
                                                       unsigned long long *__restrict__ stat) {
  const int      agent = blockIdx.y;
  const unsigned n     = counts[agent];
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // statistics for sogm_sparse_reset_state: entries read, launches
    atomicAdd(stat, (unsigned long long)(n > (unsigned)cap ? (unsigned)cap : n));
    if (agent == 0) atomicAdd(stat + 1, 1ull);
  }
  char          *base  = grid + (size_t)agent * agent_bytes;
  const vfloat4  z     = {0.f, 0.f, 0.f, 0.f};
  const size_t   tid   = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
  if (n > (unsigned)cap) {
    // dense fall-back for this agent: 16-byte stores over the aligned body, bytes at the two ends
    char  *lo = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(base) + 15) & ~(uintptr_t)15);
    char  *hi = reinterpret_cast<char *>(reinterpret_cast<uintptr_t>(base + agent_bytes) & ~(uintptr_t)15);
    if (hi < lo) hi = lo = base + agent_bytes;
    const size_t nv = (size_t)(hi - lo) / 16;
    for (size_t i = tid; i < nv; i += nthr) __builtin_nontemporal_store(z, reinterpret_cast<vfloat4 *>(lo) + i);
    if (tid == 0) {
      for (char *q = base; q < lo && q < base + agent_bytes; ++q) *q = 0;
      for (char *q = hi; q < base + agent_bytes; ++q) *q = 0;
    }
    return;
  }
  const unsigned *e       = entries + (size_t)agent * cap;
  unsigned        n_lines = 0;  // lines this lane group zeroed (counted on the group's first lane)
  const int       part    = (int)(threadIdx.x % LANES);
  const bool      first   = (threadIdx.x & 63) < LANES;  // the wave's first entry has no predecessor to compare with
  const bool      aligned = (reinterpret_cast<uintptr_t>(base) & (16 * LANES - 1)) == 0;
  // UNROLL entries per trip, their loads issued together (a trip is otherwise one dependent load -> store pair)
  const size_t  stride = nthr / LANES;
  for (size_t i0 = tid / LANES; i0 < n; i0 += UNROLL * stride) {
    unsigned sct[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) sct[u] = i0 + u * stride < n ? e[i0 + u * stride] : 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const unsigned id   = LANES == 4 ? sct[u] >> 1 : sct[u];  // the 16*LANES-byte line this entry zeroes
      const unsigned prev = __shfl_up(id, LANES);               // (lanes below an active lane are active: smaller i0)
      if (sct[u] == 0xFFFFFFFFu || (!first && prev == id)) continue;
      if (part == 0) ++n_lines;
      const size_t off = (size_t)id * (16 * LANES) + 16 * part;
      if (aligned) {
        if (off + 16 <= agent_bytes) *reinterpret_cast<vfloat4 *>(base + off) = z;
        else if (off < agent_bytes)  // a short last sector
          for (size_t b = off; b + 2 <= agent_bytes; b += 2) *reinterpret_cast<unsigned short *>(base + b) = 0;
      } else if (part == 0) {  // odd grid sizes: the agent's base is only cell-aligned
        const size_t o32 = (size_t)sct[u] * 32;
        const size_t end = o32 + 32 <= agent_bytes ? o32 + 32 : agent_bytes;
        for (size_t b = o32; b + 2 <= end; b += 2) *reinterpret_cast<unsigned short *>(base + b) = 0;
      }
    }
  }
  // statistics: the 16 * LANES-byte lines zeroed (what the launch wrote), one atomic per wave
  for (int d = 32; d >= 1; d >>= 1) n_lines += (unsigned)__shfl_xor((int)n_lines, d, 64);
  if ((threadIdx.x & 63) == 0 && n_lines) atomicAdd(stat + 2, (unsigned long long)n_lines * (unsigned)(16 * LANES));
}
template <bool POLITE>
__global__ __launch_bounds__(256) void k_clear_chunks(vfloat4 *__restrict__ p, size_t n_vec4,
                                                      float *__restrict__ tail, int n_tail,
                                                      unsigned long long *__restrict__ cursor,
                                                      unsigned long long *__restrict__ next_cursor,
                                                      const int *__restrict__ epoch_word, int epoch, int bound) {
  __shared__ unsigned long long s_next;
  // two cursors take turns: the narrow launch of a clear zeroes the one the NEXT clear will use (the clear that used
  // it last is complete — both of its launches are ordered before this one on the side stream); no memset node
  if (POLITE && blockIdx.x == 0 && threadIdx.x == 0) *next_cursor = 0ull;
  const size_t nchunks = (n_vec4 + CLEAR_CHUNK_V4 - 1) / CLEAR_CHUNK_V4;
  // the wide launch only streams while the replan it was opened for is in flight (*epoch_word == epoch): once the
  // next update starts (word reset) its workgroups take no further chunk and the narrow launch finishes alone
  auto take = [&]() -> unsigned long long {
    if (!POLITE && __hip_atomic_load(epoch_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) return ~0ull;
    return atomicAdd(cursor, 1ull);
  };
  if (threadIdx.x == 0) s_next = take();
  __syncthreads();
  unsigned long long cur = s_next;
  while (cur < nchunks) {
    __syncthreads();  // everybody holds `cur`
    unsigned long long nxt = 0;
    if (threadIdx.x == 0) nxt = take();  // in flight under the stores below
    const size_t b = (size_t)cur * CLEAR_CHUNK_V4;
    const size_t e = b + CLEAR_CHUNK_V4 < n_vec4 ? b + CLEAR_CHUNK_V4 : n_vec4;
    size_t       i = b + threadIdx.x;
    for (; i + 768 < e; i += 1024) {
      clear_store<true>(p + i);
      clear_store<true>(p + i + 256);
      clear_store<true>(p + i + 512);
      clear_store<true>(p + i + 768);
      if (POLITE) __builtin_amdgcn_s_waitcnt(0x0F75);  // vmcnt <= 5: four stores (+ the cursor's atomic on lane 0)
      else if (bound == 8) __builtin_amdgcn_s_waitcnt(0x0F78);
      else if (bound == 12) __builtin_amdgcn_s_waitcnt(0x0F7C);
      else if (bound == 16) __builtin_amdgcn_s_waitcnt(0x4F70);
      else if (bound == 24) __builtin_amdgcn_s_waitcnt(0x4F78);
      else if (bound == 32) __builtin_amdgcn_s_waitcnt(0x8F70);
    }
    for (; i < e; i += 256) clear_store<true>(p + i);
    if (threadIdx.x == 0) s_next = nxt;
    __syncthreads();
    cur = s_next;
  }
  if (POLITE && blockIdx.x == 0 && (int)threadIdx.x < n_tail) tail[threadIdx.x] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// stamp: cloud -> slice 0, GT velocity -> slices 1..T-1
// ------------------------------------------------------------------------------------------------
#define SOGM_MAX_CYL_LDS 1024

// Ring obstacle test (fake_particle_risk_voxel.cpp:137-149): the voxel corner pt lies within 2 voxels of the
// ring's plane and of its circle of diameter w.  Eigen's operation sequence in fp32 (as oracle/map_oracle.cpp):
// q * v = _transformVector (uv = q.vec x v; uv += uv; v + w uv + q.vec x uv), Hyperplane::Through(p0, p1, p2)
// (normal = (p2 - p0) x (p1 - p0), normalised; offset = -p0.normal), projection, absDistance.
__device__ inline void cross3f(const float a[3], const float b[3], float o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ inline void quat_rotate_f(float qw, const float qv[3], const float v[3], float o[3]) {
  float uv[3], w2[3];
  cross3f(qv, v, uv);
  for (int k = 0; k < 3; ++k) uv[k] += uv[k];
  cross3f(qv, uv, w2);
  for (int k = 0; k < 3; ++k) o[k] = (v[k] + qw * uv[k]) + w2[k];
}
__device__ __noinline__ bool ring_contains(const SogmCylinder &cy, float px, float py, float pz, float res) {
  const float qw = (float)cy.qw, qv[3] = {(float)cy.qx, (float)cy.qy, (float)cy.qz};
  const float c0[3] = {(float)cy.x, (float)cy.y, (float)cy.z};
  const float ey[3] = {0, 1, 0}, ex[3] = {1, 0, 0};
  float       ry[3], rx[3], v0[3], v1[3], n[3];
  quat_rotate_f(qw, qv, ey, ry);
  quat_rotate_f(qw, qv, ex, rx);
  for (int k = 0; k < 3; ++k) {
    const float p1 = c0[k] + ry[k], p2 = c0[k] + rx[k];
    v0[k] = p2 - c0[k];
    v1[k] = p1 - c0[k];
  }
  cross3f(v0, v1, n);
  const float nn = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
  for (int k = 0; k < 3; ++k) n[k] /= nn;
  const float off = -((c0[0] * n[0] + c0[1] * n[1]) + c0[2] * n[2]);
  const float sd  = ((n[0] * px + n[1] * py) + n[2] * pz) + off;
  const float b0 = px - sd * n[0], b1 = py - sd * n[1], b2 = pz - sd * n[2];
  const float e0 = c0[0] - b0, e1 = c0[1] - b1, e2 = c0[2] - b2;
  const float dist = sqrtf((e0 * e0 + e1 * e1) + e2 * e2);
  return fabs(cy.w / 2 - (double)dist) < (double)(2 * res) && fabsf(sd) < 2 * res;
}

// Candidate cylinders of an agent: those that can touch its window, in their original order (the reference takes
// the FIRST record containing the voxel, :127-154).  A voxel corner lies inside the window, so a record further than
// range + res + reach from the map centre along x or y can never match; culling keeps the per-point loop independent
// of the size of the global obstacle field (tens of candidates instead of thousands).  One wave per agent, no LDS.
struct CylCand {
  float  x, y, vx, vy;
  double wlim;  // w + clearance (fp64, as the reference compares)
  int    type, orig;
};
static_assert(sizeof(CylCand) == 32, "CylCand is copied in 16-byte pieces");
// candidates a pre-stamp wave keeps in LDS (16 KiB; the bench scene keeps ~280 of its 555 cylinders per agent, the rest
// of a longer list is read from global memory as before)
constexpr int PRESTAMP_CAND_LDS = 512;
// (The kernel also files the update's poses and stamps into the context's arrays — the later kernels of the update
//  and the queries read those —, which saves two copy nodes in front of it.)
__device__ __forceinline__ void cull_agent(const GridGeom &g, const SogmCylinder *__restrict__ cyl, int n_cyl,
                                           float q0, float q1, CylCand *__restrict__ out, int *__restrict__ n_out,
                                           int lane) {
  int kept = 0;
  for (int c0 = 0; c0 < n_cyl; c0 += 64) {
    const int c    = c0 + lane;
    bool      keep = false;
    double    wl   = 0.0;
    if (c < n_cyl) {
      wl = cyl[c].w + (double)g.clearance;
      // reach of a record around its centre: a cylinder matches within w + clearance; a ring (type 2) within
      // w/2 + 2 res of its axis point and 2 res of its plane, i.e. at most w/2 + 4 res from the centre
      const double reach = cyl[c].type == 2 ? 0.5 * cyl[c].w + 4.0 * (double)g.res : wl;
      const double lim_x = (double)g.rx + (double)g.res + reach + 0.5, lim_y = (double)g.ry + (double)g.res + reach + 0.5;
      keep = fabs((double)(float)cyl[c].x - (double)q0) <= lim_x && fabs((double)(float)cyl[c].y - (double)q1) <= lim_y;
    }
    const unsigned long long m   = __ballot(keep);
    const int                off = kept + __popcll(m & ((1ull << lane) - 1ull));
    if (keep && off < SOGM_MAX_CYL_LDS)
      out[off] = CylCand{(float)cyl[c].x, (float)cyl[c].y, (float)cyl[c].vx, (float)cyl[c].vy, wl, cyl[c].type, c};
    kept += __popcll(m);
  }
  if (lane == 0) *n_out = kept;  // > SOGM_MAX_CYL_LDS: the stamp falls back to the full list
}
__device__ __forceinline__ void cull_blocks_agent(const GridGeom &g, const CloudBlocks &cb, int agent, float q0, float q1,
                                                  int lane);
__global__ __launch_bounds__(64) void k_cull_cylinders(GridGeom g, const SogmCylinder *__restrict__ cyl, int n_cyl,
                                                       const float *__restrict__ poses, CylCand *__restrict__ cand,
                                                       int *__restrict__ n_cand, const double *__restrict__ stamps_in,
                                                       float *__restrict__ poses_out, double *__restrict__ stamps_out,
                                                       CloudBlocks cb, long long *__restrict__ tick_clock) {
  const int    agent = blockIdx.x, lane = threadIdx.x;
  if (tick_clock && agent == 0 && lane == 0) tick_clock[0] = wall_clock64();  // the update's first kernel is running
  const float  q0 = poses[agent * 3], q1 = poses[agent * 3 + 1];
  if (lane < 3) poses_out[agent * 3 + lane] = poses[agent * 3 + lane];
  if (lane == 3) stamps_out[agent] = stamps_in[agent];
  cull_agent(g, cyl, n_cyl, q0, q1, cand + (size_t)agent * SOGM_MAX_CYL_LDS, n_cand + agent, lane);
  if (cb.bounds) cull_blocks_agent(g, cb, agent, q0, q1, lane);  // the agent's crop of a SogmWorld cloud
}

// The stamp in two passes (one-wave workgroups, no LDS; the candidates come from k_cull_cylinders through L1 / the
// scalar cache).  The cloud arrives z-fastest (an obstacle generator walks x, y, z), i.e. consecutive points are a
// whole z-layer apart in the grid, and the T - 1 future marks of a voxel sit a slice (V cells) apart: stamped point
// by point, every 4-byte mark dirtied a sector of its own (rocprof WRITE_SIZE 8.8 x the marked bytes, r02).  Now
//   k_stamp_bits   per (agent, cloud point): crop, voxel index, ONE fire-and-forget atomic OR into the agent's
//                  occupancy bitmask of slice 0 (V bits: on-chip traffic; duplicates — the 0.10 m cloud lattice is
//                  finer than the 0.15 m voxels — cost nothing more);
//   k_stamp_marks  per (agent, 64 mask words): lanes take the set bits of a non-zero word — 32 x-consecutive
//                  voxels — so the slice-0 mark and the T - 1 future marks of neighbouring voxels leave the wave as
//                  stores to neighbouring addresses of one slice (a few 64-byte lines per instruction instead of one
//                  line per mark); the velocity lookup runs once per occupied voxel as before; consumed words are
//                  zeroed for the next update.
// The set of marked cells is the one the per-point form produced (marks are idempotent, the lookup depends on the
// voxel only).
// one cloud point of the bits pass: PassThrough limits (fake_particle_risk_voxel.cpp:88-104, fp32), voxel index, one OR
struct CropBox {
  float lox, hix, loy, hiy, loz, hiz, p0, p1, p2;
  __device__ __forceinline__ CropBox(const GridGeom &g, float q0, float q1, float q2)
      : lox(q0 - g.rx), hix(q0 + g.rx), loy(q1 - g.ry), hiy(q1 + g.ry), loz(q2 - g.rz), hiz(q2 + g.rz), p0(q0), p1(q1), p2(q2) {}
};
// the bit a cloud point sets (the slice's storage order), -1 if the point sets none
__device__ __forceinline__ int stamp_bit_of(const GridGeom &g, const CropBox &b, float px, float py, float pz) {
  if (!(px >= b.lox && px <= b.hix && py >= b.loy && py <= b.hiy && pz >= b.loz && pz <= b.hiz)) return -1;
  const float x = px - b.p0, y = py - b.p1, z = pz - b.p2;
  if (!g.in_range(x, y, z)) return -1;
  const int v = g.voxel_of(x, y, z);
  if (v >= g.V) return -1;  // (the reference's out-of-array index, see stamp_bits_point)
  return g.phys_of(v);
}
__device__ __forceinline__ void stamp_bits_point(const GridGeom &g, const CropBox &b, float px, float py, float pz,
                                                 unsigned *__restrict__ mask) {
  if (!(px >= b.lox && px <= b.hix && py >= b.loy && py <= b.hiy && pz >= b.loz && pz <= b.hiz)) return;
  const float x = px - b.p0, y = py - b.p1, z = pz - b.p2;
  if (!g.in_range(x, y, z)) return;
  const int v = g.voxel_of(x, y, z);
  // Reference UB (map.h:169-174): a coordinate one ulp below +range rounds up to range in "x + r" and to the full
  // count in the fp32 division, so the index component equals the axis size; for z (or y on the top layer) the
  // voxel index is >= V and the reference writes outside risk_maps_.  Such marks are dropped, here and in the
  // oracle (x / y overflows inside the array wrap into the next row / layer exactly as the reference's do).
  if (v >= g.V) return;
  const int p = g.phys_of(v);  // the mask is kept in the slice's storage order: neighbouring bits share a sector
  __hip_atomic_fetch_or(mask + (p >> 5), 1u << (p & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// points first, first + stride, ... of [begin, end) (lane included in `first`)
__device__ __forceinline__ void stamp_bits_range(const GridGeom &g, const float *__restrict__ cloud, int first, int end,
                                                 int stride, float p0, float p1, float p2, unsigned *__restrict__ mask) {
  const CropBox box(g, p0, p1, p2);
  // four points per trip: their twelve loads are in flight together (one point per trip was a dependent HBM / L2 round
  // trip each, ~2.3 us per point and lane inside the tick; the marks are idempotent ORs, their order is free)
  int i = first;
  for (; (long long)i + 3ll * stride < end; i += 4 * stride) {
    float px[4], py[4], pz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t q = (size_t)(i + u * stride) * 3;
      px[u] = cloud[q];
      py[u] = cloud[q + 1];
      pz[u] = cloud[q + 2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) stamp_bits_point(g, box, px[u], py[u], pz[u], mask);
  }
  for (; i < end; i += stride) stamp_bits_point(g, box, cloud[(size_t)i * 3], cloud[(size_t)i * 3 + 1], cloud[(size_t)i * 3 + 2], mask);
}

// ---- the crop of updateMap on the device (SogmWorld): the cloud comes in blocks of `block_points` consecutive points
// with their xy bounds (sogm_cloud_block_bounds); an agent's crop is the ascending list of the blocks whose bounds
// intersect its window, built once per agent-update by one wave (cull_blocks_agent), and the bits pass walks that list.
// The PassThrough test per point is unchanged, so the set of marked voxels is the one a scan of the whole cloud gives.
__device__ __forceinline__ void cull_blocks_agent(const GridGeom &g, const CloudBlocks &cb, int agent, float q0, float q1,
                                                  int lane) {
  const float lox = q0 - g.rx, hix = q0 + g.rx, loy = q1 - g.ry, hiy = q1 + g.ry;
  int        *out = cb.list + (size_t)agent * cb.row;
  int         kept = 0;
  for (int b0 = 0; b0 < cb.n_blocks; b0 += 64) {
    const int b    = b0 + lane;
    bool      keep = false;
    if (b < cb.n_blocks) {
      const float4 bb = reinterpret_cast<const float4 *>(cb.bounds)[b];  // {xmin, xmax, ymin, ymax}
      keep            = !(bb.y < lox || bb.x > hix || bb.w < loy || bb.z > hiy);
    }
    const unsigned long long m = __ballot(keep);
    if (keep) out[kept + __popcll(m & ((1ull << lane) - 1ull))] = b;
    kept += __popcll(m);
  }
  if (lane == 0) cb.n_list[agent] = kept;
}
// 64 points of a wave: ONE atomic OR per distinct mask word.  The ORs are device-scope atomics that execute at the memory
// side — their count is what the bits pass costs (a predecessor-lane de-duplication alone: 448 -> 318 us per tick) — and
// the points of a block are neighbours (consecutive along z, 0.10 m apart in 0.15 m voxels, columns side by side), so a
// wave's 64 points fall into a handful of words (a word = four 2 x 2 x 2 tiles along x).  Leader loop: the first lane
// still to do names its word, the lanes with that word OR their bits together, the leader issues the atomic.
__device__ __forceinline__ void stamp_bits_wave(const GridGeom &g, const CropBox &box, float px, float py, float pz,
                                                unsigned *__restrict__ mask, int lane) {
  const int          p   = stamp_bit_of(g, box, px, py, pz);
  const int          w   = p >= 0 ? p >> 5 : -1;
  const unsigned     bit = p >= 0 ? 1u << (p & 31) : 0u;
  unsigned long long todo = __ballot(p >= 0);
  while (todo) {  // uniform
    const int                leader = __builtin_ctzll(todo);
    const int                wv     = __shfl(w, leader, 64);
    const bool               mine   = w == wv;
    const unsigned long long m      = __ballot(mine);
    unsigned                 b      = mine ? bit : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) b |= (unsigned)__shfl_xor((int)b, d, 64);
    if (lane == leader) __hip_atomic_fetch_or(mask + wv, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    todo &= ~m;
  }
}
// ... and for the 256 points a wave has in flight (four per lane): the columns of a block lie side by side, 0.10 m apart,
// so their bits share words across the four sets as well
// (called by ALL 64 lanes — the wave OR reads lane 63 —; `valid` bit u = the lane's point u exists)
__device__ __forceinline__ void stamp_bits_wave4(const GridGeom &g, const CropBox &box, const float (&px)[4],
                                                 const float (&py)[4], const float (&pz)[4], unsigned *__restrict__ mask,
                                                 int lane, unsigned valid) {
  int                w[4];
  unsigned           bit[4];
  unsigned long long todo[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int p = ((valid >> u) & 1u) ? stamp_bit_of(g, box, px[u], py[u], pz[u]) : -1;
    w[u]        = p >= 0 ? p >> 5 : -1;
    bit[u]      = p >= 0 ? 1u << (p & 31) : 0u;
    todo[u]     = __ballot(p >= 0);
  }
  for (;;) {  // uniform
    int wv;
    if (todo[0]) wv = __shfl(w[0], __builtin_ctzll(todo[0]), 64);
    else if (todo[1]) wv = __shfl(w[1], __builtin_ctzll(todo[1]), 64);
    else if (todo[2]) wv = __shfl(w[2], __builtin_ctzll(todo[2]), 64);
    else if (todo[3]) wv = __shfl(w[3], __builtin_ctzll(todo[3]), 64);
    else break;
    unsigned b = 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool mine = w[u] == wv;
      b |= mine ? bit[u] : 0u;
      todo[u] &= ~__ballot(mine);
    }
    // OR over the wave with DPP (row_shr 1, 2, 4, 8: lane 15 of every row holds its row; row_bcast 15 / 31: lane 63 holds
    // the wave) — seven instructions instead of six ds_bpermute round trips with their address arithmetic: the bits pass
    // is bound by instruction issue (80 % of the slots), and this loop is most of it
    int x = (int)b;
    x |= __builtin_amdgcn_update_dpp(0, x, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x112 /* row_shr:2 */, 0xF, 0xF, true);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x118 /* row_shr:8 */, 0xF, 0xF, true);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x142 /* row_bcast:15 */, 0xA, 0xF, true);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x143 /* row_bcast:31 */, 0xC, 0xF, true);
    b = (unsigned)__builtin_amdgcn_readlane(x, 63);
    if (lane == 0) __hip_atomic_fetch_or(mask + wv, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// listed blocks first, first + stride, ... of the agent (one wave per call; a block's points are taken 64 at a time)
// WAVE_DEDUPE: the word-level de-duplication over the 256 points in flight (k_stamp_bits_blocks: 32768 short-lived waves
// hide its ~10 us of dependent cross-lane steps per block, and the atomics' count is that kernel's cost); the persistent
// kernels, where a ticket walks dozens of blocks one after the other, only drop the bit of the predecessor lane (one DPP
// move: with the leader loop the flight's bits stage took 775 instead of 146 wave-ms per tick)
template <bool WAVE_DEDUPE = false>
__device__ __forceinline__ void stamp_bits_blocks(const GridGeom &g, const float *__restrict__ cloud, const CloudBlocks &cb,
                                                  int agent, int first, int stride, float p0, float p1, float p2,
                                                  unsigned *__restrict__ mask, int lane) {
  const CropBox box(g, p0, p1, p2);
  const int    *list = cb.list + (size_t)agent * cb.row;
  const int     n    = cb.n_list[agent];
  for (int i = first; i < n; i += stride) {  // uniform
    const int b   = list[i];
    const int beg = b * cb.block_points;
    const int end = beg + cb.block_points < cb.n_points ? beg + cb.block_points : cb.n_points;
    if constexpr (WAVE_DEDUPE) {
      for (int jb = beg; jb < end; jb += 256) {  // uniform trips of 256 points: every lane takes part in the wave OR
        float    px[4], py[4], pz[4];
        unsigned valid = 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = jb + lane + u * 64;
          valid |= (idx < end ? 1u : 0u) << u;
          const size_t q = (size_t)(idx < end ? idx : end - 1) * 3;
          px[u] = cloud[q];
          py[u] = cloud[q + 1];
          pz[u] = cloud[q + 2];
        }
        stamp_bits_wave4(g, box, px, py, pz, mask, lane, valid);
      }
      continue;
    }
    int       j   = beg + lane;
    for (; j + 192 < end; j += 256) {  // four points of this lane in flight
      float px[4], py[4], pz[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t q = (size_t)(j + u * 64) * 3;
        px[u] = cloud[q];
        py[u] = cloud[q + 1];
        pz[u] = cloud[q + 2];
      }
      // consecutive points of a block are neighbours along z, 0.10 m apart in 0.15 m voxels: a lane whose bit is its
      // predecessor's leaves the OR to it (the ORs are device-scope atomics that execute at the memory side: their count
      // is what the pass costs)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p    = stamp_bit_of(g, box, px[u], py[u], pz[u]);
        const int prev = __builtin_amdgcn_update_dpp(-2, p, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
        if (p >= 0 && prev != p)
          __hip_atomic_fetch_or(mask + (p >> 5), 1u << (p & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    for (; j < end; j += 64) stamp_bits_point(g, box, cloud[(size_t)j * 3], cloud[(size_t)j * 3 + 1], cloud[(size_t)j * 3 + 2], mask);
  }
}
__global__ __launch_bounds__(64) void k_stamp_bits(GridGeom g, const float *__restrict__ cloud,
                                                   const int32_t *__restrict__ cloud_range,
                                                   const float *__restrict__ poses, unsigned *__restrict__ bits,
                                                   int words_per_agent, int agent0) {
  const int    agent = blockIdx.y + agent0;
  const int    begin = cloud_range[agent * 2], end = cloud_range[agent * 2 + 1];
  const float *pose  = poses + agent * 3;
  stamp_bits_range(g, cloud, begin + (int)(blockIdx.x * blockDim.x + threadIdx.x), end, (int)(gridDim.x * blockDim.x),
                   pose[0], pose[1], pose[2], bits + (size_t)agent * words_per_agent);
}

__global__ __launch_bounds__(64) void k_stamp_bits_blocks(GridGeom g, const float *__restrict__ cloud, CloudBlocks cb,
                                                          const float *__restrict__ poses, unsigned *__restrict__ bits,
                                                          int words_per_agent) {
  const int    agent = blockIdx.y;
  const float *pose  = poses + agent * 3;
  stamp_bits_blocks<true>(g, cloud, cb, agent, (int)blockIdx.x, (int)gridDim.x, pose[0], pose[1], pose[2],
                          bits + (size_t)agent * words_per_agent, (int)threadIdx.x);
}
// xy bounds of every block of `block_points` consecutive points (sogm_cloud_block_bounds): one wave per block
__global__ __launch_bounds__(64) void k_block_bounds(const float *__restrict__ cloud, int n_points, int block_points,
                                                     float *__restrict__ out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int beg = b * block_points, end = beg + block_points < n_points ? beg + block_points : n_points;
  float xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
  for (int j = beg + lane; j < end; j += 64) {
    const float x = cloud[(size_t)j * 3], y = cloud[(size_t)j * 3 + 1];
    xlo = fminf(xlo, x);
    xhi = fmaxf(xhi, x);
    ylo = fminf(ylo, y);
    yhi = fmaxf(yhi, y);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    xlo = fminf(xlo, __shfl_xor(xlo, d, 64));
    xhi = fmaxf(xhi, __shfl_xor(xhi, d, 64));
    ylo = fminf(ylo, __shfl_xor(ylo, d, 64));
    yhi = fmaxf(yhi, __shfl_xor(yhi, d, 64));
  }
  if (lane == 0) reinterpret_cast<float4 *>(out)[b] = make_float4(xlo, xhi, ylo, yhi);
}

// Profiling build only (make EXTRA=-DSOGM_PROFILE_PRESTAMP, tools/diag_prestamp.py): 100 MHz ticks the pre-stamp's waves
// spend [0] waiting for a published agent, [1] waiting for the agent's cull / bits pass, [2] in the cull, [3] in the bits
// pass, [4] in the marks pass, of which [5] the candidate walk and [6] the slice loops; [7] tickets, [8] chunks of 64 voxels
#ifdef SOGM_PROFILE_PRESTAMP
__device__ unsigned long long g_ps_prof[12];
#define PS_CLK() wall_clock64()
#define PS_ADD(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_ps_prof[i], (unsigned long long)(v)); } while (0)
#else
#define PS_CLK() 0ll
#define PS_ADD(i, v) do { } while (0)
#endif
// CACHED: the slice loop keeps a group's sectors in registers between its store pass and its log pass (fewer
// instructions: the persistent kernels, whose occupancy is two waves per SIMD anyway); otherwise it recomputes them in a
// second pass (fewer registers: k_stamp_marks, whose scattered stores want every wave the CU can hold — 1.61 ms against
// 1.76 for the cached form on the full machine).
template <bool CACHED>
__device__ __forceinline__ void stamp_marks_trips(const GridGeom &g, void *__restrict__ grid,
                                                  unsigned *__restrict__ bits, int words_per_agent,
                                                  const SogmCylinder *__restrict__ cyl, int n_cyl,
                                                  const float *__restrict__ poses,
                                                  const CylCand *__restrict__ cand_all,
                                                  const int *__restrict__ n_cand, int agent, const MarkLog &lg,
                                                  int w_first, int w_stride, CylCand *cand_lds = nullptr,
                                                  int cand_lds_cap = 0, unsigned *lds_secs = nullptr) {
  const int      kept   = n_cand[agent];
  const bool     culled = kept <= SOGM_MAX_CYL_LDS;
  const int      n_lds  = culled ? kept : 0;
  const int      n_loop = culled ? kept : n_cyl;
  const CylCand *cand   = cand_all + (size_t)agent * SOGM_MAX_CYL_LDS;
  const float   *pose   = poses + agent * 3;
  const float    p0 = pose[0], p1 = pose[1], p2 = pose[2];
  char          *base = reinterpret_cast<char *>(grid) + (size_t)agent * g.T * (size_t)g.V * (g.half ? 2 : 4);
  unsigned      *mask = bits + (size_t)agent * words_per_agent;
  const int      lane = threadIdx.x;
  unsigned n_marks = 0, n_logged = 0;  // this lane's marks, the wave's log entries (statistics: two atomics per call)
  // One-wave workgroups of the dataflow pre-stamp stage the head of the agent's candidate list in LDS at their first
  // occupied trip (16-byte pieces, coalesced): the walk below reads every candidate for every chunk of 64 voxels, and
  // from global memory each batch of four cost a ~0.7 us L2 round trip in the tick — 51 us per chunk, three quarters of
  // the pre-stamp's wave time (tools/diag_prestamp.py) and what the tick's end waited for in half of the ticks.
  int n_staged = 0;
  bool staged  = cand_lds == nullptr;
  // a trip covers 256 mask words (8192 voxels): every lane loads four, the set bits of the whole trip are numbered
  // by a wave scan of the pop counts, and lane t of chunk b takes set bit b + t — dense lanes whatever the
  // occupancy pattern, and neighbouring lanes still hold neighbouring voxels (words_per_agent is padded to 256)
  for (int w0 = w_first; w0 < words_per_agent; w0 += w_stride) {
    uint4 *wp = reinterpret_cast<uint4 *>(mask + w0) + lane;
    uint4  w4 = *wp;
    if ((w4.x | w4.y | w4.z | w4.w) != 0u) *wp = make_uint4(0u, 0u, 0u, 0u);  // consumed: clean for the next update
    const int c0 = __popc(w4.x), c1 = c0 + __popc(w4.y), c2 = c1 + __popc(w4.z), cnt = c2 + __popc(w4.w);
    int       inc = cnt;  // inclusive prefix sum over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    const int excl  = inc - cnt;
    const int total = __shfl(inc, 63, 64);
    if (total > 0 && !staged) {  // uniform
      n_staged            = n_lds < cand_lds_cap ? n_lds : cand_lds_cap;
      const uint4 *src    = reinterpret_cast<const uint4 *>(cand);
      uint4       *dst    = reinterpret_cast<uint4 *>(cand_lds);
      const int    pieces = n_staged * (int)(sizeof(CylCand) / 16);
      for (int q = lane; q < pieces; q += 64) dst[q] = src[q];
      __syncthreads();  // (one wave: the LDS writes above are visible to the reads of the walk)
      staged = true;
    }
    for (int b0 = 0; b0 < total; b0 += 64) {  // uniform
      const int  i      = b0 + lane;
      const bool active = i < total;
      // owner lane: the last one whose exclusive prefix is <= i (binary search with lane reads; lanes with no set
      // bit share their successor's prefix and lose the comparison)
      int lo = 0, hi = 63;
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const int mid = (lo + hi + 1) >> 1;
        const int em  = __shfl(excl, mid, 64);
        if (em <= i) lo = mid; else hi = mid - 1;
      }
      const int      own = lo;
      const int      r   = i - __shfl(excl, own, 64);  // rank of the bit among the owner's 128
      const unsigned q0 = (unsigned)__shfl((int)w4.x, own, 64), q1 = (unsigned)__shfl((int)w4.y, own, 64);
      const unsigned q2 = (unsigned)__shfl((int)w4.z, own, 64), q3 = (unsigned)__shfl((int)w4.w, own, 64);
      const int      k0 = __shfl(c0, own, 64), k1 = __shfl(c1, own, 64), k2 = __shfl(c2, own, 64);
      // Every lane of the wave goes through the slice loops (the mark log's offsets come from ballots): lanes without a
      // voxel carry v = 0 and write nothing.
      const int esh = g.half ? 4 : 3;  // cells per 32-byte sector, as a shift
      int       v   = 0;
      if (active) {
        const int sel = r < k0 ? 0 : r < k1 ? 1 : r < k2 ? 2 : 3;
        unsigned  wv  = sel == 0 ? q0 : sel == 1 ? q1 : sel == 2 ? q2 : q3;
        int       rr  = r - (sel == 0 ? 0 : sel == 1 ? k0 : sel == 2 ? k1 : k2);
        while (rr-- > 0) wv &= wv - 1;  // drop the lower set bits
        v = (w0 + own * 4 + sel) * 32 + __builtin_ctz(wv);
      }
      // slice 0 (:114), then the occupied voxel's future marks (:121-170): GT velocity of the first matching record
      if (active) {
        cell_st(base, (size_t)v, 1.0F, g.half);
        ++n_marks;
      }
      float cx, cy, cz;
      g.corner_of(g.logical_of(v), pose, cx, cy, cz);  // (v is the cell's position in the slice's storage order)
      float vx = 0.f, vy = 0.f;
      // GT velocity of the FIRST record containing the voxel (:127-154).  The walk is wave-uniform — every lane visits
      // the candidates in order until all lanes have their match — and takes the candidates four at a time: their loads
      // do not depend on the tests, so four are in flight instead of one (the walk stopped at each lane's own match
      // before: one dependent L1 round trip per candidate, ~30 us per chunk of 64 voxels with the bench scene's ~280
      // candidates, which is what an agent's pre-stamp — and the tick's tail behind the last QP — waited for).
      bool found = !active;
      [[maybe_unused]] const long long ps_w0 = PS_CLK();
      // The ordered walk only has to VISIT the candidates that can contain a voxel of this chunk.  The chunk's 64 voxels
      // are neighbouring set bits (one obstacle's cross-section, usually): their xy bounding box is tested against 64
      // candidates at a time, one per lane (a cylinder whose axis is farther from the box than its radius + 1 cm cannot
      // pass the test below for any voxel in it; rings are always visited), and the walk takes the surviving candidates
      // in list order — the first match per voxel is the one the full walk finds.  ~280 visits per chunk become a
      // handful (pre-stamp wave, list in LDS: 25 -> 5 us per chunk with the bench scene).  `cl`: the culled list, in
      // LDS (pre-stamp waves) or in global memory (k_stamp_marks: one coalesced load per 64 candidates).
      auto walk_filtered = [&](const auto *cl) __attribute__((always_inline)) {
        float xlo = active ? cx : INFINITY, xhi = active ? cx : -INFINITY;
        float ylo = active ? cy : INFINITY, yhi = active ? cy : -INFINITY;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
          xlo = fminf(xlo, __shfl_xor(xlo, d, 64));
          xhi = fmaxf(xhi, __shfl_xor(xhi, d, 64));
          ylo = fminf(ylo, __shfl_xor(ylo, d, 64));
          yhi = fmaxf(yhi, __shfl_xor(yhi, d, 64));
        }
        bool done = false;
        for (int c0 = 0; c0 < n_loop && !done; c0 += 64) {  // uniform
          bool maybe = false;
          if (c0 + lane < n_loop) {
            const CylCand cc = cl[c0 + lane];
            if (cc.type == 3) {
              const float ex = fmaxf(fmaxf(xlo - cc.x, cc.x - xhi), 0.0F), ey = fmaxf(fmaxf(ylo - cc.y, cc.y - yhi), 0.0F);
              maybe          = (double)sqrtf(ex * ex + ey * ey) <= cc.wlim + 0.01;
            } else {
              maybe = cc.type == 2;
            }
          }
          unsigned long long m = __ballot(maybe);
          while (m) {  // uniform
            const int c = c0 + __builtin_ctzll(m);
            m &= m - 1;
            const CylCand cc = cl[c];
            if (!found) {
              bool hit = false;
              if (cc.type == 2) {  // ring (:137-149)
                hit = ring_contains(cyl[cc.orig], cx, cy, cz, g.res);
              } else if (cc.type == 3) {
                const float dx = cx - cc.x, dy = cy - cc.y, dz = cz - cz;
                const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
                hit = (double)dist <= cc.wlim;
              }
              if (hit) {
                vx    = cc.vx;
                vy    = cc.vy;
                found = true;
              }
            }
            if (__builtin_amdgcn_readfirstlane((int)(__ballot(!found) == 0ull))) {
              done = true;
              break;
            }
          }
        }
      };
      if (cand_lds != nullptr && n_staged == n_loop) {
        walk_filtered(cand_lds);
      } else if (culled) {
        walk_filtered(cand);
      } else {
        for (int c0 = 0; c0 < n_loop; c0 += 4) {  // uniform
          if (__builtin_amdgcn_readfirstlane((int)(__ballot(!found) == 0ull))) break;
          int    type[4], orig[4];
          float  ox[4], oy[4], wx[4], wy[4];
          double wlim[4];
  #pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int c = c0 + u < n_loop ? c0 + u : n_loop - 1;  // (clamped: the test below skips it)
            if (c < n_lds) {
              const CylCand cc = c < n_staged ? cand_lds[c] : cand[c];
              type[u] = cc.type;
              orig[u] = cc.orig;
              ox[u]   = cc.x;
              oy[u]   = cc.y;
              wx[u]   = cc.vx;
              wy[u]   = cc.vy;
              wlim[u] = cc.wlim;
            } else {
              type[u] = cyl[c].type;
              orig[u] = c;
              ox[u]   = (float)cyl[c].x;
              oy[u]   = (float)cyl[c].y;
              wx[u]   = (float)cyl[c].vx;
              wy[u]   = (float)cyl[c].vy;
              wlim[u] = cyl[c].w + (double)g.clearance;
            }
          }
  #pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (found || c0 + u >= n_loop) continue;
            bool hit = false;
            if (type[u] == 2) {  // ring (:137-149)
              hit = ring_contains(cyl[orig[u]], cx, cy, cz, g.res);
            } else if (type[u] == 3) {  // (unknown type: the reference prints a warning and goes on, :150-152)
              const float dx = cx - ox[u], dy = cy - oy[u], dz = cz - cz;
              const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
              hit = (double)dist <= wlim[u];
            }
            if (hit) {
              vx    = wx[u];
              vy    = wy[u];
              found = true;
            }
          }
        }
      }
      [[maybe_unused]] const long long ps_w1 = PS_CLK();
      PS_ADD(5, ps_w1 - ps_w0);
      PS_ADD(8, 1);
      // the cell of slice k this voxel marks (g.V: none — outside the grid, or the reference's out-of-bounds index)
      // (the z velocity is zero — (cz + (0.0F * dt) * k) - p2 is the same number for every k — so the z part of the range
      //  test and the z index are the voxel's own, computed once: one division and four comparisons less per mark)
      const float fzc = (cz + 0.0F) - p2;
      const bool  zin = fzc > -g.rz && fzc < g.rz;
      const int   izc = (int)g.div_res(fzc + g.rz);
      auto future_cell = [&](int k) -> int {
        const float fx = (cx + (vx * g.dt) * (float)k) - p0;
        const float fy = (cy + (vy * g.dt) * (float)k) - p1;
        return zin && fx > -g.rx && fx < g.rx && fy > -g.ry && fy < g.ry ? g.cell_of_xy(fx, fy, izc) : g.V;
      };
      // Mark log (sparse reset).  The lanes of a trip hold neighbouring voxels of the slice's storage order, so the marks of
      // neighbouring lanes fall into the same 32-byte sector most of the time, in slice 0 and — same velocity — in every
      // later slice: a lane logs its sector only when no lane just below it holds the same one (rows: the lower neighbour;
      // tiles: the lanes 1, 2 and 4 below — a moving obstacle's future cells straddle two tiles in alternation), and
      // marks outside the grid log nothing.  One pass per group of eight slices: the marks are stored, the group's sectors and
      // keep bits stay in registers, one atomic reserves the group's entries, then the sectors are written.  (Until round 5 a
      // second pass recomputed every future cell — three IEEE divisions each — and every keep ballot: the stamp is bound by
      // instruction issue, not by its stores.)
      const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
      // (the lanes below through DPP — v_mov_b32_dpp, no LDS round trip like __shfl_up: a lane with no source keeps the
      //  sentinel.  wave_shr:1 crosses the rows of 16; row_shr:2 / :4 do not, so a row's first lanes keep a duplicate.)
      auto keep_of = [&](unsigned sector, bool in) -> unsigned long long {
        constexpr int NONE = (int)0xFFFFFFFDu;
        bool dup = (unsigned)__builtin_amdgcn_update_dpp(NONE, (int)sector, 0x138 /* wave_shr:1 */, 0xF, 0xF, false) == sector;
        if (g.tile) {
          dup = dup || (unsigned)__builtin_amdgcn_update_dpp(NONE, (int)sector, 0x112 /* row_shr:2 */, 0xF, 0xF, false) == sector;
          dup = dup || (unsigned)__builtin_amdgcn_update_dpp(NONE, (int)sector, 0x114 /* row_shr:4 */, 0xF, 0xF, false) == sector;
#ifdef SOGM_LOOKBACK_FULL
          dup = dup || (unsigned)__builtin_amdgcn_update_dpp(NONE, (int)sector, 0x113 /* row_shr:3 */, 0xF, 0xF, false) == sector;
          dup = dup || (unsigned)__builtin_amdgcn_update_dpp(NONE, (int)sector, 0x115 /* row_shr:5 */, 0xF, 0xF, false) == sector;
          dup = dup || (unsigned)__builtin_amdgcn_update_dpp(NONE, (int)sector, 0x116 /* row_shr:6 */, 0xF, 0xF, false) == sector;
          dup = dup || (unsigned)__builtin_amdgcn_update_dpp(NONE, (int)sector, 0x117 /* row_shr:7 */, 0xF, 0xF, false) == sector;
#endif
        }
        return __ballot(active && in && !dup);
      };
      // slices in groups of SG: a group's sectors and keep bits stay in registers between the store pass and the log pass
      // (a cache of all T slices cost 40 registers and halved the occupancy of the waves that call this)
      if constexpr (CACHED) {
      constexpr int SG = 8;
      unsigned     *lent = lg.entries ? lg.entries + (size_t)agent * lg.cap : nullptr;
      // the log pass of group i runs after the store pass of group i + 1: the reservation's atomic has a group's worth of
      // work to come back in
      unsigned p_secs[SG], p_keep = 0, p_off = 0;
      int      p_n = 0;  // slices of the pending group (0: none)
      auto     flush = [&]() __attribute__((always_inline)) {
        unsigned off = (unsigned)__shfl((int)p_off, 0, 64);
#pragma unroll
        for (int q = 0; q < SG; ++q)
          if (q < p_n) {  // uniform
            const bool               mine = ((p_keep >> q) & 1u) != 0;
            const unsigned long long m    = __ballot(mine);
            const unsigned           li   = off + (unsigned)__popcll(m & lt);
            if (mine && li < (unsigned)lg.cap) lent[li] = p_secs[q];
            off += (unsigned)__popcll(m);
          }
      };
      for (int k0 = 0; k0 < g.T; k0 += SG) {  // uniform
        unsigned secs[SG];
        unsigned keepbits = 0, run = 0;
#pragma unroll
        for (int q = 0; q < SG; ++q) {
          const int k = k0 + q;
          secs[q]     = 0xFFFFFFFFu;
          if (k < g.T) {  // uniform
            const int    fv = k == 0 ? v : future_cell(k);
            const bool   in = k == 0 ? true : fv < g.V;
            const size_t ci = (size_t)k * g.V + (in ? fv : 0);
            if (k != 0 && active && in) {  // (slice 0 was stored above)
              cell_st(base, ci, 1.0F, g.half);
              ++n_marks;
            }
            if (lent) {
              secs[q]                    = in ? (unsigned)(ci >> esh) : 0xFFFFFFFFu;
              const unsigned long long m = keep_of(secs[q], in);
              keepbits |= (unsigned)((m >> lane) & 1ull) << q;
              run += (unsigned)__popcll(m);
            }
          }
        }
        if (lent) {
          unsigned off = 0;
          if (lane == 0 && run) off = atomicAdd(lg.n + agent, run);  // (consumed one group later)
          n_logged += run;
          if (p_n) flush();
#pragma unroll
          for (int q = 0; q < SG; ++q) p_secs[q] = secs[q];
          p_keep = keepbits;
          p_off  = off;
          p_n    = g.T - k0 < SG ? g.T - k0 : SG;
        }
      }
      if (lent && p_n) flush();
      } else {
        unsigned *lent = lg.entries ? lg.entries + (size_t)agent * lg.cap : nullptr;
        unsigned  pref = 0, run = 0;
        // lds_secs ([T][64] words of the wave's LDS; k_stamp_marks): the store pass leaves every slice's sector there — or
        // NOT_KEPT for a lane that does not log it — and the log pass reads it back instead of recomputing the future cell
        // and the keep ballot (half of the slice loops' instructions: the kernel is bound by instruction issue)
        constexpr unsigned NOT_KEPT = 0xFFFFFFFEu;
        {
          const unsigned long long m = keep_of((unsigned)v >> esh, true);
          run                        = (unsigned)__popcll(m);
          if (lds_secs) lds_secs[lane] = ((m >> lane) & 1ull) ? (unsigned)v >> esh : NOT_KEPT;
        }
        for (int k = 1; k < g.T; ++k) {
          const int    fv = future_cell(k);
          const bool   in = fv < g.V;
          const size_t ci = (size_t)k * g.V + (in ? fv : 0);
          if (active && in) {
            cell_st(base, ci, 1.0F, g.half);
            ++n_marks;
          }
          if (lent) {
            const unsigned long long m = keep_of(in ? (unsigned)(ci >> esh) : 0xFFFFFFFFu, in);
            if (lane == (k & 63)) pref = run;
            run += (unsigned)__popcll(m);
            if (lds_secs) lds_secs[k * 64 + lane] = ((m >> lane) & 1ull) ? (unsigned)(ci >> esh) : NOT_KEPT;
          }
        }
        if (lent && g.T <= 64) {
          unsigned lbase = 0;
          if (lane == 0) lbase = atomicAdd(lg.n + agent, run);
          lbase          = (unsigned)__shfl((int)lbase, 0, 64);
          n_logged += run;
          if (lds_secs) {
            for (int k = 0; k < g.T; ++k) {
              const unsigned           sec  = lds_secs[k * 64 + lane];  // (this lane's own word: no cross-lane hazard)
              const bool               mine = sec != NOT_KEPT;
              const unsigned long long m    = __ballot(mine);
              const unsigned           li   = lbase + (unsigned)__shfl((int)pref, k, 64) + (unsigned)__popcll(m & lt);
              if (mine && li < (unsigned)lg.cap) lent[li] = sec;
            }
          } else
          for (int k = 0; k < g.T; ++k) {
            const int                fv  = k == 0 ? v : future_cell(k);
            const bool               in  = fv < g.V;
            const unsigned           sec = in ? (unsigned)(((size_t)k * g.V + fv) >> esh) : 0xFFFFFFFFu;
            const unsigned long long m   = keep_of(sec, in);
            const unsigned           li  = lbase + (unsigned)__shfl((int)pref, k, 64) + (unsigned)__popcll(m & lt);
            if (((m >> lane) & 1ull) && li < (unsigned)lg.cap) lent[li] = sec;
          }
        } else if (lent && lane == 0) {
          // (T > 64: no log — the agent's next reset is the dense one.  A saturating mark: an add per trip could carry the
          //  unsigned counter past 2^32 and back under `cap`, and the next sparse reset would trust an empty log)
          atomicMax(lg.n + agent, (unsigned)lg.cap + 1u);
        }
      }
      PS_ADD(6, PS_CLK() - ps_w1);
    }
  }
  if (lg.stat) {
    for (int d = 32; d >= 1; d >>= 1) n_marks += (unsigned)__shfl_xor((int)n_marks, d, 64);
    if (lane == 0 && n_marks) {
      atomicAdd(lg.stat, (unsigned long long)n_marks);
      atomicAdd(lg.stat + 1, (unsigned long long)n_logged);
    }
  }
}

__global__ __launch_bounds__(64) void k_stamp_marks(GridGeom g, void *__restrict__ grid,
                                                    unsigned *__restrict__ bits, int words_per_agent,
                                                    const SogmCylinder *__restrict__ cyl, int n_cyl,
                                                    const float *__restrict__ poses,
                                                    const CylCand *__restrict__ cand_all,
                                                    const int *__restrict__ n_cand, int agent0, MarkLog lg, int lds_log) {
  extern __shared__ unsigned s_marks_secs[];  // [T][64] when the launch provides it (lds_log != 0)
  stamp_marks_trips<false>(g, grid, bits, words_per_agent, cyl, n_cyl, poses, cand_all, n_cand, (int)blockIdx.y + agent0, lg,
                           (int)blockIdx.x * 256, (int)gridDim.x * 256, nullptr, 0, lds_log ? s_marks_secs : nullptr);
}
// (the register-cached single-pass form of the slice loops: tuning key stamp_cached)
__global__ __launch_bounds__(64) void k_stamp_marks_cached(GridGeom g, void *__restrict__ grid,
                                                           unsigned *__restrict__ bits, int words_per_agent,
                                                           const SogmCylinder *__restrict__ cyl, int n_cyl,
                                                           const float *__restrict__ poses,
                                                           const CylCand *__restrict__ cand_all,
                                                           const int *__restrict__ n_cand, int agent0, MarkLog lg) {
  stamp_marks_trips<true>(g, grid, bits, words_per_agent, cyl, n_cyl, poses, cand_all, n_cand, (int)blockIdx.y + agent0, lg,
                          (int)blockIdx.x * 256, (int)gridDim.x * 256);
}

// ------------------------------------------------------------------------------------------------
// neighbour overlay
// ------------------------------------------------------------------------------------------------
// Mark-log reservation of `count` entries for every lane active at the call (count uniform): one atomic per agent
// present among those lanes — a wave of the overlay holds one or two agents — instead of one per lane (2540 lanes
// per agent hammering one counter cost the overlay 0.8 ms).
__device__ inline unsigned log_reserve(const MarkLog &lg, int agent, unsigned count) {
  const int                lane = (int)(threadIdx.x & 63);
  const unsigned long long lt   = lane ? (~0ull >> (64 - lane)) : 0ull;
  unsigned long long       todo = __ballot(1);
  unsigned                 base = 0;
  while (todo) {  // uniform over the active lanes
    const int                leader = __ffsll((long long)todo) - 1;
    const int                la     = __shfl(agent, leader, 64);
    const unsigned long long grp    = __ballot(agent == la) & todo;
    unsigned                 b      = 0;
    if (lane == leader) b = atomicAdd(lg.n + la, count * (unsigned)__popcll(grp));
    b = (unsigned)__shfl((int)b, leader, 64);
    if (agent == la) base = b + count * (unsigned)__popcll(grp & lt);
    todo &= ~grp;
  }
  return base;
}

// One lane per (agent, record, slice).  The reference's is_swarm_traj_valid chain
// (risk_base.cpp:154-159) collapses to: record contributes at slice t  iff
//     time_start < t_abs(0)  and  t_abs(s) < time_end for every s <= t
// and t_abs is increasing, i.e.  time_start < t_abs(0) && t_abs(t) < time_end.
// one (agent, record, slice) item of the overlay
__device__ inline void splat_item(const GridGeom &g, void *__restrict__ grid, const SogmTrajRecord &R, int agent,
                                  int t, const int32_t *__restrict__ ego_ids, const float *__restrict__ poses,
                                  const double *__restrict__ stamps, const double *__restrict__ body, int n_body,
                                  const MarkLog &lg) {
  // mark log (sparse reset): a lane that writes reserves n_body entries (or one) and fills them with the sectors of
  // its cells, ~0 for a particle outside the grid
  const int    esh   = g.half ? 4 : 3;
  const size_t tslab = (size_t)t * g.V;
  unsigned    *lent  = lg.entries ? lg.entries + (size_t)agent * lg.cap : nullptr;
  if (R.n_pieces <= 0 || R.drone_id == ego_ids[agent]) return;
  double time_end = R.time_start;
  for (int k = 0; k < R.n_pieces; ++k) time_end += R.duration[k];
  const double stamp = stamps[agent];
  const double t0    = stamp + (double)(g.dt * (float)0);
  const double tt    = stamp + (double)(g.dt * (float)t);
  if (g.map_kind == SOGM_MAP_RISKVOXEL) {
    // RiskVoxel::addOtherAgents (risk_voxel.cpp:258-288) calls getWaypoints (particles.cpp:316-344)
    // directly and SETS cells (:311-318).  The record is visited at slice t iff every earlier slice
    // returned true (running or not yet started); on the slice where it has ended its last point is
    // stamped once, without body particles.
    for (int s_ = 0; s_ < t; ++s_) {
      const double ts = stamp + (double)(g.dt * (float)s_);
      if (!((R.time_start < ts && time_end > ts) || R.time_start > ts)) return;
    }
    const float *pose = poses + agent * 3;
    const double q0 = (double)pose[0], q1 = (double)pose[1], q2 = (double)pose[2];
    char        *slab = reinterpret_cast<char *>(grid) + ((size_t)agent * g.T + t) * (size_t)g.V * (g.half ? 2 : 4);
    double       p[3];
    if (R.time_start < tt && time_end > tt) {
      bezier_pos(R, tt - R.time_start, p);
      const unsigned lb = lent ? log_reserve(lg, agent, (unsigned)n_body) : 0u;
      for (int e = 0; e < n_body; ++e) {
        const float fx = (float)((p[0] + body[e * 3 + 0]) - q0);
        const float fy = (float)((p[1] + body[e * 3 + 1]) - q1);
        const float fz = (float)((p[2] + body[e * 3 + 2]) - q2);
        const int   vl = g.in_range(fx, fy, fz) ? g.voxel_of(fx, fy, fz) : g.V;
        const bool  in = vl < g.V;  // (index >= V: the reference's out-of-bounds case, see k_stamp_bits)
        const int   vx = in ? g.phys_of(vl) : g.V;
        if (in) cell_st(slab, vx, 1.0F, g.half);
        if (lent && lb + (unsigned)e < (unsigned)lg.cap) lent[lb + e] = in ? (unsigned)((tslab + vx) >> esh) : 0xFFFFFFFFu;
      }
    } else if (time_end < tt) {
      double dur = 0.0;
      for (int k = 0; k < R.n_pieces; ++k) dur += R.duration[k];
      bezier_pos(R, dur, p);
      const float fx = (float)(p[0] - q0), fy = (float)(p[1] - q1), fz = (float)(p[2] - q2);
      const int vl = g.in_range(fx, fy, fz) ? g.voxel_of(fx, fy, fz) : g.V;
      const int vx = vl < g.V ? g.phys_of(vl) : g.V;
      if (vx < g.V) {
        cell_st(slab, vx, 1.0F, g.half);
        if (lent) {
          const unsigned lb = atomicAdd(lg.n + agent, 1u);
          if (lb < (unsigned)lg.cap) lent[lb] = (unsigned)((tslab + vx) >> esh);
        }
      }
    }
    return;
  }
  if (!(R.time_start < t0 && time_end > t0)) return;  // chain broken at slice 0
  if (!(R.time_start < tt && time_end > tt)) return;

  double p[3];
  bezier_pos(R, tt - R.time_start, p);
  const float *pose = poses + agent * 3;
  const double q0 = (double)pose[0], q1 = (double)pose[1], q2 = (double)pose[2];
  char        *slab = reinterpret_cast<char *>(grid) + ((size_t)agent * g.T + t) * (size_t)g.V * (g.half ? 2 : 4);
  // getParticlesWithRisk (particles.cpp:365-409): pos_stddev = replan_risk_rate * (t - time_start) in float; below 1e-3
  // (every shipped configuration: the rate is 0) each body particle counts 1.0; otherwise each is replaced by
  // num_resample Gaussian samples whose weights are normalised to num_resample per particle.
  const float sd = g.rs_rate * (float)(tt - R.time_start);
  if (g.rs_n > 0 && g.rs_z != nullptr && !(sd < 1e-3F)) {
    // The reference draws from std::default_random_engine(time(NULL)) — re-seeded at every call, so every call within
    // one second sees the same sequence: the injected table plays that sequence, entry 3 (e n + i) + d for sample i of
    // particle e, axis d (sogm_set_resample).  Weights: float arithmetic in the reference's order; exp through
    // sogm_det::expf_neg on both sides.  (RiskBase::addOtherAgents indexes ONE risks vector, cleared by every
    // getParticlesWithRisk call, with the particles of ALL agents — an out-of-bounds read once two agents contribute;
    // here and in the oracle a particle adds its own weight, the evident intent.)
    const int      n  = g.rs_n;
    const unsigned lb = lent ? log_reserve(lg, agent, (unsigned)(n_body * n)) : 0u;
    for (int e = 0; e < n_body; ++e) {
      const double w0 = p[0] + body[e * 3 + 0], w1 = p[1] + body[e * 3 + 1], w2 = p[2] + body[e * 3 + 2];
      const float *z  = g.rs_z + 3 * e * n;
      float        sum = 0.0F;  // std::accumulate(risk_buf.begin(), risk_buf.end(), 0.0f)
      for (int i = 0; i < n; ++i) {
        const float nx = z[3 * i] * sd, ny = z[3 * i + 1] * sd, nz = z[3 * i + 2] * sd;
        sum += sogm_det::expf_neg((-0.5F * ((nx * nx + ny * ny) + nz * nz)) / (sd * sd));
      }
      for (int i = 0; i < n; ++i) {
        const float nx = z[3 * i] * sd, ny = z[3 * i + 1] * sd, nz = z[3 * i + 2] * sd;
        const float rk = (sogm_det::expf_neg((-0.5F * ((nx * nx + ny * ny) + nz * nz)) / (sd * sd)) * (float)n) / sum;
        const float fx = (float)((w0 + (double)nx) - q0);
        const float fy = (float)((w1 + (double)ny) - q1);
        const float fz = (float)((w2 + (double)nz) - q2);
        const int   vl = g.in_range(fx, fy, fz) ? g.voxel_of(fx, fy, fz) : g.V;
        const bool  in = vl < g.V;
        const int   vx = in ? g.phys_of(vl) : g.V;
        const unsigned li = lb + (unsigned)(e * n + i);
        if (lent && li < (unsigned)lg.cap) lent[li] = in ? (unsigned)((tslab + vx) >> esh) : 0xFFFFFFFFu;
        if (in) cell_add(slab, vx, rk, g.half);  // (fractional weights: the sum depends on the order in the last bits)
      }
    }
    return;
  }
  // a neighbour farther outside the map than its body reaches marks nothing: no log entries, no 36 range tests (most
  // (agent, record) pairs of a swarm spread over many map widths; 1 cm of slack covers the fp32 rounding of the test below)
  if (fabs(p[0] - q0) > (double)g.rx + (double)g.body_ext[0] + 0.01 || fabs(p[1] - q1) > (double)g.ry + (double)g.body_ext[1] + 0.01 ||
      fabs(p[2] - q2) > (double)g.rz + (double)g.body_ext[2] + 0.01)
    return;
  const unsigned lb = lent ? log_reserve(lg, agent, (unsigned)n_body) : 0u;
  for (int e = 0; e < n_body; ++e) {
    const float fx = (float)((p[0] + body[e * 3 + 0]) - q0);
    const float fy = (float)((p[1] + body[e * 3 + 1]) - q1);
    const float fz = (float)((p[2] + body[e * 3 + 2]) - q2);
    const int   vl = g.in_range(fx, fy, fz) ? g.voxel_of(fx, fy, fz) : g.V;
    const bool  in = vl < g.V;
    const int   vx = in ? g.phys_of(vl) : g.V;
    if (lent && lb + (unsigned)e < (unsigned)lg.cap) lent[lb + e] = in ? (unsigned)((tslab + vx) >> esh) : 0xFFFFFFFFu;
    if (!in) continue;
    // += 1.0f per body particle; sums of 1.0 are exact in fp32 (and in fp16 up to 2048), so the order
    // is immaterial
    cell_add(slab, vx, 1.0F, g.half);
  }
}
__global__ __launch_bounds__(256) void k_splat_neighbours(
    GridGeom g, void *__restrict__ grid, const SogmTrajRecord *__restrict__ rec, int n_rec,
    const int32_t *__restrict__ ego_ids, const float *__restrict__ poses,
    const double *__restrict__ stamps, const double *__restrict__ body, int n_body, int n_agents, int agent0,
    MarkLog lg, const int *wait_stage, int *wait_err) {
  const long long total  = (long long)n_agents * n_rec * g.T;
  const long long stride = (long long)gridDim.x * blockDim.x;  // (one item per lane unless the launch is narrower)
  for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += stride) {
    const int t     = (int)(gid % g.T);
    const int r     = (int)((gid / g.T) % n_rec);
    const int agent = agent0 + (int)(gid / ((long long)g.T * n_rec));
    if (wait_stage) {
      // launched under the tail of the pre-stamp that builds this grid (sogm_update_prestamped): an agent's overlay
      // starts when ITS stamp is complete — stores of 1.0 first, the overlay's additions after them, as in the serial
      // order.  The launch is narrow (the waiting lanes must leave room for the pre-stamp's own waves: a full-size
      // grid of pollers starved a pre-stamp that was not resident yet until the timeouts fired).  Bounded: a
      // pre-stamp that failed has set the error word (the tick is reported as failed); nothing is written then, and an
      // overlay whose own wait runs out sets it (code 8).
      const long long t0 = wall_clock64();
      bool            ok = false;
      for (;;) {
        if (__hip_atomic_load(wait_stage + agent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= FLOW_PS_DONE) {
          ok = true;
          break;
        }
        if (__hip_atomic_load(wait_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        if (wall_clock64() - t0 > FLOW_TIMEOUT_TICKS) {
          atomicExch(wait_err, 8);  // the tick is reported as failed (the report runs behind the pre-stamp), not
          break;                    // silently left without its overlay
        }
        flow_pause();
      }
      if (!ok) return;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    splat_item(g, grid, rec[r], agent, t, ego_ids, poses, stamps, body, n_body, lg);
  }
}

// ------------------------------------------------------------------------------------------------
// queries
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_query_clear(MapView m, const int32_t *__restrict__ agent,
                                                     const double *__restrict__ pos,
                                                     const double *__restrict__ t, int t_is_index,
                                                     int n, int8_t *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = agent[i];
  int       r;
  if (t_is_index) {
    r = query_clear_idx(m, a, pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2], (int)t[i]);
  } else {
    r = query_clear_time(m, a, pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2], t[i]);
  }
  out[i] = (int8_t)r;
}

// Order-preserving extraction.  Cells of the box are visited in the reference's z,y,x order, 256
// cells per trip; each lane counts the slices of its cell that exceed the threshold, a block-wide
// exclusive scan turns counts into output offsets, so the emitted sequence is exactly the
// reference's (FIRI's greedy selection breaks ties by point order).
__global__ __launch_bounds__(256) void k_obstacle_points(
    MapView m, const int32_t *__restrict__ agent_idx, const double *__restrict__ box_lo,
    const double *__restrict__ box_hi, const double *__restrict__ t0v,
    const double *__restrict__ t1v, double *__restrict__ out_pts, int32_t *__restrict__ out_cnt,
    int cap) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const GridGeom &g     = m.g;
  const int       b     = blockIdx.x;
  const int       agent = agent_idx[b];
  const float    *pose  = m.poses + agent * 3;
  const double    stamp = m.stamps[agent];
  const double    tr    = (double)g.dt;
  int             js    = (int)floor((t0v[b] - stamp) / tr);
  int             je    = (int)ceil((t1v[b] - stamp) / tr);
  js                    = js < 0 ? 0 : js;
  js                    = js > g.T ? g.T : js;
  je                    = je > g.T ? g.T : je;
  je                    = je < 0 ? 0 : je;
  if (je > g.T - 1) je = g.T - 1;  // slices >= T do not exist (reference reads one past the end)
  int lx = (int)((box_lo[b * 3 + 0] - pose[0] + g.rx) / g.res);
  int ly = (int)((box_lo[b * 3 + 1] - pose[1] + g.ry) / g.res);
  int lz = (int)((box_lo[b * 3 + 2] - pose[2] + g.rz) / g.res);
  int hx = (int)((box_hi[b * 3 + 0] - pose[0] + g.rx) / g.res);
  int hy = (int)((box_hi[b * 3 + 1] - pose[1] + g.ry) / g.res);
  int hz = (int)((box_hi[b * 3 + 2] - pose[2] + g.rz) / g.res);
  hx     = min(hx, g.L - 1);
  hy     = min(hy, g.W - 1);
  hz     = min(hz, g.H - 1);
  lx     = max(lx, 0);
  ly     = max(ly, 0);
  lz     = max(lz, 0);
  const int nx = hx - lx + 1, ny = hy - ly + 1, nz = hz - lz + 1;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  if (nx <= 0 || ny <= 0 || nz <= 0 || js > je) {
    if (threadIdx.x == 0) out_cnt[b] = 0;
    return;
  }
  const int    cells = nx * ny * nz;
  const void  *grid0 = m.slab(agent, 0);
  double      *outp  = out_pts + (size_t)b * cap * 3;
  const int    lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  for (int c0 = 0; c0 < cells; c0 += blockDim.x) {
    const int c    = c0 + threadIdx.x;
    int       cnt  = 0;
    unsigned  mask = 0;  // bit j-js set when slice j exceeds
    int       vi   = 0;
    if (c < cells) {
      const int x = lx + c % nx;
      const int y = ly + (c / nx) % ny;
      const int z = lz + c / (nx * ny);
      vi          = x + y * g.L + z * g.L * g.W;
      const int pi = g.phys(x, y, z);
      for (int j = js; j <= je; ++j) {
        const float thr =
            g.map_kind == SOGM_MAP_FAKE ? g.risk_threshold : g.risk_threshold - g.decay_voxel * (float)j;
        if (cell_ld(grid0, (size_t)j * g.V + pi, g.half) > thr) {
          ++cnt;
          mask |= 1u << (j - js);
        }
      }
    }
    // wave-level inclusive scan (64 lanes), then 4-wave combine through LDS
    int incl = cnt;
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int wave_off = 0, total = 0;
    for (int w = 0; w < 4; ++w) {
      const int v = s_wave[w];
      if (w < wave) wave_off += v;
      total += v;
    }
    const int base = s_base;
    int       off  = base + wave_off + incl - cnt;
    if (cnt) {
      float fx, fy, fz;
      g.corner_of(vi, pose, fx, fy, fz);
      for (int j = 0; j < 32 && (mask >> j); ++j) {
        if ((mask >> j) & 1u) {
          if (off < cap) {
            outp[off * 3 + 0] = (double)fx;
            outp[off * 3 + 1] = (double)fy;
            outp[off * 3 + 2] = (double)fz;
          }
          ++off;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base = base + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) out_cnt[b] = s_base;
}

// ------------------------------------------------------------------------------------------------
// layout converters ([T][V] slabs <-> reference [V][T])
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_slabs_to_vt(const void *__restrict__ slabs, GridGeom g, float *__restrict__ vt) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= g.V) return;
  const int p = g.phys_of(v);
  for (int t = 0; t < g.T; ++t) vt[(size_t)v * g.T + t] = cell_ld(slabs, (size_t)t * g.V + p, g.half);
}
__global__ __launch_bounds__(256) void k_vt_to_slabs(const float *__restrict__ vt, GridGeom g, void *__restrict__ slabs) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= g.V) return;
  const int p = g.phys_of(v);
  for (int t = 0; t < g.T; ++t) cell_st(slabs, (size_t)t * g.V + p, vt[(size_t)v * g.T + t], g.half);
}

// Bezier pos / vel / acc of a trajectory record at an absolute time (bernstein.cpp:25-59)
// returns false (and zeros) for an empty record
__device__ inline bool traj_eval_record(const SogmTrajRecord &r, double t_abs, double out[9]) {
  if (r.n_pieces <= 0) {
    for (int k = 0; k < 9; ++k) out[k] = 0.0;
    return false;
  }
  double total = 0;
  for (int k = 0; k < r.n_pieces; ++k) total += r.duration[k];
  double tt = t_abs - r.time_start;
  tt        = tt < 0 ? 0 : (tt > total ? total : tt);
  // locatePiece (bernstein.hpp:164-172)
  int    piece = r.n_pieces - 1;
  double rem   = tt;
  for (int k = 0; k < r.n_pieces; ++k) {
    rem -= r.duration[k];
    if (rem < 0) {
      piece = k;
      break;
    }
  }
  double t0 = 0;
  for (int k = 0; k < piece; ++k) t0 += r.duration[k];
  const double tf = t0 + r.duration[piece], dur = tf - t0, s = (tt - t0) / dur;
  const double A[5][5] = {{1, -4, 6, -4, 1}, {0, 4, -12, 12, -4}, {0, 0, 6, -12, 6},
                          {0, 0, 0, 4, -4},  {0, 0, 0, 0, 1}};
  const double S0[5] = {1, s, s * s, s * s * s, (s * s) * (s * s)};
  const double S1[5] = {0, 1, 2 * s, 3 * (s * s), 4 * (s * s * s)};
  const double S2[5] = {0, 0, 2, 6 * s, 12 * (s * s)};
  const double *c    = r.cpts + piece * 15;
  for (int d = 0; d < 3; ++d) {
    double p = 0, v = 0, a = 0;
    for (int j = 0; j < 5; ++j) {
      double b = 0;
      for (int q = 0; q < 5; ++q) b += c[q * 3 + d] * A[q][j];
      p += b * S0[j];
      v += b * S1[j];
      a += b * S2[j];
    }
    out[d]     = p;
    out[3 + d] = v / dur;
    out[6 + d] = a / (dur * dur);
  }
  return true;
}
__global__ __launch_bounds__(64) void k_traj_eval(const SogmTrajRecord *__restrict__ rec, int n,
                                                  const double *__restrict__ t,
                                                  double *__restrict__ out, int32_t *__restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double o[9];
  ok[i] = traj_eval_record(rec[i], t[i], o) ? 1 : 0;
  for (int k = 0; k < 9; ++k) out[i * 9 + k] = o[k];
}

// Start of a replan tick for n agents in one launch (the tick driver's glue): the replan start state of every agent
// from the trajectory it is executing — FiniteStateMachine samples traj_ at the planned start time
// (plan_manager/src/plan_manager.cpp:169-175), an agent without a trajectory starts from where it hovers
// (odom, :127-133) — plus the stamps the map update and the replan take.
// One wave per agent: the 2064-byte record is fetched with ONE coalesced batch of loads into LDS (a lane walking
// n_pieces -> durations -> control points through global memory needs four dependent round trips, which the tail
// of the streaming clear beside this kernel stretches to a millisecond each), lane 0 evaluates it there.
// One agent's tick inputs (plan_manager.cpp:169-175): the start state sampled from its executing trajectory at
// stamp + start_offset, or where it hovers; the map centre of the tick is that position.
__device__ inline void tick_inputs_agent(const SogmTrajRecord &rec, const double *hov, int i, double stamp,
                                         double start_offset, double *__restrict__ hover, double *__restrict__ now,
                                         double *__restrict__ t_start, double *__restrict__ pva,
                                         float *__restrict__ poses) {
  const double ts = stamp + start_offset;
  double       o[9];
  if (!traj_eval_record(rec, ts, o))
    for (int k = 0; k < 9; ++k) o[k] = hov[k];
  for (int k = 0; k < 9; ++k) pva[i * 9 + k] = o[k];
  for (int k = 0; k < 3; ++k) {
    hover[i * 9 + k]     = o[k];
    hover[i * 9 + 3 + k] = 0.0;
    hover[i * 9 + 6 + k] = 0.0;
    poses[i * 3 + k]     = (float)o[k];
  }
  now[i]     = stamp;
  t_start[i] = ts;
}
__global__ __launch_bounds__(64) void k_tick_inputs(const SogmTrajRecord *__restrict__ own, int n, double stamp,
                                                    double start_offset, double *__restrict__ hover,
                                                    double *__restrict__ now, double *__restrict__ t_start,
                                                    double *__restrict__ pva, float *__restrict__ poses) {
  __shared__ __attribute__((aligned(16))) SogmTrajRecord s_rec;
  __shared__ double                                      s_hov[9];
  const int i = blockIdx.x;
  if (i >= n) return;
  constexpr int W = (int)(sizeof(SogmTrajRecord) / 16);
  const uint4  *src = reinterpret_cast<const uint4 *>(own + i);
  uint4        *dst = reinterpret_cast<uint4 *>(&s_rec);
  for (int w = threadIdx.x; w < W; w += 64) dst[w] = src[w];
  if (threadIdx.x < 9) s_hov[threadIdx.x] = hover[i * 9 + threadIdx.x];
  __syncthreads();
  if (threadIdx.x != 0) return;
  tick_inputs_agent(s_rec, s_hov, i, stamp, start_offset, hover, now, t_start, pva, poses);
}

// ------------------------------------------------------------------------------------------------
// Pre-stamp: the NEXT tick's map built inside this tick's replan (dataflow replan only).
// The tick's critical path was "stamp (1.8 ms, nothing else running) -> chain of the slowest agent"; but an agent's
// next map centre only depends on its OWN new record, which is final the moment k_finish_flow publishes it.  This
// persistent kernel (launched behind the gate that holds stores back until every agent's corridors are final)
// takes agents in publication order and, for each: samples the start state of the next tick (k_tick_inputs' rule),
// culls the cylinders, sets the occupancy bits and writes the marks + log entries into the pool's next grid — split
// into ps.n_bits + ps.n_marks one-wave tickets per agent (64 + 64 by default) handed out in order (a ticket only ever waits for lower ones, so
// any number of resident waves makes progress).  The next update then only adopts the grid and adds the overlay
// (sogm_update_prestamped).  Same kernels' code, same cells.
// ------------------------------------------------------------------------------------------------
__device__ inline int flow_wait_count(int *p, int target, int *err) {
  const long long t0 = wall_clock64();
  for (;;) {
    const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (v >= target) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      return 0;
    }
    flow_pause();
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0)
      return -1;
    if (wall_clock64() - t0 > FLOW_TIMEOUT_TICKS) {
      if ((threadIdx.x & 63) == 0) atomicExch(err, 6);
      return -1;
    }
  }
}
// Launch gate of the pre-stamp (one lane, on its stream in front of it): every agent's corridors final — store streams
// stay away from the searches and point scans — AND every QP workgroup and finishing wave of this replan resident:
// the pre-stamp's waves wait for what those produce, and a QP workgroup that is not placed yet (its launch can sit
// behind another kernel when streams share a hardware queue) needs a whole CU, which thousands of small waiting
// waves would never leave it.  Bounded like every wait of the tick: on a timeout the tick fails, the waves drain.
__global__ void k_prestamp_gate(int *hdr, int n_agents, int n_qp, int n_finish) {
  if (threadIdx.x != 0) return;
  const long long t0 = wall_clock64();
  for (;;) {
    if (__hip_atomic_load(&hdr[FLOW_ERR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    if (__hip_atomic_load(&hdr[FLOW_Q_READY_N], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_agents &&
        __hip_atomic_load(&hdr[FLOW_Q_RESIDENT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_qp &&
        __hip_atomic_load(&hdr[FLOW_F_TICKET], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_finish)
      return;
    if (wall_clock64() - t0 > FLOW_TIMEOUT_TICKS) {
      atomicExch(&hdr[FLOW_ERR], 7);
      return;
    }
    flow_pause();
  }
}
__global__ __launch_bounds__(64) void k_prestamp_flow(GridGeom g, FlowCtl fc, PrestampDev ps) {
  __shared__ __attribute__((aligned(16))) SogmTrajRecord s_rec;
  __shared__ double                                      s_hov[9];
  __shared__ __attribute__((aligned(16))) CylCand        s_cand[PRESTAMP_CAND_LDS];
  const int lane  = threadIdx.x;
  // the last agents to be published get finer tickets: nobody is left to share the waves with, and their stamps are
  // what trails the replan
  const int n_early = ps.n_agents > ps.n_late ? ps.n_agents - ps.n_late : 0;
  const int per_e = ps.n_bits + ps.n_marks, per_l = ps.n_bits_late + ps.n_marks_late;
  const int total = n_early * per_e + (ps.n_agents - n_early) * per_l;
  for (;;) {
    const int t = flow_ticket(&fc.hdr[FLOW_P_TICKET]);
    if (t >= total) break;
    [[maybe_unused]] const long long ps_t0 = PS_CLK();
    PS_ADD(7, 1);
    const bool late = t >= n_early * per_e;
    const int  tl   = late ? t - n_early * per_e : t;
    const int  per  = late ? per_l : per_e;
    const int  n_bits = late ? ps.n_bits_late : ps.n_bits, n_marks = late ? ps.n_marks_late : ps.n_marks;
    const int agent = flow_wait_slot(fc.p_ready + (late ? n_early : 0) + tl / per, &fc.hdr[FLOW_ERR]);
    if (agent < 0) break;
    PS_ADD(0, PS_CLK() - ps_t0);
    __threadfence();  // the agent's own record was published before its slot
    const int  s   = tl % per;
    long long *pts = fc.ts + 8 * (size_t)ps.n_agents + 4 * (size_t)agent;  // diagnostics (sogm_debug_prestamp_times)
    if (s == 0) {
      [[maybe_unused]] const long long ps_c0 = PS_CLK();
      if (lane == 0) pts[0] = wall_clock64();
      // next tick's inputs of this agent (k_tick_inputs), then its candidate cylinders around the new centre
      constexpr int W   = (int)(sizeof(SogmTrajRecord) / 16);
      const uint4  *src = reinterpret_cast<const uint4 *>(ps.own + agent);
      uint4        *dst = reinterpret_cast<uint4 *>(&s_rec);
      for (int w = lane; w < W; w += 64) dst[w] = src[w];
      if (lane < 9) s_hov[lane] = ps.hover[agent * 9 + lane];
      __syncthreads();
      if (lane == 0) {
        tick_inputs_agent(s_rec, s_hov, agent, ps.stamp, ps.start_offset, ps.hover, ps.now, ps.t_start, ps.pva, ps.poses);
        ps.stamps[agent] = ps.stamp;
        if (ps.poses_host)
          for (int k = 0; k < 3; ++k) ps.poses_host[agent * 3 + k] = ps.poses[agent * 3 + k];
      }
      __threadfence();
      __syncthreads();
      cull_agent(g, ps.cyl, ps.n_cyl, ps.poses[agent * 3], ps.poses[agent * 3 + 1],
                 (CylCand *)ps.cand + (size_t)agent * SOGM_MAX_CYL_LDS, ps.n_cand + agent, lane);
      if (ps.cb.bounds) cull_blocks_agent(g, ps.cb, agent, ps.poses[agent * 3], ps.poses[agent * 3 + 1], lane);
      __threadfence();
      if (lane == 0) {
        pts[1] = wall_clock64();
        atomicAdd(&fc.stage[agent], 1);
      }
      PS_ADD(2, PS_CLK() - ps_c0);
    }
    [[maybe_unused]] const long long ps_t1 = PS_CLK();
    if (s < n_bits) {
      if (flow_wait_count(&fc.stage[agent], 1, &fc.hdr[FLOW_ERR])) break;
      [[maybe_unused]] const long long ps_t2 = PS_CLK();
      PS_ADD(1, ps_t2 - ps_t1);
      const float p0 = ps.poses[agent * 3], p1 = ps.poses[agent * 3 + 1], p2 = ps.poses[agent * 3 + 2];
      if (ps.cb.bounds) {
        stamp_bits_blocks(g, ps.cloud, ps.cb, agent, s, n_bits, p0, p1, p2, ps.bits + (size_t)agent * ps.words, lane);
      } else {
        const int begin = ps.cloud_range[agent * 2], end = ps.cloud_range[agent * 2 + 1];
        stamp_bits_range(g, ps.cloud, begin + s * 64 + lane, end, n_bits * 64, p0, p1, p2,
                         ps.bits + (size_t)agent * ps.words);
      }
      __threadfence();
      PS_ADD(3, PS_CLK() - ps_t2);
      if (lane == 0 && atomicAdd(&fc.stage[agent], 1) + 1 == 1 + n_bits) pts[2] = wall_clock64();
    } else {
      if (flow_wait_count(&fc.stage[agent], 1 + n_bits, &fc.hdr[FLOW_ERR])) break;
      [[maybe_unused]] const long long ps_t2 = PS_CLK();
      PS_ADD(1, ps_t2 - ps_t1);
      stamp_marks_trips<true>(g, ps.grid, ps.bits, ps.words, ps.cyl, ps.n_cyl, ps.poses, (const CylCand *)ps.cand, ps.n_cand, agent, ps.lg,
                        (s - n_bits) * 256, n_marks * 256, s_cand, PRESTAMP_CAND_LDS);
      __syncthreads();  // (the next ticket's staging overwrites s_cand)
      // the agent's last marks ticket to finish declares its grid complete (the next update's overlay waits for it)
      __threadfence();
      PS_ADD(4, PS_CLK() - ps_t2);
      if (lane == 0 && atomicAdd(&fc.stage[agent], 1) + 1 == 1 + n_bits + n_marks) {
        pts[3] = wall_clock64();
        __hip_atomic_store(&fc.stage[agent], FLOW_PS_DONE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
#ifdef SOGM_PROFILE_PRESTAMP
}  // namespace sogm
extern "C" int sogm_debug_prestamp_prof(unsigned long long *out12_host, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (out12_host && hipMemcpyFromSymbol(out12_host, HIP_SYMBOL(sogm::g_ps_prof), sizeof(unsigned long long) * 12) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[12] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(sogm::g_ps_prof), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
namespace sogm {
#endif
// ------------------------------------------------------------------------------------------------
// Flight kernel M (sogm_flight_run; the flight is described in sogm_planner.hpp): an agent's map of its next tick, built
// the moment the agent's previous tick is finished.  Per (agent, tick) 1 + n_reset + n_bits + n_marks + n_splat one-wave
// tickets in four phases; every phase is a queue of its own and a wave claims — never blocking — a ticket of the LATEST
// phase that has one (a first version handed all tickets of an item out in order and let them wait for each other: 512
// waves / 61 tickets = 8 items in flight, the swarm's maps took 17 ms per tick).  Only the head may wait: for the swarm's
// tick k - 2.
//   head    gate "every agent has finished tick k - 2" (the staleness rule's other half: table ver(k - 2) is complete);
//           start state / map centre / stamp of tick k from the agent's executed record (k_tick_inputs' rule);
//           candidate cylinders and cloud blocks of frame k around the new centre
//   reset   the agent's grid back to zero through its mark log (k_reset_sectors' 2-lanes-per-sector form; an overflowed
//           log: the whole grid), then the log restarts
//   bits    occupancy bits of slice 0 from the listed cloud blocks
//   marks   slice 0 + T - 1 future marks per occupied voxel, logged
//   splat   the neighbours' records of table ver(k - 2), logged -> the agent goes to the search queue
// Same device functions as the per-tick kernels: same cells.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void flight_reset_ticket(char *__restrict__ base, size_t agent_bytes, const unsigned *__restrict__ e,
                                                    unsigned n, int cap, int first, int stride, int lane,
                                                    unsigned long long *__restrict__ stat) {
  const vfloat4 z = {0.f, 0.f, 0.f, 0.f};
  if (n > (unsigned)cap) {  // overflowed log: this ticket's share of the whole grid (agent_bytes is a multiple of 32)
    const size_t nv = agent_bytes / 16;
    for (size_t i = (size_t)first * 64 + lane; i < nv; i += (size_t)stride * 64)
      __builtin_nontemporal_store(z, reinterpret_cast<vfloat4 *>(base) + i);
    return;
  }
  const int part = lane & 1, pair = lane >> 1;
  unsigned  n_lines = 0;
  // 32 entries per trip, two lanes each; EIGHT trips' entry loads are issued together (a trip by itself is one dependent
  // load -> store pair: ~1 us per trip, 1.05 ms per ticket of 40 k entries in the first flights)
  constexpr int U = 8;
  const size_t  step = (size_t)stride * 32;
  for (size_t i0 = (size_t)first * 32; i0 < n; i0 += U * step) {
    unsigned sct[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * step + pair;
      sct[u]         = i < n ? e[i] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned prev = __shfl_up(sct[u], 2);
      if (sct[u] == 0xFFFFFFFFu || (pair > 0 && prev == sct[u])) continue;
      const size_t off = (size_t)sct[u] * 32 + 16 * part;
      if (off + 16 <= agent_bytes) *reinterpret_cast<vfloat4 *>(base + off) = z;
      if (part == 0) ++n_lines;
    }
  }
  if (stat) {
    for (int d = 32; d >= 1; d >>= 1) n_lines += (unsigned)__shfl_xor((int)n_lines, d, 64);
    if (lane == 0 && n_lines) atomicAdd(stat + 2, (unsigned long long)n_lines * 32ull);
  }
}
__global__ __launch_bounds__(64) void k_flight_map(GridGeom g, FlightCtl fl, FlightMapDev d, unsigned long long *reset_stat) {
  __shared__ __attribute__((aligned(16))) SogmTrajRecord s_rec;
  __shared__ double                                      s_hov[9];
  __shared__ __attribute__((aligned(16))) CylCand        s_cand[PRESTAMP_CAND_LDS];
  const int lane = threadIdx.x;
  const int A    = fl.n_agents;
  // (the urgent lane cuts a map into finer tickets: a map's latency is the sum of its phases' longest tickets — 0.25 reset +
  //  0.3 marks + 0.17 overlay ms with the plain lane's counts even on idle workers — and the urgent lane exists for latency)
  int      *err = &fl.hdr[FL_ERR];
  fl_wg_started(fl, 3);
  // (every lane of this kernel ends when the call's last finish has stored the call's epoch in hdr[FL_END])
  if ((int)blockIdx.x < d.n_head_wgs) {
    // ---- admitting waves: heads, in the order the agents' previous ticks finished; the first n_uhead_wgs serve the
    // urgent ring: no admission order, no pace, no window (the gate is open for an agent that is behind) ----
    const bool urgent = (int)blockIdx.x < d.n_uhead_wgs;
    const int  n_r = urgent ? d.un_reset : d.n_reset, n_b = urgent ? d.un_bits : d.n_bits;
    unsigned long long *const wq      = urgent ? fl.uw : fl.mw;
    int *const                wq_tail = &fl.hdr[urgent ? FL_UW_TAIL : FL_MW_TAIL];
    for (;;) {
      const int t     = flow_ticket(&fl.hdr[urgent ? FL_U_TICKET : FL_M_TICKET]);
      const int agent = fl_wait_item_end(urgent ? fl.u_ring : fl.m_ring, fl.ring_mask, t, err, &fl.hdr[FL_END], fl.epoch, !urgent);
      if (agent < 0) break;
      __threadfence();
      const int          k = fl.tick_of[agent], kl = k - fl.first_tick;
      const FlightWorld &w = d.worlds[kl];
      CloudBlocks        cb = d.cb;
      cb.bounds       = w.bounds;
      cb.n_blocks     = w.n_blocks;
      cb.block_points = w.block_points;
      cb.n_points     = w.n_points;
      long long *ts = fl.ts + (size_t)agent * FL_TS;
      if (lane == 0) ts[8] = wall_clock64();
      // admission, in ticket order: at most n_admit maps under construction, and no faster than one agent per pace_ticks
      // — agents then leave the map stage (and reach every later stage) at a steady rate instead of in a burst, which is
      // what lets kernels with FIXED compute units all be busy at once: a swarm that moves in step serves one stage at a
      // time and the tick becomes the SUM of the stages' times (measured: 12.9 ms = 5.6 map + 3.7 corridors + 3.4 QP).
      if (!urgent) {  // my turn: a tight poll — a handful of waves wait here, and the hand-over from head to head is the admission rate
        const long long w0 = wall_clock64();
        bool            bad = false;
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&fl.hdr[FL_ADMITTED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < t) {
          __builtin_amdgcn_s_sleep(8);
          if (wall_clock64() - w0 > FLOW_TIMEOUT_TICKS ||
              __builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) {
            if (lane == 0) atomicCAS(err, 0, 16);
            bad = true;
            break;
          }
        }
        if (bad) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      if (!urgent && flow_wait_count(&fl.hdr[FL_MAPS_DONE], t - d.n_admit + 1, err)) break;
      if (!urgent && lane == 0) {
        long long *pc = reinterpret_cast<long long *>(&fl.hdr[FL_PACE_CLOCK]);
        while (wall_clock64() - *pc < d.pace_ticks) __builtin_amdgcn_s_sleep(16);
        *pc = wall_clock64();
        __hip_atomic_store(&fl.hdr[FL_ADMITTED], t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) ts[9] = wall_clock64();
      constexpr int W   = (int)(sizeof(SogmTrajRecord) / 16);
      const uint4  *src = reinterpret_cast<const uint4 *>(d.own + agent);
      uint4        *dst = reinterpret_cast<uint4 *>(&s_rec);
      for (int q = lane; q < W; q += 64) dst[q] = src[q];
      if (lane < 9) s_hov[lane] = d.hover[agent * 9 + lane];
      __syncthreads();
      const double stamp = d.t0 + k * d.period;
      if (lane == 0) {
        tick_inputs_agent(s_rec, s_hov, agent, stamp, d.start_offset, d.hover, d.now, d.t_start, d.pva, d.poses);
        d.stamps[agent] = stamp;
      }
      __threadfence();
      __syncthreads();
      cull_agent(g, w.cyl, w.n_cyl, d.poses[agent * 3], d.poses[agent * 3 + 1],
                 (CylCand *)d.cand + (size_t)agent * SOGM_MAX_CYL_LDS, d.n_cand + agent, lane);
      cull_blocks_agent(g, cb, agent, d.poses[agent * 3], d.poses[agent * 3 + 1], lane);
      __threadfence();
      if (lane == 0) {  // the agent's grid may be reset and its occupancy bits set (neither touches what the other writes)
        ts[12] = wall_clock64();
        atomicExch(&fl.stage[agent], 1);  // (the agent's previous map is complete: nobody else touches the counter now)
        wq_push(wq, wq_tail, ((unsigned)WK_MAP_RESET << 28) | (unsigned)agent, n_r);
        wq_push(wq, wq_tail, ((unsigned)WK_MAP_BITS << 28) | (unsigned)agent, n_b);
      }
      __syncthreads();
    }
    return;
  }
  // ---- workers.  Every worker holds a ticket of the plain queue, as before; the first n_uwork_wgs of them look at the urgent
  // queue first — before they take a plain descriptor and while they wait for one — and claim an urgent descriptor that is
  // THERE with a compare-and-swap on the queue's head (never a ticket for one that is not: a worker must not be lost to the
  // plain lane waiting for urgent work; the claim is tried once per look, so the waves do not spin on the counter).  A
  // first version gave the urgent lane 128 workers of its own: they idled most of the time, which a flight whose map
  // kernel is the bottleneck paid for (300^3 x 30: 11.5 -> 13.7 ms per tick).  A map stays in the lane its head put it in:
  // the next phase's descriptors go into the queue the finished one came from. ----
  const bool ulane = (int)blockIdx.x - d.n_head_wgs < d.n_uwork_wgs;
  WqWorker   ww;
  long long c1_prev = 0;
  int       kind_prev = 0;
  for (;;) {
    const long long c0     = wall_clock64();
    bool            urgent = false;
    // (a worker that also serves the urgent lane naps 14 ... 28 us between two looks, the others up to 110)
    const int desc = wq_take2(fl.mw, &fl.hdr[FL_MW_HEAD], fl.uw, &fl.hdr[FL_UW_TAIL], &fl.hdr[FL_UW_HEAD], ww, ulane, ulane ? 1 : 7,
                              err, &fl.hdr[FL_END], fl.epoch, urgent);
    if (desc < 0) break;
    const int n_r = urgent ? d.un_reset : d.n_reset, n_b = urgent ? d.un_bits : d.n_bits;
    const int n_m = urgent ? d.un_marks : d.n_marks, n_s = urgent ? d.un_splat : d.n_splat;
    const int S   = 1 + n_r + n_b + n_m + n_s;  // stage counts per (agent, tick); the agent's stage counter restarts at its head
    unsigned long long *const wq      = urgent ? fl.uw : fl.mw;
    int *const                wq_tail = &fl.hdr[urgent ? FL_UW_TAIL : FL_MW_TAIL];
    __threadfence();
    const long long c1 = wall_clock64();
    if (lane == 0 && c1_prev) {
      atomicAdd(&fl.prof[kind_prev], (unsigned long long)(c0 - c1_prev));  // the previous descriptor's work
      atomicAdd(&fl.prof[8 + kind_prev], 1ull);
    }
    if (lane == 0) atomicAdd(&fl.prof[0], (unsigned long long)(c1 - c0));
    const int          kind = desc >> 28, sub = (desc >> 16) & 0xFFF, agent = desc & 0xFFFF;
    c1_prev   = c1;
    kind_prev = kind;
    const int          k = fl.tick_of[agent], kl = k - fl.first_tick;
    const FlightWorld &w = d.worlds[kl];
    CloudBlocks        cb = d.cb;
    cb.bounds       = w.bounds;
    cb.n_blocks     = w.n_blocks;
    cb.block_points = w.block_points;
    cb.n_points     = w.n_points;
    long long     *ts  = fl.ts + (size_t)agent * FL_TS;
    const unsigned adr = (unsigned)agent;
    if (kind == WK_MAP_RESET || kind == WK_MAP_BITS) {
      if (kind == WK_MAP_RESET) {
        const unsigned n = d.lg.n[agent];
        if (sub == 0 && lane == 0 && reset_stat) {
          atomicAdd(reset_stat, (unsigned long long)(n > (unsigned)d.lg.cap ? (unsigned)d.lg.cap : n));
          if (agent == 0) atomicAdd(reset_stat + 1, 1ull);
        }
        flight_reset_ticket(reinterpret_cast<char *>(d.grid) + (size_t)agent * d.agent_bytes, d.agent_bytes,
                            d.lg.entries + (size_t)agent * d.lg.cap, n, d.lg.cap, sub, n_r, lane, reset_stat);
      } else {
        stamp_bits_blocks(g, w.cloud, cb, agent, sub, n_b, d.poses[agent * 3], d.poses[agent * 3 + 1],
                          d.poses[agent * 3 + 2], d.bits + (size_t)agent * d.words, lane);
      }
      __threadfence();
      if (lane == 0 && atomicAdd(&fl.stage[agent], 1) + 1 == 1 + n_r + n_b) {
        // the grid is clean and the bits are set: the log restarts, then the marks may append
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        d.lg.n[agent] = 0u;
        ts[13]        = wall_clock64();
        wq_push(wq, wq_tail, ((unsigned)WK_MAP_MARKS << 28) | adr, n_m);
      }
    } else if (kind == WK_MAP_MARKS) {
      stamp_marks_trips<true>(g, d.grid, d.bits, d.words, w.cyl, w.n_cyl, d.poses, (const CylCand *)d.cand, d.n_cand, agent, d.lg,
                        sub * 256, n_m * 256, s_cand, PRESTAMP_CAND_LDS);
      __syncthreads();  // (the next descriptor's staging overwrites s_cand)
      __threadfence();
      if (lane == 0 && atomicAdd(&fl.stage[agent], 1) + 1 == 1 + n_r + n_b + n_m) {
        const long long now = wall_clock64();
        ts[10] = now;
        if (!urgent) atomicAdd(&fl.hdr[FL_MAPS_DONE], 1);  // one more agent may be admitted (a map at the gate holds no worker)
        // the gate of the staleness rule: the overlay reads table ver(k - 2) — every agent must have finished tick k - 2.
        // Early: the overlay is parked, and the finish that completes that tick queues it (k_flight_light).
        const bool open = fl_gate_open(fl, kl);
        if (open) {
          ts[14] = now;
          wq_push(wq, wq_tail, ((unsigned)WK_MAP_SPLAT << 28) | adr, n_s);
        } else {
          int      *lst  = fl.parked + (size_t)kl * A;
          const int slot = atomicAdd(&fl.parked_n[kl], 1);
          __hip_atomic_store(&lst[slot], agent, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          __threadfence();
          // (the releaser may have scanned the list before this slot was written)
          if (fl_gate_open(fl, kl) && atomicCAS(&lst[slot], agent, -2) == agent) {
            ts[14] = wall_clock64();
            wq_push(wq, wq_tail, ((unsigned)WK_MAP_SPLAT << 28) | adr, n_s);
          }
        }
      }
    } else {  // WK_MAP_SPLAT: the neighbours' records of table ver(k - 2)
      if (d.tables && d.n_total > 0) {
        const SogmTrajRecord *tab = d.tables + (size_t)((k - fl.lag) & 3) * d.n_total;
        const int             n   = d.n_total * g.T;
        for (int i = sub * 64 + lane; i < n; i += n_s * 64)
          splat_item(g, d.grid, tab[i / g.T], agent, i % g.T, d.ego_ids, d.poses, d.stamps, d.body, d.n_body, d.lg);
      }
      __threadfence();
      if (lane == 0 && atomicAdd(&fl.stage[agent], 1) + 1 == S) {  // the agent's map of tick k is complete
        ts[11] = wall_clock64();
        fl_publish(fl.s_ring, fl.ring_mask, &fl.hdr[FL_S_READY], agent);
      }
    }
  }
}
int launch_flight_map(const GridGeom &g, const FlightCtl &fl, const FlightMapDev &d, int n_workgroups, hipStream_t st) {
  hipLaunchKernelGGL(k_flight_map, dim3(n_workgroups), dim3(64), 0, st, g, fl, d, d.reset_stat);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_prestamp_flow(const GridGeom &g, const FlowCtl &fc, const PrestampDev &ps, int n_workgroups, int n_qp,
                         int n_finish, hipStream_t st) {
  hipLaunchKernelGGL(k_prestamp_gate, dim3(1), dim3(64), 0, st, fc.hdr, ps.gate_agents, n_qp, n_finish);
  hipLaunchKernelGGL(k_prestamp_flow, dim3(n_workgroups), dim3(64), 0, st, g, fc, ps);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// End of a tick: latest-wins per drone (particles.cpp:179-190) — a successful replan replaces the agent's record,
// a failed one keeps the trajectory being executed (plan_manager.cpp:176-196); `all` (optional) is the swarm table of
// a single-process run, refreshed in the same pass.  One lane per 16 bytes of a record.
__global__ __launch_bounds__(256) void k_merge_latest(const SogmTrajRecord *__restrict__ fresh,
                                                      const int32_t *__restrict__ ok, SogmTrajRecord *__restrict__ own,
                                                      SogmTrajRecord *__restrict__ all, int n) {
  constexpr int W = (int)(sizeof(SogmTrajRecord) / 16);
  static_assert(sizeof(SogmTrajRecord) % 16 == 0, "record size");
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)n * W) return;
  const int a = (int)(gid / W), w = (int)(gid % W);
  // all three loads are issued together: this kernel often runs beside the tail of the streaming clear, where a
  // dependent global round trip costs a millisecond
  const int   good = ok[a];
  const uint4 vf = reinterpret_cast<const uint4 *>(fresh + a)[w], vo = reinterpret_cast<const uint4 *>(own + a)[w];
  const uint4 v  = good ? vf : vo;
  if (good) reinterpret_cast<uint4 *>(own + a)[w] = v;
  if (all) reinterpret_cast<uint4 *>(all + a)[w] = v;
}

// BaselinePlanner::isTrajSafe (plan_manager/src/baseline.cpp:45-68): sample the executed trajectory every
// 0.1 s from "now" up to min(T, duration) and query the SOGM at the sample's time after the map stamp.
__global__ __launch_bounds__(64) void k_traj_safe(MapView m, const SogmTrajRecord *__restrict__ rec,
                                                  const double *__restrict__ t_now, double T,
                                                  int32_t *__restrict__ out) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= m.n_agents) return;
  const SogmTrajRecord &r = rec[a];
  if (r.n_pieces <= 0) {  // nothing is being executed
    out[a] = 1;
    return;
  }
  double t0 = t_now[a] - r.time_start;
  if (t0 < 0) t0 = 0;
  if (t0 > T) {
    out[a] = 1;
    return;
  }
  double dur = 0;
  for (int k = 0; k < r.n_pieces; ++k) dur += r.duration[k];
  T = T > dur ? dur : T;
  int safe = 1;
  for (double t = t0; t < T; t += 0.1) {
    double p[3];
    bezier_pos(r, t, p);
    const double dt = t + r.time_start - m.stamps[a];
    if (query_clear_time(m, a, p[0], p[1], p[2], dt) == 1) {
      safe = 0;
      break;
    }
  }
  out[a] = safe;
}

static int launch_clear_impl(sogm_ctx *c, hipStream_t st, float *grid, bool polite, int part, size_t split);
// ---- sparse reset: logs per pool slot --------------------------------------------------------------------
static int slot_of_grid(const sogm_ctx *c, const float *grid) {
  if (c->n_pool == 0) return 0;
  for (int i = 0; i < c->n_pool; ++i)
    if (c->pool[i] == grid) return i;
  return -1;
}
// the log of a slot, allocated on first use; {nullptr, ...} when the feature is off or there is no room for it (the
// slot then stays untracked and is cleared densely)
MarkLog mark_log(sogm_ctx *c, int slot) {
  MarkLog none{nullptr, nullptr, 0, nullptr};
  if (!c->sparse || slot < 0 || slot > 2) return none;
  if (!c->d_log[slot]) {
    unsigned *e = nullptr, *n = nullptr;
    // (The counters' first zeroing is COMPLETE when this returns: the memset is a null-stream operation, which the
    //  library's non-blocking streams do not wait for — with a second context busy on the device it was seen to run
    //  after the first stamp had appended its entries, i.e. it threw them away, and the slot's first reset through
    //  its log left that stamp's marks in the grid.  Only the null stream is synchronised: persistent kernels of a
    //  replan in flight on other streams are not waited for.)
    if (hipMalloc((void **)&e, sizeof(unsigned) * (size_t)c->log_cap * c->n_agents) != hipSuccess ||
        hipMalloc((void **)&n, sizeof(unsigned) * (size_t)c->n_agents) != hipSuccess ||
        hipMemset(n, 0, sizeof(unsigned) * (size_t)c->n_agents) != hipSuccess ||
        hipStreamSynchronize(nullptr) != hipSuccess) {
      (void)hipGetLastError();
      if (e) (void)hipFree(e);
      if (n) (void)hipFree(n);
      c->sparse = 0;  // no room: dense clears from here on
      for (int i = 0; i < 3; ++i) c->tracked[i] = 0;
      return none;
    }
    if (!c->d_reset_stat &&
        (hipMalloc((void **)&c->d_reset_stat, 8 * sizeof(unsigned long long)) != hipSuccess ||
         hipMemset(c->d_reset_stat, 0, 8 * sizeof(unsigned long long)) != hipSuccess ||
         hipStreamSynchronize(nullptr) != hipSuccess)) {
      (void)hipGetLastError();
      (void)hipFree(e);
      (void)hipFree(n);
      c->sparse = 0;
      for (int i = 0; i < 3; ++i) c->tracked[i] = 0;
      return none;
    }
    c->d_log[slot]   = e;
    c->d_log_n[slot] = n;
    c->tracked[slot] = 0;  // what the grid holds now was written without a log
  }
  return MarkLog{c->d_log[slot], c->d_log_n[slot], c->log_cap, c->d_reset_stat ? c->d_reset_stat + 4 : nullptr};
}
static size_t agent_grid_bytes(const sogm_ctx *c) { return (size_t)c->spec.T * (size_t)c->geom.V * c->cell_bytes(); }

int reset_slot(sogm_ctx *c, hipStream_t st, int slot, float *grid, bool polite) {
  const MarkLog lg = mark_log(c, slot);
  if (lg.entries && c->tracked[slot]) {
    const int wgs = c->tune_i(SOGM_TUNE_RESET_WGS) > 0 ? c->tune_i(SOGM_TUNE_RESET_WGS) : 32;  // workgroups per agent
    // under the replan (polite: beside the QP stage, few CUs free) two lanes and 32-byte lines are faster - 1.05 ms
    // against 1.17 for the 64-byte lines, half the write traffic; with the machine to itself (reset in the update's
    // own stream) four lanes x eight entries per trip - 0.75 ms against 0.91.  Both switches are tuning aids.
    const int lanes_env = c->tune_i(SOGM_TUNE_RESET_LANES), unroll_env = c->tune_i(SOGM_TUNE_RESET_UNROLL);
    const int lanes  = lanes_env == 2 || lanes_env == 4 ? lanes_env : polite ? 2 : 4;
    const int unroll = unroll_env == 1 || unroll_env == 8 ? unroll_env : polite ? 1 : 8;
    prof_begin(c, SOGM_PROF_CLEAR, st);
    auto *kern = lanes == 4 ? (unroll == 1 ? k_reset_sectors<4, 1> : k_reset_sectors<4, 8>)
                            : (unroll == 1 ? k_reset_sectors<2, 1> : k_reset_sectors<2, 8>);
    hipLaunchKernelGGL(kern, dim3(wgs, c->n_agents), dim3(256), 0, st, reinterpret_cast<char *>(grid),
                       agent_grid_bytes(c), lg.entries, lg.n, lg.cap, c->d_reset_stat);
    prof_end(c, SOGM_PROF_CLEAR, st);
    SOGM_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_zero_words, dim3((c->n_agents + 255) / 256), dim3(256), 0, st, lg.n, c->n_agents);
    SOGM_HIP_CHECK(hipGetLastError());
    c->hist_sparse[slot]++;
    return SOGM_OK;
  }
  return launch_clear(c, st, grid, polite);  // (a complete dense clear restarts the slot's log, see launch_clear)
}

static int stamp_scratch(sogm_ctx *c, hipStream_t st, int *words_out) {
  const int A = c->n_agents;
  if (!c->d_cand) {
    SOGM_HIP_CHECK(hipMalloc(&c->d_cand, sizeof(CylCand) * SOGM_MAX_CYL_LDS * (size_t)A));
    SOGM_HIP_CHECK(hipMalloc(&c->d_ncand, sizeof(int) * (size_t)A));
  }
  const int words = (((c->geom.V + 31) / 32) + 255) & ~255;  // k_stamp_marks reads 256 words per trip
  if (!c->d_stamp_bits) {
    SOGM_HIP_CHECK(hipMalloc((void **)&c->d_stamp_bits, sizeof(unsigned) * (size_t)words * A));
    SOGM_HIP_CHECK(hipMemsetAsync(c->d_stamp_bits, 0, sizeof(unsigned) * (size_t)words * A, st));
  }
  *words_out = words;
  return SOGM_OK;
}
int prestamp_buffers(sogm_ctx *c, PrestampDev *d) {
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  if (!c->d_poses_next) {
    SOGM_HIP_CHECK(hipMalloc(&c->d_poses_next, sizeof(float) * 3 * (size_t)c->n_agents));
    SOGM_HIP_CHECK(hipMalloc(&c->d_stamps_next, sizeof(double) * (size_t)c->n_agents));
  }
  int words = 0;
  if (int rc = stamp_scratch(c, nullptr, &words)) return rc;
  d->bits   = c->d_stamp_bits;
  d->words  = words;
  d->cand   = c->d_cand;
  d->n_cand = c->d_ncand;
  d->poses  = c->d_poses_next;
  d->stamps = c->d_stamps_next;
  return SOGM_OK;
}

// Growing stalls the host for as long as the device needs to drain (hipDeviceSynchronize + hipFree + hipMalloc): callers
// reach it before they launch anything of their tick — sogm_update_world / sogm_update_gt_swarm at their top,
// sogm_planner_set_prestamp (so that sogm_replan's own call below never grows), sogm_flight_run before its first launch.
int world_blocks(sogm_ctx *c, const SogmWorld *w, CloudBlocks *out) {
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  if (w->n_blocks > c->blk_cap || !c->d_blk_n) {  // (also the first frame of all, even an empty one: the counts exist)
    // (grown between ticks only: a frame with more blocks than any before; the old lists may still be read by a
    //  pre-stamp in flight, so everything drains first)
    SOGM_HIP_CHECK(hipDeviceSynchronize());
    if (c->d_blk_list) (void)hipFree(c->d_blk_list);
    c->d_blk_list = nullptr;
    const int cap = (w->n_blocks + 1023) & ~1023;
    SOGM_HIP_CHECK(hipMalloc((void **)&c->d_blk_list, sizeof(int) * (size_t)cap * (size_t)c->n_agents));
    if (!c->d_blk_n) SOGM_HIP_CHECK(hipMalloc((void **)&c->d_blk_n, sizeof(int) * (size_t)c->n_agents));
    c->blk_cap = cap;
  }
  // (rows of blk_cap entries whatever the frame holds: a row never exceeds n_blocks <= blk_cap)
  *out = CloudBlocks{w->block_bounds, w->n_blocks, w->block_points, w->n_points, c->d_blk_list, c->d_blk_n, c->blk_cap};
  return SOGM_OK;
}

// a new tick: wide clear workgroups opened for the replan that just ended retire (the stamp, the searches and the
// corridor stage want the memory pipeline responsive), the narrow launch goes on.  (Only when a dense clear was
// queued since the last update: sparse resets have no wide launch.)  Behind the side stream's work of that replan: its
// gate kernels compare the word with their epoch.
int retire_wide_clear(sogm_ctx *c, hipStream_t st) {
  if (c->overlap >= 2 && c->clear_gate && c->wide_clear_pending) {
    hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, st, c->clear_epoch_word, 0);
    SOGM_HIP_CHECK(hipGetLastError());
    c->wide_clear_pending = 0;
  }
  return SOGM_OK;
}

int adopt_preclear(sogm_ctx *c, hipStream_t st, bool join) {
  if (join)  // (sogm_update_prestamped joins behind its overlay instead, and retires the wide clear there)
    if (int rc = join_prestamp(c, st)) return rc;
  if (!c->precleared) return SOGM_OK;
  if (c->overlap >= 2) {
    // rotate: the front of the ready queue becomes the current grid, the old current grid is dirty
    const int nxt = c->ready[0];
    for (int i = 1; i < c->n_ready; ++i) c->ready[i - 1] = c->ready[i];
    c->n_ready--;
    c->dirty[c->n_dirty++] = c->cur_idx;
    c->cur_idx             = nxt;
    c->d_grid              = c->pool[nxt];
    SOGM_HIP_CHECK(hipStreamWaitEvent(st, c->pool_ev[nxt], 0));
    c->precleared = c->n_ready > 0;
    const int early = c->tune_i(SOGM_TUNE_CLEAR_EARLY);
    if (join)
      if (int rc = retire_wide_clear(c, st)) return rc;
    if (c->clear_gate && early) {
      // tuning aid (clear_early = 1): queue the clear of the swapped-out grid NOW, under the stamp — its readers,
      // the previous replan's kernels, are complete on `st` in stream order — for the replan that will announce
      // epoch clear_epoch + 1.  Measured: the stamp beside it takes twice as long (1.4 -> 3.1 ms) and the first
      // ticks of a flight lose 5 %; later ticks gain 3 %.  Off by default.
      c->clear_epoch_ahead = 1;
      SOGM_HIP_CHECK(hipEventRecord(c->ev_grid_free, st));
      const int rc         = queue_spare_clears(c, c->ev_grid_free);
      c->clear_epoch_ahead = 0;
      return rc;
    }
    return SOGM_OK;
  }
  SOGM_HIP_CHECK(hipStreamWaitEvent(st, c->ev_cleared, 0));
  c->precleared = 0;
  return SOGM_OK;
}

// modes 2 / 3: queue the clear of every dirty spare grid on the side stream once `after` has fired (every reader
// of those grids is ordered before it); sogm_replan calls this right after its fan-out event
int queue_spare_clears(sogm_ctx *c, hipEvent_t after) {
  if (c->overlap < 2 || c->n_dirty == 0) return SOGM_OK;
  SOGM_HIP_CHECK(hipStreamWaitEvent(c->side, after, 0));
  // The clear shares the machine with the whole replan, as ONE narrow launch by default.  clear_head_gb (a
  // tuning aid) splits it into a narrow head of that many GB and a full-width rest: measured with the dataflow
  // replan (profiles/r02_*), a wide rest shortens the clear (15.9 -> 14.3 ms) but costs the planner kernels more
  // than it saves (tick 18.2 -> 19.3 ms), because per-agent chaining spreads the global-memory phases (searches,
  // point scans, FIRI set-up of late agents) over the whole tick.
  const double head_gb = c->tune[SOGM_TUNE_CLEAR_HEAD_GB] >= 0 ? c->tune[SOGM_TUNE_CLEAR_HEAD_GB] : 1.0e9;
  const size_t total = clear_vec4_total(c);
  size_t       head  = (size_t)(head_gb * 1e9 / 16.0);
  if (head > total) head = total;
  for (int i = 0; i < c->n_dirty; ++i) {
    const int g  = c->dirty[i];
    int       rc = SOGM_OK;
    if (c->sparse && c->tracked[g] && mark_log(c, g).entries) {
      // The reset is held back until every agent's corridors are final (the gate the dense clear's wide launch
      // uses): beside the searches and the corridor stage's point scans its 3 GB of scattered stores cost the
      // chain 0.8 ms (tick 13.8 -> 12.9 ms); under the QP stage, which lives in LDS, they cost nothing and the
      // reset itself takes 1.3 instead of 2.4 ms.  reset_late = 0: start it with the replan.
      const int late = c->tune_i(SOGM_TUNE_RESET_LATE);
      if (late && c->clear_gate) {
        if (c->gate_frac_agents > 0 && c->gate_frac_agents < c->clear_gate_target && !c->gate_frac_valid) {
          // the pre-stamp may start when this share of the agents' corridors is final (tuning key prestamp_gate_frac):
          // the same gate kernel with a lower target, on THIS stream (which spins for the full gate anyway), and an
          // event for the pre-stamp's stream — no spinning kernel at the head of a second stream
          hipLaunchKernelGGL(k_clear_gate, dim3(1), dim3(64), 0, c->side, c->clear_cursor, ~(size_t)0, c->clear_gate,
                             c->clear_gate_err, c->gate_frac_agents, c->clear_epoch_word, c->clear_epoch);
          SOGM_HIP_CHECK(hipGetLastError());
          SOGM_HIP_CHECK(hipEventRecord(c->ev_gate_frac, c->side));
          c->gate_frac_valid = 1;
        }
        hipLaunchKernelGGL(k_clear_gate, dim3(1), dim3(64), 0, c->side, c->clear_cursor, ~(size_t)0, c->clear_gate,
                           c->clear_gate_err, c->clear_gate_target, c->clear_epoch_word, c->clear_epoch);
        SOGM_HIP_CHECK(hipGetLastError());
        // "every agent's corridors are final" as an EVENT for the pre-stamp's stream (sogm_replan): no second gate kernel
        // spinning at the head of a stream (with shared or oversubscribed hardware queues every spinner is a hazard)
        SOGM_HIP_CHECK(hipEventRecord(c->ev_gate_open, c->side));
        c->gate_open_valid = 1;
      }
      rc = reset_slot(c, c->side, g, c->pool[g], true);  // the logged sectors only: a fraction of a millisecond
    } else if (head == 0) {
      rc = launch_clear(c, c->side, c->pool[g], false);
    } else if (head >= total) {
      rc = launch_clear(c, c->side, c->pool[g], true);
    } else {
      rc = launch_clear(c, c->side, c->pool[g], true, 1, head);
      if (!rc) rc = launch_clear(c, c->side, c->pool[g], false, 2, head);
    }
    if (rc) return rc;
    SOGM_HIP_CHECK(hipEventRecord(c->pool_ev[g], c->side));
    c->ready[c->n_ready++] = g;
  }
  c->n_dirty    = 0;
  c->precleared = 1;
  return SOGM_OK;
}

// polite = the clear shares the machine with latency-bound kernels that read global memory (double-buffered
// mode): a full-width clear (2048 persistent workgroups, unbounded stores in flight) starves every other
// kernel's loads for its whole duration; 64 workgroups with <= 4 stores in flight per wave still stream at
// ~5.7 TB/s and leave the memory pipeline responsive.  SOGM_CLEAR_WGS / SOGM_CLEAR_THROTTLE / SOGM_CLEAR_NT
// override the choice (tuning aids: clear_wgs / clear_throttle / clear_nt).
size_t clear_vec4_total(const sogm_ctx *c) {
  return (size_t)c->n_agents * c->spec.T * (size_t)c->geom.V * c->cell_bytes() / 4 / 4;
}
// part: 0 = the whole grid, 1 = the first `split` 16-byte elements, 2 = everything from `split` on
int launch_clear(sogm_ctx *c, hipStream_t st, float *grid, bool polite, int part, size_t split) {
  if (!grid) grid = c->d_grid;
  const int rc = launch_clear_impl(c, st, grid, polite, part, split);
  if (rc == SOGM_OK && part != 1) {
    // this launch completes a dense clear of the slot: behind it (stream order) the slot's mark log starts empty and
    // covers every non-zero cell again
    const int     slot = slot_of_grid(c, grid);
    const MarkLog lg   = mark_log(c, slot);
    if (slot >= 0 && slot < 3) c->hist_dense[slot]++;
    if (lg.entries)
    {
      hipLaunchKernelGGL(k_zero_words, dim3((c->n_agents + 255) / 256), dim3(256), 0, st, lg.n, c->n_agents);
      c->tracked[slot] = hipGetLastError() == hipSuccess ? 1 : 0;
    }
  }
  return rc;
}
static int launch_clear_impl(sogm_ctx *c, hipStream_t st, float *grid, bool polite, int part, size_t split) {
  // the clear is a byte stream: n = number of 4-byte words of the grid (fp16 grids: 2 cells per word)
  // (rounded up: an odd number of fp16 cells ends in half a word; allocations are padded to 16 B)
  const size_t n     = ((size_t)c->n_agents * c->spec.T * (size_t)c->geom.V * c->cell_bytes() + 3) / 4;
  const size_t nall  = n / 4;
  const size_t first = part == 2 ? split : 0;
  const size_t nv4   = part == 1 ? split : nall - first;
  const int    tail  = part == 1 ? 0 : (int)(n - nall * 4);
  const int    slot  = part == 1 ? SOGM_PROF_CLEAR_HEAD : SOGM_PROF_CLEAR;
  size_t       want  = (nv4 + 255) / 256;
  const int env_wgs = c->tune_i(SOGM_TUNE_CLEAR_WGS) > 0 ? c->tune_i(SOGM_TUNE_CLEAR_WGS) : 0;
  const int env_throttle = c->tune_i(SOGM_TUNE_CLEAR_THROTTLE), nt = c->tune_i(SOGM_TUNE_CLEAR_NT) != 0;
  const size_t max_wgs  = env_wgs ? (size_t)env_wgs : (polite ? (c->clear_gate && part == 0 ? 80 : 64) : 2048);
  const int    throttle = env_wgs ? env_throttle : (polite ? 4 : 0);
  const int    nblk     = (int)(want < 1 ? 1 : (want > max_wgs ? max_wgs : want));
  // clear_wide_wgs = 0 switches the adaptive width off; clear_wide_bound: stores in flight per wave of the wide launch
  const int wide_wgs = c->tune_i(SOGM_TUNE_CLEAR_WIDE_WGS), wide_bound = c->tune_i(SOGM_TUNE_CLEAR_WIDE_BOUND);
  if (polite && part == 0 && c->clear_gate && c->clear_cursor && c->side2 && wide_wgs > 0 && nt) {
    const size_t        nchunks = (nall + CLEAR_CHUNK_V4 - 1) / CLEAR_CHUNK_V4;
    unsigned long long *cur = c->clear_cursor + (c->clear_seq & 1), *nxt = c->clear_cursor + ((c->clear_seq + 1) & 1);
    ++c->clear_seq;
    SOGM_HIP_CHECK(hipEventRecord(c->ev_side2_go, st));
    SOGM_HIP_CHECK(hipStreamWaitEvent(c->side2, c->ev_side2_go, 0));
    prof_begin(c, slot, st);
    const int epoch = c->clear_epoch + c->clear_epoch_ahead;
    c->wide_clear_pending = 1;
    hipLaunchKernelGGL(k_clear_chunks<true>, dim3(nblk), dim3(256), 0, st, (vfloat4 *)grid, nall, grid + nall * 4,
                       tail, cur, nxt, c->clear_epoch_word, epoch, 0);
    hipLaunchKernelGGL(k_clear_gate, dim3(1), dim3(64), 0, c->side2, cur, nchunks, c->clear_gate,
                       c->clear_gate_err, c->clear_gate_target, c->clear_epoch_word, epoch);
    hipLaunchKernelGGL(k_clear_chunks<false>, dim3(wide_wgs), dim3(256), 0, c->side2, (vfloat4 *)grid, nall,
                       grid + nall * 4, 0, cur, nxt, c->clear_epoch_word, epoch, wide_bound);
    SOGM_HIP_CHECK(hipEventRecord(c->ev_side2_done, c->side2));
    SOGM_HIP_CHECK(hipStreamWaitEvent(st, c->ev_side2_done, 0));  // the clear is complete when both launches are
    prof_end(c, slot, st);
    SOGM_HIP_CHECK(hipGetLastError());
    return SOGM_OK;
  }
  prof_begin(c, slot, st);
  if (nt)
    hipLaunchKernelGGL(k_clear_slabs<true>, dim3(nblk), dim3(256), 0, st, (vfloat4 *)grid + first, nv4,
                       grid + nall * 4, tail, throttle);
  else
    hipLaunchKernelGGL(k_clear_slabs<false>, dim3(nblk), dim3(256), 0, st, (vfloat4 *)grid + first, nv4,
                       grid + nall * 4, tail, throttle);
  prof_end(c, slot, st);
  SOGM_HIP_CHECK(hipGetLastError());
  return SOGM_OK;
}

int next_clear_epoch(sogm_ctx *c) {
  if (++c->clear_epoch <= 0) c->clear_epoch = 1;  // 0 = "no replan in flight"
  return c->clear_epoch;
}
int announce_clear_epoch(sogm_ctx *c, hipStream_t st) {
  next_clear_epoch(c);
  hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, st, c->clear_epoch_word, c->clear_epoch);
  SOGM_HIP_CHECK(hipGetLastError());
  return SOGM_OK;
}

}  // namespace sogm

using namespace sogm;

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int sogm_abi_version(void) { return SOGM_ABI_VERSION; }
const char *sogm_last_error(void) { return sogm::g_err; }

int sogm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int sogm_create(const SogmSpec *spec, int n_agents, int device, sogm_ctx **out) {
  if (!spec || !out || n_agents <= 0) return SOGM_ERR_INVALID_ARG;
  if (spec->L <= 0 || spec->W <= 0 || spec->H <= 0 || spec->T <= 0 || spec->T > 32 ||
      !(spec->resolution > 0.f) || !(spec->time_resolution > 0.f))
    return SOGM_ERR_INVALID_ARG;
  if (spec->map_kind != SOGM_MAP_FAKE && spec->map_kind != SOGM_MAP_RISKBASE &&
      spec->map_kind != SOGM_MAP_RISKVOXEL)
    return SOGM_ERR_INVALID_ARG;
  if ((spec->storage & ~(1 | SOGM_LAYOUT_TILED)) != 0) return SOGM_ERR_INVALID_ARG;
  if ((spec->storage & SOGM_LAYOUT_TILED) && ((spec->L | spec->W | spec->H) & 1)) return SOGM_ERR_INVALID_ARG;  // whole tiles
  // one time slice is addressed with 32-bit byte offsets (window_sum_hits) and V is an int
  if ((unsigned long long)spec->L * spec->W * spec->H >= (1ull << 30)) {
    sogm::set_error_text("sogm_create: L * W * H must stay below 2^30 cells per time slice");
    return SOGM_ERR_INVALID_ARG;
  }
  *out = nullptr;
  if (sogm_device_count() <= device || device < 0) {
    std::snprintf(sogm::g_err, sizeof(sogm::g_err), "no HIP device %d", device);
    return SOGM_ERR_NO_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess) return SOGM_ERR_NO_DEVICE;
  sogm_ctx *c = new (std::nothrow) sogm_ctx();
  if (!c) return SOGM_ERR_INVALID_ARG;
  std::memset(c, 0, sizeof(*c));
  c->spec          = *spec;
  c->geom          = make_geom(*spec);
  c->n_agents      = n_agents;
  c->device        = device;
  c->prestamp_slot = -1;
  {
#define X(id, name, dflt, lo, hi) c->tune[SOGM_TUNE_##id] = (double)(dflt);
    SOGM_TUNING_TABLE(X)
#undef X
    const char *e = getenv("SOGM_SPARSE_RESET");  // (one of the library's three environment switches)
    c->sparse     = e ? atoi(e) != 0 : 1;
    // default capacity per agent: one entry per 80 cells, at least 2^20 (the bench scenes log ~0.3 M entries per
    // agent and tick at 200^3 x 20); sogm_set_sparse_reset changes it
    const long long dflt = (long long)spec->L * spec->W * spec->H * spec->T / 80;
    c->log_cap    = (int)(dflt < (1 << 20) ? (1 << 20) : (dflt > (1 << 26) ? (1 << 26) : dflt));
  }
  const size_t n   = (size_t)n_agents * spec->T * (size_t)c->geom.V;
  hipError_t   e   = hipMalloc(&c->d_grid, (n * c->cell_bytes() + 15) & ~(size_t)15);
  if (e == hipSuccess) e = hipMalloc(&c->d_poses, sizeof(float) * 3 * n_agents);
  if (e == hipSuccess) e = hipMalloc(&c->d_stamps, sizeof(double) * n_agents);
  if (e == hipSuccess) e = hipMalloc(&c->d_scratch_vt, sizeof(float) * (size_t)c->geom.V * spec->T);
  if (e == hipSuccess) e = hipMemset(c->d_poses, 0, sizeof(float) * 3 * n_agents);
  if (e == hipSuccess) e = hipMemset(c->d_stamps, 0, sizeof(double) * n_agents);
  if (e == hipSuccess) e = hipMalloc((void **)&c->clear_cursor, 2 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(c->clear_cursor, 0, 2 * sizeof(unsigned long long));
  if (e == hipSuccess) e = sogm::create_stream_partitioned(&c->side, 0);
  if (e == hipSuccess) e = sogm::create_stream_partitioned(&c->side2, 0);
  if (e == hipSuccess) e = sogm::create_stream_partitioned(&c->pstream, 0);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_side2_go, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_side2_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_grid_free, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_cleared, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_gate_open, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_gate_frac, hipEventDisableTiming);
  // the memsets above are null-stream operations, which the non-blocking streams every later call uses do not wait for
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  if (e != hipSuccess) {
    sogm::set_error("sogm_create", e);
    sogm_destroy(c);
    return SOGM_ERR_HIP;
  }
  *out = c;
  return SOGM_OK;
}

void sogm_destroy(sogm_ctx *c) {
  if (!c) return;
  if (c->n_pool == 0 && c->d_grid) (void)hipFree(c->d_grid);
  for (int i = 0; i < c->n_pool; ++i)
    if (c->pool[i]) (void)hipFree(c->pool[i]);
  for (int i = 0; i < 3; ++i)
    if (c->pool_ev[i]) (void)hipEventDestroy(c->pool_ev[i]);
  for (int i = 0; i < 3; ++i) {
    if (c->d_log[i]) (void)hipFree(c->d_log[i]);
    if (c->d_log_n[i]) (void)hipFree(c->d_log_n[i]);
  }
  if (c->d_reset_stat) (void)hipFree(c->d_reset_stat);
  if (c->d_poses_next) (void)hipFree(c->d_poses_next);
  if (c->d_stamps_next) (void)hipFree(c->d_stamps_next);
  if (c->d_poses) (void)hipFree(c->d_poses);
  if (c->d_stamps) (void)hipFree(c->d_stamps);
  if (c->clear_cursor) (void)hipFree(c->clear_cursor);
  if (c->d_body) (void)hipFree(c->d_body);
  if (c->d_scratch_vt) (void)hipFree(c->d_scratch_vt);
  if (c->d_cand) (void)hipFree(c->d_cand);
  if (c->d_stamp_bits) (void)hipFree(c->d_stamp_bits);
  if (c->d_ncand) (void)hipFree(c->d_ncand);
  if (c->h_tick_clock) (void)hipHostFree(c->h_tick_clock);
  if (c->d_blk_list) (void)hipFree(c->d_blk_list);
  if (c->d_blk_n) (void)hipFree(c->d_blk_n);
  if (c->d_filter_cells) (void)hipFree(c->d_filter_cells);
  if (c->d_filter_box) (void)hipFree(c->d_filter_box);
  if (c->d_filter_blocks) (void)hipFree(c->d_filter_blocks);
  if (c->side) {
    (void)hipStreamSynchronize(c->side);
    (void)hipStreamDestroy(c->side);
  }
  if (c->side2) {
    (void)hipStreamSynchronize(c->side2);
    (void)hipStreamDestroy(c->side2);
  }
  if (c->pstream) {
    (void)hipStreamSynchronize(c->pstream);
    (void)hipStreamDestroy(c->pstream);
  }
  if (c->ustream) {
    (void)hipStreamSynchronize(c->ustream);
    (void)hipStreamDestroy(c->ustream);
  }
  if (c->ev_uin) (void)hipEventDestroy(c->ev_uin);
  if (c->ev_udone) (void)hipEventDestroy(c->ev_udone);
  if (c->d_map_ready) (void)hipFree(c->d_map_ready);
  if (c->d_update_ctl) (void)hipFree(c->d_update_ctl);
  if (c->d_update_order) (void)hipFree(c->d_update_order);
  if (c->d_update_ts) (void)hipFree(c->d_update_ts);
  if (c->ev_side2_go) (void)hipEventDestroy(c->ev_side2_go);
  if (c->ev_side2_done) (void)hipEventDestroy(c->ev_side2_done);
  if (c->ev_grid_free) (void)hipEventDestroy(c->ev_grid_free);
  if (c->ev_cleared) (void)hipEventDestroy(c->ev_cleared);
  if (c->ev_gate_open) (void)hipEventDestroy(c->ev_gate_open);
  if (c->ev_gate_frac) (void)hipEventDestroy(c->ev_gate_frac);
  if (c->xstream) {
    (void)hipStreamSynchronize(c->xstream);
    (void)hipStreamDestroy(c->xstream);
  }
  if (c->ev_xin) (void)hipEventDestroy(c->ev_xin);
  if (c->ev_xdone) (void)hipEventDestroy(c->ev_xdone);
  for (int k = 0; k < SOGM_PROF_N; ++k) {
    if (c->ring[k]) {
      for (int i = 0; i < 2 * SOGM_PROF_RING; ++i)
        if (c->ring[k][i]) (void)hipEventDestroy(c->ring[k][i]);
      delete[] c->ring[k];
    }
  }
  delete c;
}

int64_t sogm_grid_bytes(const sogm_ctx *c) {
  return c ? (int64_t)c->n_agents * c->spec.T * (int64_t)c->geom.V * (int64_t)c->cell_bytes() : 0;
}
float *sogm_grid_ptr(sogm_ctx *c) {
  if (!c) return nullptr;
  if (c->update_pending) {  // an update flow still building this grid: the pointer is handed out complete
    (void)hipStreamSynchronize(c->ustream);
    c->update_pending = 0;
  }
  c->tracked[sogm::cur_slot(c)] = 0;  // the caller may write cells the mark log does not see
  return c->d_grid;
}

int sogm_set_sparse_reset(sogm_ctx *c, int enable, int log_capacity) {
  if (!c || log_capacity < 0) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  SOGM_HIP_CHECK(hipDeviceSynchronize());  // resets / writers in flight use the logs
  for (int i = 0; i < 3; ++i) {
    if (c->d_log[i]) (void)hipFree(c->d_log[i]);
    if (c->d_log_n[i]) (void)hipFree(c->d_log_n[i]);
    c->d_log[i]   = nullptr;
    c->d_log_n[i] = nullptr;
    c->tracked[i] = 0;  // contents unknown to the (new) logs: each slot's next reset is dense
  }
  c->sparse = enable ? 1 : 0;
  if (log_capacity > 0) c->log_cap = log_capacity;
  return SOGM_OK;
}

int sogm_sparse_reset_state(sogm_ctx *c, int32_t *out) {
  if (!c || !out) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  const int slot = sogm::cur_slot(c);
  out[0] = c->sparse;
  out[1] = c->log_cap;
  out[2] = c->tracked[slot];
  out[3] = 0;  // largest per-agent entry count of the current grid's log
  out[4] = 0;  // entries of all agents (what the grid's next reset reads; capped at the capacity per agent)
  if (c->sparse && c->d_log_n[slot]) {
    std::vector<unsigned> n((size_t)c->n_agents);
    SOGM_HIP_CHECK(hipMemcpy(n.data(), c->d_log_n[slot], sizeof(unsigned) * n.size(), hipMemcpyDeviceToHost));
    unsigned           mx  = 0;
    unsigned long long tot = 0;
    for (unsigned v : n) {
      mx = v > mx ? v : mx;
      tot += v > (unsigned)c->log_cap ? (unsigned)c->log_cap : v;
    }
    out[3] = (int32_t)(mx > 0x7FFFFFFFu ? 0x7FFFFFFFu : mx);
    out[4] = (int32_t)(tot > 0x7FFFFFFFull ? 0x7FFFFFFFull : tot);
  }
  out[5] = out[6] = out[7] = 0;  // sparse resets since the previous call: launches, entries read and KiB zeroed per launch (means)
  if (c->d_reset_stat) {
    unsigned long long st[4] = {0, 0, 0, 0};
    SOGM_HIP_CHECK(hipMemcpy(st, c->d_reset_stat, sizeof(st), hipMemcpyDeviceToHost));
    SOGM_HIP_CHECK(hipMemset(c->d_reset_stat, 0, sizeof(st)));  // (the reset's counters only: the stamp's stay)
    out[5] = (int32_t)(st[1] > 0x7FFFFFFFull ? 0x7FFFFFFFull : st[1]);
    const unsigned long long mean = st[1] ? st[0] / st[1] : 0;
    out[6] = (int32_t)(mean > 0x7FFFFFFFull ? 0x7FFFFFFFull : mean);
    const unsigned long long kib = st[1] ? st[2] / st[1] / 1024 : 0;
    out[7] = (int32_t)(kib > 0x7FFFFFFFull ? 0x7FFFFFFFull : kib);
  }
  return SOGM_OK;
}

static const char *const k_tune_names[SOGM_TUNE_N] = {
#define X(id, name, dflt, lo, hi) name,
    SOGM_TUNING_TABLE(X)
#undef X
};
static const double k_tune_range[SOGM_TUNE_N][2] = {
#define X(id, name, dflt, lo, hi) {(double)(lo), (double)(hi)},
    SOGM_TUNING_TABLE(X)
#undef X
};
static int tune_index(const char *key) {
  if (!key) return -1;
  for (int i = 0; i < SOGM_TUNE_N; ++i)
    if (std::strcmp(key, k_tune_names[i]) == 0) return i;
  return -1;
}
int sogm_set_tuning(sogm_ctx *c, const char *key, double value) {
  const int i = tune_index(key);
  if (!c) return SOGM_ERR_INVALID_ARG;
  if (i < 0) {
    char buf[160];
    std::snprintf(buf, sizeof(buf), "sogm_set_tuning: unknown key '%s'", key ? key : "(null)");
    sogm::set_error_text(buf);
    return SOGM_ERR_INVALID_ARG;
  }
  // the values end up as launch dimensions and ticket counts: NaN, infinities and anything outside the key's range
  // (SOGM_TUNING_TABLE) are refused here rather than cast to int later
  if (!(value >= k_tune_range[i][0] && value <= k_tune_range[i][1])) {
    char buf[200];
    std::snprintf(buf, sizeof(buf), "sogm_set_tuning: '%s' = %g is outside [%g, %g]", key, value, k_tune_range[i][0],
                  k_tune_range[i][1]);
    sogm::set_error_text(buf);
    return SOGM_ERR_INVALID_ARG;
  }
  c->tune[i] = value;
  return SOGM_OK;
}
int sogm_get_tuning(const sogm_ctx *c, const char *key, double *out) {
  const int i = tune_index(key);
  if (!c || !out || i < 0) return SOGM_ERR_INVALID_ARG;
  *out = c->tune[i];
  return SOGM_OK;
}
const char *sogm_tuning_key(int index) { return index >= 0 && index < SOGM_TUNE_N ? k_tune_names[index] : nullptr; }

int sogm_set_resample(sogm_ctx *c, float replan_risk_rate, int num_resample, const float *normal_table_dev, int n_table) {
  if (!c || !(replan_risk_rate >= 0.0f) || num_resample < 0 || n_table < 0) return SOGM_ERR_INVALID_ARG;
  const bool on = replan_risk_rate > 0.0f && num_resample > 0;
  if (on) {
    if (!normal_table_dev || c->n_body <= 0 || n_table < 3 * c->n_body * num_resample) {
      sogm::set_error_text("sogm_set_resample: the table must hold 3 * body particles * num_resample standard normals "
                           "(call sogm_set_body_particles first)");
      return SOGM_ERR_INVALID_ARG;
    }
    if (c->geom.half) {
      sogm::set_error_text("sogm_set_resample: fractional weights need SOGM_STORE_F32 cells");
      return SOGM_ERR_INVALID_ARG;
    }
  }
  c->geom.rs_rate = on ? replan_risk_rate : 0.0f;
  c->geom.rs_n    = on ? num_resample : 0;
  c->geom.rs_z    = on ? normal_table_dev : nullptr;
  c->geom.rs_nz   = on ? n_table : 0;
  return SOGM_OK;
}

int sogm_map_traffic(sogm_ctx *c, int64_t *out, int reset) {
  if (!c || !out) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c->d_reset_stat) SOGM_HIP_CHECK(hipMemcpy(st, c->d_reset_stat, sizeof(st), hipMemcpyDeviceToHost));
  out[0] = (int64_t)st[1];  // resets through the mark logs
  out[1] = (int64_t)st[0];  // log entries they read
  out[2] = (int64_t)st[2];  // bytes they zeroed
  out[3] = (int64_t)c->n_stamps;
  out[4] = (int64_t)st[4];  // marks (cells set to 1) the stamps wrote
  out[5] = (int64_t)st[5];  // log entries the stamps appended
  if (reset) {
    if (c->d_reset_stat) SOGM_HIP_CHECK(hipMemset(c->d_reset_stat, 0, sizeof(st)));
    c->n_stamps = 0;
  }
  return SOGM_OK;
}

// diagnostics (tools/ only): one agent's CURRENT grid ([T][V] cells, device layout) copied to a device buffer in stream
// order — no device synchronisation, no effect on the mark logs (unlike sogm_grid_ptr)
int sogm_debug_copy_grid(sogm_ctx *c, int agent, void *dst_dev, void *stream) {
  if (!c || !dst_dev || agent < 0 || agent >= c->n_agents) return SOGM_ERR_INVALID_ARG;
  const size_t bytes = (size_t)c->spec.T * (size_t)c->geom.V * c->cell_bytes();
  if (int rc = sogm::join_update(c, (hipStream_t)stream)) return rc;
  SOGM_HIP_CHECK(hipMemcpyAsync(dst_dev, (const char *)c->d_grid + (size_t)agent * bytes, bytes, hipMemcpyDeviceToDevice,
                                (hipStream_t)stream));
  return SOGM_OK;
}

}  // extern "C"
// every float whose bit pattern lies in [bits_lo, bits_hi) and its negative: GridGeom::div_res against the IEEE division
__global__ __launch_bounds__(256) void k_div_check(sogm::GridGeom g, unsigned bits_lo, unsigned bits_hi, unsigned long long *out) {
  const unsigned long long n    = (unsigned long long)(bits_hi - bits_lo);
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  unsigned                 bad = 0, first = 0xFFFFFFFFu;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const unsigned b = bits_lo + (unsigned)i;
    const float    a = __uint_as_float(b);
    const float    q = g.div_res(a), t = a / g.res, qn = g.div_res(-a), tn = (-a) / g.res;
    if (__float_as_uint(q) != __float_as_uint(t) || __float_as_uint(qn) != __float_as_uint(tn)) {
      ++bad;
      if (b < first) first = b;
    }
  }
  if (bad) {
    atomicAdd(out, (unsigned long long)bad);
    atomicMin(out + 1, (unsigned long long)first);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[2] = (unsigned long long)g.fast_div;
}
extern "C" {
// diagnostics (bench.py, tools/): how many DISTINCT 32-byte sectors the current grid's mark log names.  The log holds one entry
// per mark the wave-local lookback could not merge (sogm_map.hip, stamp_marks_trips) — marks of different waves in one sector
// are logged once each, the reset zeroes such a sector more than once and the stores merge in the L2 — so "4 B x entries +
// 32 B x entries" over-counts what HBM moves; 4 B x entries + 32 B x DISTINCT sectors is the honest denominator.  A
// test-and-set over a throw-away bitmap (one bit per sector and agent), outside any timed region: a returning atomic per
// entry at 25-30 G/s would cost the stamp more than the duplicates cost the reset.
}  // extern "C"
__global__ __launch_bounds__(256) void k_log_distinct(sogm::MarkLog lg, int n_agents, unsigned *bitmap, size_t words_per_agent,
                                                      unsigned long long *out) {
  const int agent = blockIdx.y;
  if (agent >= n_agents) return;
  const unsigned  n = lg.n[agent] > (unsigned)lg.cap ? (unsigned)lg.cap : lg.n[agent];
  const unsigned *e = lg.entries + (size_t)agent * lg.cap;
  unsigned       *bm = bitmap + (size_t)agent * words_per_agent;
  unsigned long long ent = 0, dis = 0, near8 = 0, near63 = 0, near1k = 0;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned sec = e[i];
    if (sec == 0xFFFFFFFFu || (size_t)(sec >> 5) >= words_per_agent) continue;
    ++ent;
    const unsigned bit = 1u << (sec & 31);
    if (!(atomicOr(bm + (sec >> 5), bit) & bit)) ++dis;
    // where the duplicates sit: an equal entry among the previous 8 / 63 / 1023 positions of the log (a wave appends its
    // entries as one block: "within 63" ~ what a wave-wide de-duplication could remove, "within 1023" a workgroup-wide one)
    bool d8 = false, d63 = false, d1k = false;
    for (unsigned b = 1; b <= 1023 && b <= i; ++b)
      if (e[i - b] == sec) {
        d1k = true;
        if (b <= 63) d63 = true;
        if (b <= 8) d8 = true;
        break;
      }
    near8 += d8, near63 += d63, near1k += d1k;
  }
  for (int d = 32; d >= 1; d >>= 1) {
    ent += __shfl_xor(ent, d, 64);
    dis += __shfl_xor(dis, d, 64);
    near8 += __shfl_xor(near8, d, 64);
    near63 += __shfl_xor(near63, d, 64);
    near1k += __shfl_xor(near1k, d, 64);
  }
  if ((threadIdx.x & 63) == 0 && ent) {
    atomicAdd(out, ent);
    atomicAdd(out + 1, dis);
    atomicAdd(out + 3, near8);
    atomicAdd(out + 4, near63);
    atomicAdd(out + 5, near1k);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && lg.n[agent] > (unsigned)lg.cap) atomicAdd(out + 2, 1ull);
}
extern "C" {
// host out[6] = {valid entries of the current grid's mark logs (all agents), distinct sectors among them, agents whose log
// overflowed (their reset is dense: not counted), entries with an equal entry among the previous 8 / 63 / 1023 log positions}.
// Synchronises; allocates and frees V T / 64 bytes per agent.
int sogm_debug_log_distinct(sogm_ctx *c, unsigned long long *out3_host) {
  if (!c || !out3_host) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  const int     slot = sogm::cur_slot(c);
  sogm::MarkLog lg   = c->sparse ? sogm::mark_log(c, slot) : sogm::MarkLog{nullptr, nullptr, 0, nullptr};
  for (int i = 0; i < 6; ++i) out3_host[i] = 0;
  if (!lg.entries || !c->tracked[slot]) return SOGM_ERR_STATE;
  const size_t cells_per_sector = 32 / c->cell_bytes();
  const size_t sectors = ((size_t)c->spec.T * (size_t)c->geom.V + cells_per_sector - 1) / cells_per_sector;
  const size_t words   = (sectors + 31) / 32;
  unsigned           *bm = nullptr;
  unsigned long long *d  = nullptr;
  SOGM_HIP_CHECK(hipMalloc((void **)&bm, sizeof(unsigned) * words * (size_t)c->n_agents));
  hipError_t e = hipMalloc((void **)&d, 6 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(bm, 0, sizeof(unsigned) * words * (size_t)c->n_agents);
  if (e == hipSuccess) e = hipMemset(d, 0, 6 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_log_distinct, dim3(64, c->n_agents), dim3(256), 0, nullptr, lg, c->n_agents, bm, words, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out3_host, d, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  (void)hipFree(bm);
  if (d) (void)hipFree(d);
  SOGM_HIP_CHECK(e);
  return SOGM_OK;
}

// diagnostics (tests/ only): GridGeom::div_res — the three-instruction fp32 division the stamp and the search use — against
// the IEEE division, on the device, for EVERY float in [lo, hi) and its negative.  out3_host = {mismatches, bit pattern of the
// first one (2^64 - 1 if none), whether the context uses the fast sequence at all (its resolution is 0.15f)}
int sogm_debug_div_check(sogm_ctx *c, float lo, float hi, unsigned long long *out3_host) {
  if (!c || !out3_host || !(lo > 0.0f) || !(hi > lo)) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  unsigned long long *d = nullptr;
  SOGM_HIP_CHECK(hipMalloc((void **)&d, 3 * sizeof(unsigned long long)));
  const unsigned long long init[3] = {0ull, ~0ull, 0ull};
  SOGM_HIP_CHECK(hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice));
  unsigned bl, bh;
  std::memcpy(&bl, &lo, 4);
  std::memcpy(&bh, &hi, 4);
  hipLaunchKernelGGL(k_div_check, dim3(4096), dim3(256), 0, nullptr, c->geom, bl, bh, d);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out3_host, d, sizeof(init), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  SOGM_HIP_CHECK(e);
  return SOGM_OK;
}

enum { UF_ERR = 1024, UF_STAGE = 2048, UF_STAGE_STRIDE = 32 };  // the update flow's control words (k_update_flow)
// diagnostics (tools/ only): the update flow's control words after a device synchronisation —
// out = {epoch, pending, ticket, error, ...ctl[2..7], stage[A], map_ready[A]}
int sogm_debug_update_flow(sogm_ctx *c, int32_t *out, int cap) {
  if (!c || !out || cap < 10 + 2 * c->n_agents) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  (void)hipDeviceSynchronize();
  const int A = c->n_agents;
  out[0]      = c->map_epoch;
  out[1]      = c->update_pending;
  if (!c->d_update_ctl) return SOGM_ERR_STATE;
  {
    std::vector<int> w((size_t)(UF_STAGE + UF_STAGE_STRIDE * A));
    SOGM_HIP_CHECK(hipMemcpy(w.data(), c->d_update_ctl, sizeof(int) * w.size(), hipMemcpyDeviceToHost));
    out[2] = w[0];
    out[3] = w[UF_ERR];
    for (int a = 0; a < A; ++a) out[10 + a] = w[(size_t)(UF_STAGE + UF_STAGE_STRIDE * a)];
  }
  SOGM_HIP_CHECK(hipMemcpy(out + 10 + A, c->d_map_ready, sizeof(int) * A, hipMemcpyDeviceToHost));
  return SOGM_OK;
}
// ... and its per-agent stamps [A][4] (100 MHz wall clock) + the order it took the agents in [A]
int sogm_debug_update_flow_times(sogm_ctx *c, int64_t *out_ts, int32_t *out_order) {
  if (!c || !out_ts || !out_order || !c->d_update_ts) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  (void)hipDeviceSynchronize();
  SOGM_HIP_CHECK(hipMemcpy(out_ts, c->d_update_ts, sizeof(long long) * 4 * c->n_agents, hipMemcpyDeviceToHost));
  SOGM_HIP_CHECK(hipMemcpy(out_order, c->d_update_order, sizeof(int) * c->n_agents, hipMemcpyDeviceToHost));
  return SOGM_OK;
}
int sogm_grid_history(sogm_ctx *c, int32_t *out) {
  if (!c || !out) return SOGM_ERR_INVALID_ARG;
  const int slot = sogm::cur_slot(c);
  out[0] = slot;
  out[1] = c->hist_sparse[slot];
  out[2] = c->hist_dense[slot];
  out[3] = c->cur_prestamped;
  return SOGM_OK;
}

int sogm_set_overlap_clear(sogm_ctx *c, int mode) {
  if (!c || mode < 0 || mode > 3) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  if (c->update_pending) {
    (void)hipDeviceSynchronize();
    c->update_pending = 0;
  }
  if (c->precleared || c->n_ready || c->n_dirty) {
    // pre-clears may be in flight: let them finish and forget them (the next update clears its grid itself)
    (void)hipDeviceSynchronize();
    c->precleared = 0;
  }
  c->prestamp_slot = -1;  // (a pre-stamped spare is dirty like the others: reset below)
  const int want = mode >= 2 ? mode : 1;  // grids in the pool
  // the current grid stays where it is (slot cur_idx); spares are added / released around it
  if (c->n_pool == 0) {
    c->pool[0] = c->d_grid;
    c->n_pool  = 1;
    c->cur_idx = 0;
  }
  if (c->cur_idx != 0) {  // keep the current grid in slot 0 so that spares are slots 1..n_pool-1
    float *t           = c->pool[0];
    c->pool[0]         = c->pool[c->cur_idx];
    c->pool[c->cur_idx] = t;
    std::swap(c->d_log[0], c->d_log[c->cur_idx]);  // the mark logs follow their grids
    std::swap(c->d_log_n[0], c->d_log_n[c->cur_idx]);
    std::swap(c->tracked[0], c->tracked[c->cur_idx]);
    std::swap(c->hist_sparse[0], c->hist_sparse[c->cur_idx]);
    std::swap(c->hist_dense[0], c->hist_dense[c->cur_idx]);
    c->cur_idx         = 0;
  }
  while (c->n_pool > want) {
    (void)hipFree(c->pool[--c->n_pool]);
    c->pool[c->n_pool] = nullptr;
  }
  for (int i = 1; i < 3; ++i) c->tracked[i] = 0;  // spares hold garbage: their first reset is the dense clear
  for (int i = 1; i < 3; ++i) c->hist_sparse[i] = c->hist_dense[i] = 0;
  const size_t bytes = ((size_t)sogm_grid_bytes(c) + 15) & ~(size_t)15;
  const int    had   = c->n_pool;
  while (c->n_pool < want) {
    float *g = nullptr;
    if (hipMalloc(&g, bytes) != hipSuccess) {
      (void)hipGetLastError();
      while (c->n_pool > had) {  // all or nothing: the mode is unchanged
        (void)hipFree(c->pool[--c->n_pool]);
        c->pool[c->n_pool] = nullptr;
      }
      sogm::set_error("sogm_set_overlap_clear: no room for the spare grid(s)", hipErrorOutOfMemory);
      c->n_ready = 0;
      c->n_dirty = 0;
      for (int i = 1; i < c->n_pool; ++i) c->dirty[c->n_dirty++] = i;
      return SOGM_ERR_CAPACITY;
    }
    if (!c->pool_ev[c->n_pool] &&
        hipEventCreateWithFlags(&c->pool_ev[c->n_pool], hipEventDisableTiming) != hipSuccess) {
      // same roll-back as a failed allocation: the grids added by this call go, the lists describe the pool again
      (void)hipFree(g);
      while (c->n_pool > had) {
        (void)hipFree(c->pool[--c->n_pool]);
        c->pool[c->n_pool] = nullptr;
      }
      c->n_ready = 0;
      c->n_dirty = 0;
      for (int i = 1; i < c->n_pool; ++i) c->dirty[c->n_dirty++] = i;
      sogm::set_error("sogm_set_overlap_clear: event", hipErrorUnknown);
      return SOGM_ERR_HIP;
    }
    c->pool[c->n_pool++] = g;
  }
  for (int i = 0; i < c->n_pool; ++i)  // every slot takes the spare role in turn
    if (!c->pool_ev[i]) SOGM_HIP_CHECK(hipEventCreateWithFlags(&c->pool_ev[i], hipEventDisableTiming));
  c->n_ready = 0;
  c->n_dirty = 0;
  for (int i = 1; i < c->n_pool; ++i) c->dirty[c->n_dirty++] = i;  // spares hold garbage until a replan clears them
  c->overlap = mode;
  return SOGM_OK;
}

int sogm_set_profiling(sogm_ctx *c, int enable) {
  return sogm_set_profiling_slots(c, enable ? (1 << SOGM_PROF_N) - 1 : 0);
}

int sogm_set_profiling_slots(sogm_ctx *c, int slot_mask) {
  if (!c || slot_mask < 0 || slot_mask >= (1 << SOGM_PROF_N)) return SOGM_ERR_INVALID_ARG;
  c->profiling = slot_mask;
  for (int k = 0; k < SOGM_PROF_N; ++k) c->ring_n[k] = 0;
  return SOGM_OK;
}

int sogm_profile_read(sogm_ctx *c, double *out_ms) {
  if (!c || !out_ms) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  for (int k = 0; k < SOGM_PROF_N; ++k) {
    out_ms[k] = -1.0;
    if (c->ring_n[k] <= 0 || !c->ring[k]) continue;
    hipEvent_t *p  = c->ring[k] + 2 * ((c->ring_n[k] - 1) % SOGM_PROF_RING);
    float       ms = 0.f;
    if (hipEventElapsedTime(&ms, p[0], p[1]) == hipSuccess) out_ms[k] = (double)ms;
  }
  return SOGM_OK;
}

int sogm_profile_read_all(sogm_ctx *c, int slot, double *out_ms, int cap, int *out_n) {
  if (!c || !out_ms || !out_n || slot < 0 || slot >= SOGM_PROF_N || cap < 0) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  long long n = c->ring_n[slot];
  if (n > SOGM_PROF_RING) n = SOGM_PROF_RING;
  if (n > cap) n = cap;
  *out_n = 0;
  if (!c->ring[slot]) return SOGM_OK;
  for (long long i = c->ring_n[slot] - n; i < c->ring_n[slot]; ++i) {  // oldest kept launch first
    hipEvent_t *p  = c->ring[slot] + 2 * (i % SOGM_PROF_RING);
    float       ms = 0.f;
    if (hipEventElapsedTime(&ms, p[0], p[1]) != hipSuccess) ms = -1.f;
    out_ms[(*out_n)++] = (double)ms;
  }
  return SOGM_OK;
}

int sogm_set_body_particles(sogm_ctx *c, const double *xyz, int n) {
  if (!c || !xyz || n <= 0) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  if (c->d_body) (void)hipFree(c->d_body);
  c->d_body = nullptr;
  SOGM_HIP_CHECK(hipMalloc(&c->d_body, sizeof(double) * 3 * n));
  SOGM_HIP_CHECK(hipMemcpy(c->d_body, xyz, sizeof(double) * 3 * n, hipMemcpyHostToDevice));
  c->n_body = n;
  for (int k = 0; k < 3; ++k) {
    double m = 0.0;
    for (int e = 0; e < n; ++e) m = std::fabs(xyz[e * 3 + k]) > m ? std::fabs(xyz[e * 3 + k]) : m;
    c->geom.body_ext[k] = (float)m;
  }
  return SOGM_OK;
}

static int clear_grid(sogm_ctx *c, hipStream_t st) { return sogm::reset_slot(c, st, sogm::cur_slot(c), c->d_grid, false); }

// ---- update flow (tuning key update_flow): the maps of a lock-step tick agent by agent ----
// One persistent launch of one-wave workgroups over tickets; an agent's tickets are consecutive (bits, marks, overlay) and
// agents are taken in `order` (the previous tick's longest chains first), so the first agents' maps are complete after the
// latency of ONE ticket chain instead of after four kernels over the whole swarm; the agent's last ticket stores the
// epoch into map_ready[agent] (release), which sogm_replan's search workgroups wait for.  A ticket only ever waits for
// tickets of its own agent, all of which are held by resident or earlier waves: no residency assumption beyond one agent's
// worth of waves.  Same cells, same log as the four kernels (tests/test_update_flow_gpu.py).
struct UpdateFlowDev {
  void                 *grid;
  unsigned             *bits;
  int                   words;
  const float          *cloud;
  CloudBlocks           cb;
  const SogmCylinder   *cyl;
  int                   n_cyl;
  const void           *cand;
  const int            *n_cand;
  MarkLog               lg;
  const float          *poses;   // the context's copies (filed by k_cull_cylinders, the launch before)
  const double         *stamps;
  const SogmTrajRecord *rec;
  int                   n_rec;
  const int32_t        *ego_ids;
  const double         *body;
  int                   n_body;
  int                   n_agents, n_bits, n_marks, n_splat;
  int                  *ctl;      // ticket, error and per-agent progress words (UF_* below)
  int                   chunk;    // consecutive tickets per claim
  long long            *ts;       // [A][4] wall clock: first ticket claimed, bits complete, marks complete, map ready (diagnostics)
  int                  *map_ready;
  int                   epoch;
  const int            *order;
};
// (no LDS: the replan's corridor workgroups, resident beside this kernel and waiting for routes, hold every CU's LDS)
// Control words, each on a line of its own (thousands of waves claim tickets, bump and poll them — in one 128-byte line that
// line's memory channel is the bottleneck, as the flight's header showed): ticket at ctl[0], error at ctl[UF_ERR], the
// per-agent progress counters at ctl[UF_STAGE + 32 agent].
extern "C++" {
template <bool CACHED>
__global__ __launch_bounds__(64) void k_update_flow(GridGeom g, UpdateFlowDev u) {
  const int lane  = threadIdx.x;
  const int per   = u.n_bits + u.n_marks + u.n_splat;
  const int total = u.n_agents * per;
  int      *err   = u.ctl + UF_ERR;
  for (;;) {
    int t0 = 0;
    if (lane == 0) t0 = atomicAdd(&u.ctl[0], u.chunk);  // a chunk of consecutive tickets per claim
    t0 = __builtin_amdgcn_readfirstlane(t0);
    if (t0 >= total) break;
    for (int t = t0; t < t0 + u.chunk && t < total; ++t) {
      const int agent = u.order[t / per];
      const int s     = t % per;
      int      *stage = u.ctl + UF_STAGE + UF_STAGE_STRIDE * agent;
      bool      last  = false;
      if (s == 0 && lane == 0) u.ts[agent * 4] = wall_clock64();
      if (s < u.n_bits) {
        stamp_bits_blocks(g, u.cloud, u.cb, agent, s, u.n_bits, u.poses[agent * 3], u.poses[agent * 3 + 1],
                          u.poses[agent * 3 + 2], u.bits + (size_t)agent * u.words, lane);
        __threadfence();
        if (lane == 0 && atomicAdd(stage, 1) + 1 == u.n_bits) u.ts[agent * 4 + 1] = wall_clock64();
      } else if (s < u.n_bits + u.n_marks) {
        if (flow_wait_count(stage, u.n_bits, err)) return;
        stamp_marks_trips<CACHED>(g, u.grid, u.bits, u.words, u.cyl, u.n_cyl, u.poses, (const CylCand *)u.cand, u.n_cand,
                                  agent, u.lg, (s - u.n_bits) * 256, u.n_marks * 256);
        __threadfence();
        if (lane == 0) {
          const int n = atomicAdd(stage, 1) + 1;
          last        = n == per;
          if (n == u.n_bits + u.n_marks) u.ts[agent * 4 + 2] = wall_clock64();
        }
      } else {
        if (flow_wait_count(stage, u.n_bits + u.n_marks, err)) return;  // stores of 1.0 first, the additions after them
        const int n = u.n_rec * g.T;
        for (int i = (s - u.n_bits - u.n_marks) * 64 + lane; i < n; i += u.n_splat * 64)
          splat_item(g, u.grid, u.rec[i / g.T], agent, i % g.T, u.ego_ids, u.poses, u.stamps, u.body, u.n_body, u.lg);
        __threadfence();
        if (lane == 0) last = atomicAdd(stage, 1) + 1 == per;
      }
      if (last) {
        u.ts[agent * 4 + 3] = wall_clock64();
        __hip_atomic_store(&u.map_ready[agent], u.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
}  // extern "C++"
__global__ void k_iota(int *p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}
namespace sogm {
// stream, events and words of the update flow (first use)
static int update_flow_setup(sogm_ctx *c) {
  if (c->ustream) return SOGM_OK;
  const int A = c->n_agents;
  SOGM_HIP_CHECK(sogm::create_stream_partitioned(&c->ustream, 0));
  SOGM_HIP_CHECK(hipEventCreateWithFlags(&c->ev_uin, hipEventDisableTiming));
  SOGM_HIP_CHECK(hipEventCreateWithFlags(&c->ev_udone, hipEventDisableTiming));
  SOGM_HIP_CHECK(hipMalloc((void **)&c->d_map_ready, sizeof(int) * A));
  SOGM_HIP_CHECK(hipMalloc((void **)&c->d_update_ctl, sizeof(int) * (size_t)(UF_STAGE + UF_STAGE_STRIDE * A)));
  SOGM_HIP_CHECK(hipMalloc((void **)&c->d_update_order, sizeof(int) * A));
  SOGM_HIP_CHECK(hipMalloc((void **)&c->d_update_ts, sizeof(long long) * 4 * A));
  SOGM_HIP_CHECK(hipMemset(c->d_update_ts, 0, sizeof(long long) * 4 * A));
  SOGM_HIP_CHECK(hipMemset(c->d_map_ready, 0, sizeof(int) * A));
  hipLaunchKernelGGL(k_iota, dim3((A + 255) / 256), dim3(256), 0, nullptr, c->d_update_order, A);
  SOGM_HIP_CHECK(hipGetLastError());
  SOGM_HIP_CHECK(hipStreamSynchronize(nullptr));  // (null-stream work is not ordered with the non-blocking streams)
  return SOGM_OK;
}
}  // namespace sogm

// updateMap for every agent; with `records` (sogm_update_gt_swarm) the neighbour overlay follows in the same call
static int update_gt_impl(sogm_ctx *c, const float *cloud_xyz, const int32_t *cloud_range,
                          const SogmCylinder *cylinders, int n_cyl, const float *poses, const double *stamps,
                          const SogmTrajRecord *records, int n_records, const int32_t *ego_ids, bool fused,
                          hipStream_t st, const SogmWorld *world = nullptr) {
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  sogm::CloudBlocks cb{};
  if (world)
    if (int rc = sogm::world_blocks(c, world, &cb)) return rc;
  const int A = c->n_agents;
  if (int rc = sogm::join_update(c, st)) return rc;    // an update flow of the previous tick nobody joined
  if (int rc = sogm::join_prestamp(c, st)) return rc;  // a pre-stamp of the last replan may still be running
  if (fused)
    if (int rc = sogm::join_exchange(c, st)) return rc;  // records may come from an all-gather in flight
  // (poses / stamps are filed into the context by k_cull_cylinders below)
  // A grid the previous replan pre-stamped is not what this call builds (other inputs): it is the front of the ready
  // queue, i.e. the grid adopted below — its marks are in its log, so it is reset again before the stamp.
  const int stale = c->prestamp_slot;
  c->prestamp_slot = -1;
  c->cur_prestamped = 0;
  c->records_final_valid = 0;
  if (c->precleared) {
    // the grid was already cleared on the side stream during the previous tick
    int rc = sogm::adopt_preclear(c, st);
    if (rc) return rc;
    if (stale >= 0 && sogm::cur_slot(c) == stale) {
      rc = sogm::reset_slot(c, st, stale, c->d_grid, false);
      if (rc) return rc;
    }
    if (stale >= 0 && c->d_stamp_bits) {
      // a pre-stamp that was cut short (a failed tick) may have left occupancy bits behind — only consumed words are
      // zeroed — and they are relative to ITS map centres: start this stamp from a clean mask
      const int words = (((c->geom.V + 31) / 32) + 255) & ~255;
      SOGM_HIP_CHECK(hipMemsetAsync(c->d_stamp_bits, 0, sizeof(unsigned) * (size_t)words * A, st));
    }
  } else {
    int rc = clear_grid(c, st);
    if (rc) return rc;
    if (c->overlap >= 2 && c->n_dirty > 0 && c->clear_gate && c->tune_i(SOGM_TUNE_CLEAR_EARLY)) {
      // tuning-aid mode: the pool's spare grids (dirty at the start, no readers) are cleared from here on as well,
      // so that a run of updates alone exercises the pooled clear (tools/diag_clear_pmc.py)
      c->clear_epoch_ahead = 1;
      SOGM_HIP_CHECK(hipEventRecord(c->ev_grid_free, st));
      rc                   = sogm::queue_spare_clears(c, c->ev_grid_free);
      c->clear_epoch_ahead = 0;
      if (rc) return rc;
    }
  }
  // candidate cylinders per agent, then one-wave workgroups stride over each agent's cloud range
  int words = 0;
  if (int rc = sogm::stamp_scratch(c, st, &words)) return rc;
  const int stamp_wgs = c->tune_i(SOGM_TUNE_STAMP_WGS) > 0 ? c->tune_i(SOGM_TUNE_STAMP_WGS) : 256;  // one-wave workgroups per agent
  c->n_stamps++;
  prof_begin(c, SOGM_PROF_STAMP, st);
  hipLaunchKernelGGL(k_cull_cylinders, dim3(A), dim3(64), 0, st, c->geom, cylinders, n_cyl, poses,
                     (CylCand *)c->d_cand, c->d_ncand, stamps, c->d_poses, c->d_stamps, cb, c->h_tick_clock);
  if (world && c->tune_i(SOGM_TUNE_UPDATE_FLOW) != 0) {
    // the maps agent by agent on the flow's stream; the caller's stream goes on (sogm_replan's searches wait per agent,
    // everything else joins the flow's end: sogm::join_update)
    if (int rc = sogm::update_flow_setup(c)) return rc;
    UpdateFlowDev u{};
    u.grid    = (void *)c->d_grid;
    u.bits    = c->d_stamp_bits;
    u.words   = words;
    u.cloud   = cloud_xyz;
    u.cb      = cb;
    u.cyl     = cylinders;
    u.n_cyl   = n_cyl;
    u.cand    = c->d_cand;
    u.n_cand  = c->d_ncand;
    u.lg      = sogm::mark_log(c, sogm::cur_slot(c));
    u.poses   = c->d_poses;
    u.stamps  = c->d_stamps;
    u.rec     = records;
    u.n_rec   = fused ? n_records : 0;
    u.ego_ids = ego_ids;
    u.body    = c->d_body;
    u.n_body  = c->n_body;
    u.n_agents = A;
    u.n_bits   = c->tune_i(SOGM_TUNE_UPDATE_BITS);
    u.n_marks  = c->tune_i(SOGM_TUNE_UPDATE_MARKS);
    u.n_splat  = u.n_rec > 0 ? c->tune_i(SOGM_TUNE_UPDATE_SPLAT) : 0;
    u.ctl      = c->d_update_ctl;
    u.ts       = c->d_update_ts;
    u.chunk    = c->tune_i(SOGM_TUNE_UPDATE_CHUNK) > 0 ? c->tune_i(SOGM_TUNE_UPDATE_CHUNK) : 1;
    u.map_ready = c->d_map_ready;
    u.epoch     = ++c->map_epoch;
    u.order     = c->d_update_order;
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
    int wgs = c->tune_i(SOGM_TUNE_UPDATE_WGS) > 0 ? c->tune_i(SOGM_TUNE_UPDATE_WGS) : 16 * n_cu;
    const int total = A * (u.n_bits + u.n_marks + u.n_splat);
    if (wgs > total) wgs = total;
    SOGM_HIP_CHECK(hipEventRecord(c->ev_uin, st));
    SOGM_HIP_CHECK(hipStreamWaitEvent(c->ustream, c->ev_uin, 0));
    SOGM_HIP_CHECK(hipMemsetAsync(c->d_update_ctl, 0, sizeof(int) * (size_t)(UF_STAGE + UF_STAGE_STRIDE * A), c->ustream));
    if (c->tune_i(SOGM_TUNE_UPDATE_CACHED))
      hipLaunchKernelGGL(k_update_flow<true>, dim3(wgs), dim3(64), 0, c->ustream, c->geom, u);
    else
      hipLaunchKernelGGL(k_update_flow<false>, dim3(wgs), dim3(64), 0, c->ustream, c->geom, u);
    SOGM_HIP_CHECK(hipGetLastError());
    prof_end(c, SOGM_PROF_STAMP, c->ustream);
    SOGM_HIP_CHECK(hipEventRecord(c->ev_udone, c->ustream));
    c->update_pending = 1;
    c->updated        = 1;
    return SOGM_OK;
  }
  const int bits_wgs = c->tune_i(SOGM_TUNE_STAMP_BITS_WGS) > 0 ? c->tune_i(SOGM_TUNE_STAMP_BITS_WGS) : stamp_wgs;
  if (world)
    hipLaunchKernelGGL(k_stamp_bits_blocks, dim3(bits_wgs, A), dim3(64), 0, st, c->geom, cloud_xyz, cb, c->d_poses,
                       c->d_stamp_bits, words);
  else
    hipLaunchKernelGGL(k_stamp_bits, dim3(bits_wgs, A), dim3(64), 0, st, c->geom, cloud_xyz, cloud_range, c->d_poses,
                       c->d_stamp_bits, words, 0);
  const sogm::MarkLog lg = sogm::mark_log(c, sogm::cur_slot(c));
  // (dynamic LDS the kernel does not use bounds its waves per CU: the marks' scattered stores merge worse in L2 the more
  //  waves interleave theirs — tuning key stamp_lds_kb, 160 / kb workgroups per CU)
  const size_t marks_lds = (size_t)c->tune_i(SOGM_TUNE_STAMP_LDS_KB) * 1024;
  if (c->tune_i(SOGM_TUNE_STAMP_CACHED))
    hipLaunchKernelGGL(k_stamp_marks_cached, dim3(stamp_wgs, A), dim3(64), marks_lds, st, c->geom, (void *)c->d_grid,
                       c->d_stamp_bits, words, cylinders, n_cyl, c->d_poses, (const CylCand *)c->d_cand,
                       (const int *)c->d_ncand, 0, lg);
  else
  {
    // the log pass's sector cache: T x 64 words of LDS per (one-wave) workgroup, in front of the tuning aid's padding
    const int    lds_log = lg.entries && c->spec.T <= 64 && c->tune_i(SOGM_TUNE_STAMP_LDS_LOG) ? 1 : 0;
    const size_t lds     = marks_lds + (lds_log ? sizeof(unsigned) * 64 * (size_t)c->spec.T : 0);
    hipLaunchKernelGGL(k_stamp_marks, dim3(stamp_wgs, A), dim3(64), lds, st, c->geom, (void *)c->d_grid, c->d_stamp_bits,
                       words, cylinders, n_cyl, c->d_poses, (const CylCand *)c->d_cand, (const int *)c->d_ncand, 0, lg, lds_log);
  }
  prof_end(c, SOGM_PROF_STAMP, st);
  SOGM_HIP_CHECK(hipGetLastError());
  if (fused && n_records > 0) {
    const long long total = (long long)A * n_records * c->spec.T;
    prof_begin(c, SOGM_PROF_SPLAT, st);
    hipLaunchKernelGGL(k_splat_neighbours, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, c->geom,
                       (void *)c->d_grid, records, n_records, ego_ids, c->d_poses, c->d_stamps, c->d_body, c->n_body,
                       A, 0, lg, nullptr, nullptr);
    prof_end(c, SOGM_PROF_SPLAT, st);
    SOGM_HIP_CHECK(hipGetLastError());
  }
  c->updated = 1;
  return SOGM_OK;
}

int sogm_update_gt(sogm_ctx *c, const float *cloud_xyz, const int32_t *cloud_range,
                   const SogmCylinder *cylinders, int n_cyl, const float *poses,
                   const double *stamps, void *stream) {
  if (!c || !cloud_xyz || !cloud_range || !poses || !stamps || n_cyl < 0 ||
      (n_cyl > 0 && !cylinders))
    return SOGM_ERR_INVALID_ARG;
  return update_gt_impl(c, cloud_xyz, cloud_range, cylinders, n_cyl, poses, stamps, nullptr, 0, nullptr, false,
                        (hipStream_t)stream);
}

int sogm_update_gt_swarm(sogm_ctx *c, const float *cloud_xyz, const int32_t *cloud_range,
                         const SogmCylinder *cylinders, int n_cyl, const float *poses, const double *stamps,
                         const SogmTrajRecord *records, int n_records, const int32_t *ego_ids, void *stream) {
  if (!c || !cloud_xyz || !cloud_range || !poses || !stamps || n_cyl < 0 || (n_cyl > 0 && !cylinders) ||
      n_records < 0 || (n_records > 0 && !records) || !ego_ids)
    return SOGM_ERR_INVALID_ARG;
  if (!c->d_body || c->n_body <= 0) return SOGM_ERR_STATE;
  return update_gt_impl(c, cloud_xyz, cloud_range, cylinders, n_cyl, poses, stamps, records, n_records, ego_ids,
                        true, (hipStream_t)stream);
}

__global__ void k_device_clock(long long *out) { *out = wall_clock64(); }
int sogm_device_clock(sogm_ctx *c, int64_t *out_ticks, void *stream) {
  if (!c || !out_ticks) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  if (!c->h_tick_clock) {
    SOGM_HIP_CHECK(hipHostMalloc((void **)&c->h_tick_clock, sizeof(long long) * 4, hipHostMallocMapped));
    for (int i = 0; i < 4; ++i) c->h_tick_clock[i] = 0;
  }
  hipLaunchKernelGGL(k_device_clock, dim3(1), dim3(1), 0, (hipStream_t)stream, c->h_tick_clock + 2);
  SOGM_HIP_CHECK(hipGetLastError());
  SOGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  *out_ticks = (int64_t) * (volatile long long *)(c->h_tick_clock + 2);
  return SOGM_OK;
}
int sogm_tick_clock(sogm_ctx *c, int64_t *out2) {
  if (!c || !out2) return SOGM_ERR_INVALID_ARG;
  volatile long long *h = c->h_tick_clock;
  out2[0] = h ? (int64_t)h[0] : 0;
  out2[1] = h ? (int64_t)h[1] : 0;
  return SOGM_OK;
}

int sogm_cloud_block_bounds(const float *cloud_xyz, int n_points, int block_points, float *out_bounds, void *stream) {
  if (!cloud_xyz || !out_bounds || n_points < 0 || block_points < 64 || block_points > 4096) return SOGM_ERR_INVALID_ARG;
  const int nb = (n_points + block_points - 1) / block_points;
  if (nb == 0) return SOGM_OK;
  hipLaunchKernelGGL(k_block_bounds, dim3(nb), dim3(64), 0, (hipStream_t)stream, cloud_xyz, n_points, block_points, out_bounds);
  SOGM_HIP_CHECK(hipGetLastError());
  return SOGM_OK;
}

int sogm_update_world(sogm_ctx *c, const SogmWorld *w, const float *poses, const double *stamps,
                      const SogmTrajRecord *records, int n_records, const int32_t *ego_ids, void *stream) {
  if (!c || !w || !poses || !stamps || !w->cloud_xyz || !w->block_bounds || w->n_points < 0 || w->block_points < 64 ||
      w->block_points > 4096 || w->n_blocks != (w->n_points + w->block_points - 1) / w->block_points || w->n_cyl < 0 ||
      (w->n_cyl > 0 && !w->cylinders) || n_records < 0 || (n_records > 0 && (!records || !ego_ids)))
    return SOGM_ERR_INVALID_ARG;
  if (n_records > 0 && (!c->d_body || c->n_body <= 0)) return SOGM_ERR_STATE;
  return update_gt_impl(c, w->cloud_xyz, nullptr, w->cylinders, w->n_cyl, poses, stamps, records, n_records, ego_ids,
                        n_records > 0, (hipStream_t)stream, w);
}

int sogm_update_prestamped(sogm_ctx *c, const SogmTrajRecord *records, int n_records, const int32_t *ego_ids,
                           void *stream) {
  if (!c || n_records < 0 || (n_records > 0 && (!records || !ego_ids))) return SOGM_ERR_INVALID_ARG;
  if (c->prestamp_slot < 0 || c->n_ready <= 0 || c->ready[0] != c->prestamp_slot || !c->precleared) {
    sogm::set_error_text("sogm_update_prestamped: the previous sogm_replan did not pre-stamp the next grid");
    return SOGM_ERR_STATE;
  }
  if (n_records > 0 && (!c->d_body || c->n_body <= 0)) return SOGM_ERR_STATE;
  if (c->ps_fail_host && c->ps_fail_host[1] != c->ps_fail_seen) {
    sogm::set_error_text("sogm_update_prestamped: the replan that pre-stamped this grid failed on the device "
                         "(sogm_planner_flow_failures); build the map with sogm_update_gt[_swarm] instead");
    return SOGM_ERR_STATE;
  }
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  if (int rc = sogm::join_update(c, st)) return rc;
  if (n_records > 0)
    if (int rc = sogm::join_exchange(c, st)) return rc;  // records may come from an all-gather in flight
  if (int rc = sogm::adopt_preclear(c, st, false)) return rc;  // the pre-stamped grid becomes the current one
  c->records_final_valid = 0;
  std::swap(c->d_poses, c->d_poses_next);                // its map centres and stamps with it
  std::swap(c->d_stamps, c->d_stamps_next);
  c->prestamp_slot = -1;
  c->cur_prestamped = 1;
  if (n_records > 0) {
    // The replan that pre-stamped this grid left the caller's stream behind its fan-in, not behind the pre-stamp's
    // end: the overlay is launched now, narrow, and waits per agent for the stamp's completion word — it runs under the
    // pre-stamp's tail (the last agents' stamps, the report) instead of after it.
    // (a pre-stamp that has ended already — a host that synchronises every tick — needs no waiting: full width)
    if (c->pdone_pending && hipEventQuery(c->ev_pdone) == hipSuccess)
      if (int rc = sogm::join_prestamp(c, st)) return rc;
    (void)hipGetLastError();  // (hipErrorNotReady is not an error here)
    const int      *stage = c->pdone_pending ? c->ps_stage : nullptr;
    const long long total = (long long)c->n_agents * n_records * c->spec.T;
    long long       nblk  = (total + 255) / 256;
    // workgroups of the waiting launch (128 / 256 / 512 / 2048: 11.22 / 11.25 / 11.31 / 11.26 ms per tick)
    const int cap = c->tune_i(SOGM_TUNE_SPLAT_WGS) > 0 ? c->tune_i(SOGM_TUNE_SPLAT_WGS) : 256;
    if (stage && nblk > cap) nblk = cap;
    prof_begin(c, SOGM_PROF_SPLAT, st);
    hipLaunchKernelGGL(k_splat_neighbours, dim3((unsigned)nblk), dim3(256), 0, st, c->geom, (void *)c->d_grid, records,
                       n_records, ego_ids, c->d_poses, c->d_stamps, c->d_body, c->n_body, c->n_agents, 0,
                       sogm::mark_log(c, sogm::cur_slot(c)), stage, c->ps_err);
    prof_end(c, SOGM_PROF_SPLAT, st);
    SOGM_HIP_CHECK(hipGetLastError());
  }
  if (int rc = sogm::join_prestamp(c, st)) return rc;  // whatever follows is behind the pre-stamp's end
  if (int rc = sogm::retire_wide_clear(c, st)) return rc;
  c->updated = 1;
  return SOGM_OK;
}

int sogm_prestamp_pending(const sogm_ctx *c) { return c && c->prestamp_slot >= 0 ? 1 : 0; }
int sogm_prestamp_join(sogm_ctx *c, void *stream) {
  if (!c) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  return sogm::join_prestamp(c, (hipStream_t)stream);
}

int sogm_project_neighbours(sogm_ctx *c, const SogmTrajRecord *records, int n_records,
                            const int32_t *ego_ids, void *stream) {
  if (!c || n_records < 0 || (n_records > 0 && !records) || !ego_ids) return SOGM_ERR_INVALID_ARG;
  if (!c->updated) return SOGM_ERR_STATE;
  if (!c->d_body || c->n_body <= 0) return SOGM_ERR_STATE;
  if (n_records == 0) return SOGM_OK;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  if (int rc = sogm::join_update(c, (hipStream_t)stream)) return rc;
  if (int rc = sogm::join_exchange(c, (hipStream_t)stream)) return rc;  // records may come from an all-gather in flight
  const long long total = (long long)c->n_agents * n_records * c->spec.T;
  const int       nblk  = (int)((total + 255) / 256);
  prof_begin(c, SOGM_PROF_SPLAT, (hipStream_t)stream);
  hipLaunchKernelGGL(k_splat_neighbours, dim3(nblk), dim3(256), 0, (hipStream_t)stream, c->geom,
                     (void *)c->d_grid, records, n_records, ego_ids, c->d_poses, c->d_stamps, c->d_body,
                     c->n_body, c->n_agents, 0, sogm::mark_log(c, sogm::cur_slot(c)), nullptr, nullptr);
  prof_end(c, SOGM_PROF_SPLAT, (hipStream_t)stream);
  SOGM_HIP_CHECK(hipGetLastError());
  return SOGM_OK;
}

int sogm_set_future_risk(sogm_ctx *c, const float *grid_vt, const float *poses,
                         const double *stamps, void *stream) {
  if (!c || !grid_vt || !poses || !stamps) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  if (int rc = sogm::join_update(c, st)) return rc;
  SOGM_HIP_CHECK(hipMemcpyAsync(c->d_poses, poses, sizeof(float) * 3 * c->n_agents,
                                hipMemcpyDeviceToDevice, st));
  SOGM_HIP_CHECK(hipMemcpyAsync(c->d_stamps, stamps, sizeof(double) * c->n_agents,
                                hipMemcpyDeviceToDevice, st));
  if (c->precleared) {
    int rc = sogm::adopt_preclear(c, st);
    if (rc) return rc;
  }
  const int    V = c->geom.V, T = c->spec.T;
  const size_t per = (size_t)V * T;
  c->tracked[sogm::cur_slot(c)] = 0;  // every cell is written: the next reset of this grid is the dense clear
  c->cur_prestamped = 0;
  for (int a = 0; a < c->n_agents; ++a) {
    hipLaunchKernelGGL(k_vt_to_slabs, dim3((V + 255) / 256), dim3(256), 0, st, grid_vt + a * per, c->geom,
                       (void *)((char *)c->d_grid + a * per * c->cell_bytes()));
  }
  SOGM_HIP_CHECK(hipGetLastError());
  c->updated = 1;
  return SOGM_OK;
}

int sogm_download_reference_layout(sogm_ctx *c, int agent, float *out) {
  if (!c || !out || agent < 0 || agent >= c->n_agents) return SOGM_ERR_INVALID_ARG;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  const int    V = c->geom.V, T = c->spec.T;
  const size_t per = (size_t)V * T;
  SOGM_HIP_CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_slabs_to_vt, dim3((V + 255) / 256), dim3(256), 0, 0,
                     (const void *)((const char *)c->d_grid + (size_t)agent * per * c->cell_bytes()), c->geom,
                     c->d_scratch_vt);
  SOGM_HIP_CHECK(hipGetLastError());
  SOGM_HIP_CHECK(hipMemcpy(out, c->d_scratch_vt, per * sizeof(float), hipMemcpyDeviceToHost));
  return SOGM_OK;
}

int sogm_map_state(sogm_ctx *c, int agent, double *out_time, float *out_center, void *stream) {
  if (!c || agent < 0 || agent >= c->n_agents || (!out_time && !out_center)) return SOGM_ERR_INVALID_ARG;
  if (!c->updated) return SOGM_ERR_STATE;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  if (out_time)
    SOGM_HIP_CHECK(hipMemcpyAsync(out_time, c->d_stamps + agent, sizeof(double), hipMemcpyDeviceToHost, st));
  if (out_center)
    SOGM_HIP_CHECK(hipMemcpyAsync(out_center, c->d_poses + 3 * agent, 3 * sizeof(float), hipMemcpyDeviceToHost, st));
  SOGM_HIP_CHECK(hipStreamSynchronize(st));
  return SOGM_OK;
}

int sogm_traj_eval(const SogmTrajRecord *records, int n, const double *t, double *out_pva,
                   int32_t *out_valid, void *stream) {
  if (!records || !t || !out_pva || !out_valid || n < 0) return SOGM_ERR_INVALID_ARG;
  if (n == 0) return SOGM_OK;
  hipLaunchKernelGGL(k_traj_eval, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, records, n,
                     t, out_pva, out_valid);
  SOGM_HIP_CHECK(hipGetLastError());
  return SOGM_OK;
}

int sogm_tick_inputs(const SogmTrajRecord *own_records, int n, double stamp, double replan_start_offset,
                     double *hover_inout, double *out_now, double *out_t_start, double *out_pva, float *out_poses,
                     void *stream) {
  if (!own_records || !hover_inout || !out_now || !out_t_start || !out_pva || !out_poses || n < 0)
    return SOGM_ERR_INVALID_ARG;
  if (n == 0) return SOGM_OK;
  hipLaunchKernelGGL(k_tick_inputs, dim3(n), dim3(64), 0, (hipStream_t)stream, own_records, n, stamp,
                     replan_start_offset, hover_inout, out_now, out_t_start, out_pva, out_poses);
  SOGM_HIP_CHECK(hipGetLastError());
  return SOGM_OK;
}

int sogm_merge_latest(const SogmTrajRecord *new_records, const int32_t *ok, SogmTrajRecord *own_inout,
                      SogmTrajRecord *all_or_null, int n, void *stream) {
  if (!new_records || !ok || !own_inout || n < 0) return SOGM_ERR_INVALID_ARG;
  if (n == 0) return SOGM_OK;
  const long long lanes = (long long)n * (long long)(sizeof(SogmTrajRecord) / 16);
  hipLaunchKernelGGL(k_merge_latest, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     new_records, ok, own_inout, all_or_null, n);
  SOGM_HIP_CHECK(hipGetLastError());
  return SOGM_OK;
}

int sogm_traj_safe(sogm_ctx *c, const SogmTrajRecord *records, const double *t_now, double check_duration,
                   int32_t *out_safe, void *stream) {
  if (!c || !records || !t_now || !out_safe) return SOGM_ERR_INVALID_ARG;
  if (!c->updated) return SOGM_ERR_STATE;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  if (int rc = sogm::join_update(c, (hipStream_t)stream)) return rc;
  hipLaunchKernelGGL(k_traj_safe, dim3((c->n_agents + 63) / 64), dim3(64), 0, (hipStream_t)stream, view_of(c),
                     records, t_now, check_duration, out_safe);
  SOGM_HIP_CHECK(hipGetLastError());
  return SOGM_OK;
}

int sogm_query_clear(sogm_ctx *c, const int32_t *agent_idx, const double *pos_xyz, const double *t,
                     int t_is_index, int n_q, int8_t *out, void *stream) {
  if (!c || !agent_idx || !pos_xyz || !t || !out || n_q < 0) return SOGM_ERR_INVALID_ARG;
  if (!c->updated) return SOGM_ERR_STATE;
  if (n_q == 0) return SOGM_OK;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  if (int rc = sogm::join_update(c, (hipStream_t)stream)) return rc;
  hipLaunchKernelGGL(k_query_clear, dim3((n_q + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     view_of(c), agent_idx, pos_xyz, t, t_is_index, n_q, out);
  SOGM_HIP_CHECK(hipGetLastError());
  return SOGM_OK;
}

int sogm_obstacle_points(sogm_ctx *c, const int32_t *agent_idx, const double *box_lo,
                         const double *box_hi, const double *t0, const double *t1, int n_b,
                         double *out_pts, int32_t *out_counts, int cap, void *stream) {
  if (!c || !agent_idx || !box_lo || !box_hi || !t0 || !t1 || !out_pts || !out_counts ||
      n_b < 0 || cap <= 0)
    return SOGM_ERR_INVALID_ARG;
  if (!c->updated) return SOGM_ERR_STATE;
  if (n_b == 0) return SOGM_OK;
  SOGM_HIP_CHECK(hipSetDevice(c->device));
  if (int rc = sogm::join_update(c, (hipStream_t)stream)) return rc;
  hipLaunchKernelGGL(k_obstacle_points, dim3(n_b), dim3(256), 0, (hipStream_t)stream, view_of(c),
                     agent_idx, box_lo, box_hi, t0, t1, out_pts, out_counts, cap);
  SOGM_HIP_CHECK(hipGetLastError());
  return SOGM_OK;
}

}  // extern "C"
