// sogm_corridor.hip — batched safe-corridor generation (FIRI polytopes) on the SOGM, gfx950.
//
// Reference: the corridor stage of FakeBaselinePlanner::replan / BaselinePlanner::replan
// (plan_manager/src/baseline_fake.cpp:300-414, baseline.cpp:296-403) =
//   getObstaclePoints (plan_env/src/map.cpp:480-518 / risk_base.cpp:295-337)
//   firi::firi + maxVolInsEllipsoid + costMVIE (plan_manager/include/sfc_gen/firi.hpp:44-365)
//   lbfgs::lbfgs_optimize + Lewis-Overton line search (plan_manager/include/sfc_gen/lbfgs.hpp)
//   sdlp::linprog<3|4> (traj_utils/include/traj_utils/sdlp.hpp) — here Seidel's published algorithm
//   ShrinkCorridor / checkCorridorValidity / checkCorridorIntersect / checkGoalReachability.
//
// Mapping to CDNA4: one workgroup (one 64-lane wave) per (agent, path segment) — 7 segments x 128
// agents = 896 independent problems fill the 256 CUs.  Inside a workgroup the point-cloud work is
// lane-parallel (obstacle-point extraction with an order-preserving wave scan, ellipsoid-frame
// transform, tangent planes, the greedy plane selection with a wave arg-min that breaks ties by
// index exactly like the reference's sequential scan); the tiny dense solves (4-D LP, 9-variable
// L-BFGS, 3x3 Jacobi) run on lane 0 with their working set in LDS.  A second small kernel per
// agent does the sequential bookkeeping (first invalid corridor, adjacent intersections, goal
// projection).  All fp64, operation order identical to the CPU oracle (-ffp-contract=off,
// include/sogm_detmath.h for log) so polytopes match bit for bit.  No dense contraction -> no MFMA.
#include <hip/hip_runtime.h>

#include <cfloat>

#include "../../include/sogm_detmath.h"
#include "sogm_planner.hpp"

namespace sogm {
namespace {

#define LP_MAX_ROWS 152  // 2 * max_faces(64) + 2 * 4 box rows = 136, plus head-room; keeps the segment kernel at 4 workgroups per CU (LDS <= 40 KB)
#define LP_WORK_DOUBLES (14 * LP_MAX_ROWS)
#define FIRI_MAX_H 128  // planes selected before truncation to max_faces

__device__ const double LP_BOX  = 1.0e4;
__device__ const double LP_BIG  = 1.0e7;
__device__ const double LP_TOL  = 1.0e-10;
__device__ const double LP_TINY = 1.0e-12;

__device__ inline double dot3(const double *a, const double *b) {
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
__device__ inline double dabs(double x) { return x < 0 ? -x : x; }

// ---------------------------------------------------------------------------------------------
// Seidel LP (same algorithm and operation order as oracle/lp_oracle.cpp)
// ---------------------------------------------------------------------------------------------
template <int D>
struct Seidel {
  __device__ static bool solve(const double *a, const double *b, int m, const double *c, double *x,
                               double *work) {
    for (int j = 0; j < D; ++j) x[j] = c[j] > 0 ? -LP_BIG : (c[j] < 0 ? LP_BIG : 0.0);
    double *sa = work;
    double *sb = work + LP_MAX_ROWS * (D - 1);
    for (int i = 0; i < m; ++i) {
      const double *ai = a + i * D;
      double        v  = 0;
      for (int j = 0; j < D; ++j) v += ai[j] * x[j];
      if (v <= b[i] + LP_TOL) continue;
      int    k  = 0;
      double mx = dabs(ai[0]);
      for (int j = 1; j < D; ++j)
        if (dabs(ai[j]) > mx) {
          mx = dabs(ai[j]);
          k  = j;
        }
      if (mx < LP_TINY) return false;
      const double inv = 1.0 / ai[k];
      for (int r = 0; r < i; ++r) {
        const double *ar = a + r * D;
        const double  f  = ar[k] * inv;
        int           q  = 0;
        for (int j = 0; j < D; ++j)
          if (j != k) sa[r * (D - 1) + q++] = ar[j] - f * ai[j];
        sb[r] = b[r] - f * b[i];
      }
      double cc[D - 1];
      {
        const double f = c[k] * inv;
        int          q = 0;
        for (int j = 0; j < D; ++j)
          if (j != k) cc[q++] = c[j] - f * ai[j];
      }
      double xs[D - 1];
      if (!Seidel<D - 1>::solve(sa, sb, i, cc, xs, work + LP_MAX_ROWS * D)) return false;
      double acc = b[i];
      int    q   = 0;
      for (int j = 0; j < D; ++j)
        if (j != k) {
          x[j] = xs[q++];
          acc -= ai[j] * x[j];
        }
      x[k] = acc * inv;
    }
    return true;
  }
};
template <>
struct Seidel<1> {
  __device__ static bool solve(const double *a, const double *b, int m, const double *c, double *x,
                               double *) {
    double lo = -LP_BIG, hi = LP_BIG;
    for (int i = 0; i < m; ++i) {
      if (a[i] > LP_TINY) {
        const double v = b[i] / a[i];
        if (v < hi) hi = v;
      } else if (a[i] < -LP_TINY) {
        const double v = b[i] / a[i];
        if (v > lo) lo = v;
      } else if (b[i] < -LP_TOL) {
        return false;
      }
    }
    if (lo > hi + LP_TOL) return false;
    if (lo > hi) lo = hi = 0.5 * (lo + hi);
    if (c[0] > 0)
      x[0] = lo;
    else if (c[0] < 0)
      x[0] = hi;
    else
      x[0] = lo > 0 ? lo : (hi < 0 ? hi : 0.0);
    return true;
  }
};

// min c^T x s.t. A[i][0..D) x <= rhs[i]  (A row-major, stride D).
// work: LP_WORK_DOUBLES doubles (LDS), perm: LP_MAX_ROWS ints.  Returns +inf infeasible, -inf
// unbounded (solution on the 1e4 box), else the minimum.
template <int D>
__device__ __noinline__ double linprog(const double *c, int m, const double *A, const double *rhsv,
                                       double *x, double *work, int *perm) {
  for (int j = 0; j < D; ++j) x[j] = 0.0;
  if (m <= 0) {
    double mx = 0;
    for (int j = 0; j < D; ++j) mx = dabs(c[j]) > mx ? dabs(c[j]) : mx;
    return mx > 0.0 ? -INFINITY : 0.0;
  }
  const int M  = m + 2 * D;
  double   *a  = work;                    // [LP_MAX_ROWS][D]
  double   *bb = work + LP_MAX_ROWS * D;  // [LP_MAX_ROWS]
  for (int i = 0; i < 2 * D; ++i)
    for (int j = 0; j < D; ++j) a[i * D + j] = 0.0;
  for (int j = 0; j < D; ++j) {
    a[(2 * j) * D + j]     = 1.0;
    bb[2 * j]              = LP_BOX;
    a[(2 * j + 1) * D + j] = -1.0;
    bb[2 * j + 1]          = LP_BOX;
  }
  for (int i = 0; i < m; ++i) perm[i] = i;
  unsigned long long s = 0x9E3779B97F4A7C15ULL;
  for (int i = m - 1; i > 0; --i) {
    s           = s * 6364136223846793005ULL + 1442695040888963407ULL;
    const int j = (int)((s >> 33) % (unsigned long long)(i + 1));
    const int t = perm[i];
    perm[i]     = perm[j];
    perm[j]     = t;
  }
  for (int i = 0; i < m; ++i) {
    const double *src = A + perm[i] * D;
    const double  rhs = rhsv[perm[i]];
    double        nn  = 0;
    for (int j = 0; j < D; ++j) nn += src[j] * src[j];
    nn          = sogm_det::sqrt_rn(nn);
    double *dst = a + (2 * D + i) * D;
    if (nn > 0) {
      for (int j = 0; j < D; ++j) dst[j] = src[j] / nn;
      bb[2 * D + i] = rhs / nn;
    } else {
      for (int j = 0; j < D; ++j) dst[j] = 0;
      bb[2 * D + i] = rhs;
    }
  }
  double xs[D];
  if (!Seidel<D>::solve(a, bb, M, c, xs, work + LP_MAX_ROWS * (D + 1))) return INFINITY;
  for (int j = 0; j < D; ++j) x[j] = xs[j];
  for (int j = 0; j < D; ++j)
    if (dabs(xs[j]) > 0.99 * LP_BOX) return -INFINITY;
  double v = 0;
  for (int j = 0; j < D; ++j) v += c[j] * xs[j];
  return v;
}

// ---------------------------------------------------------------------------------------------
// The same Seidel LP executed by the whole wave (all 64 lanes call it with identical arguments; a, b, work
// live in LDS).  Arithmetic and decisions are those of Seidel<D>::solve above: "the next violated
// constraint" is found 64 constraints at a time with a ballot (x does not change between violations, so the
// first set bit IS the sequential scan's hit), the reduced rows of the (D-1)-dimensional sub-problem are
// built one per lane, and the 1-D base case is a min / max / any reduction.  Same results, bit for bit.
// ---------------------------------------------------------------------------------------------
__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int D>
struct SeidelW {
  // c and x are caller-side REGISTER arrays: every access below uses a compile-time index (select chains pick
  // the eliminated coordinate k), so nothing is demoted to scratch memory.
  __device__ static bool solve(const double *a, const double *b, int m, const double *c, double *x,
                               double *work) {
    const int lane = threadIdx.x & 63;
    double    cv[D];  // the objective by value: no conditional load from the caller's array survives
#pragma unroll
    for (int j = 0; j < D; ++j) cv[j] = c[j];
#pragma unroll
    for (int j = 0; j < D; ++j) x[j] = cv[j] > 0 ? -LP_BIG : (cv[j] < 0 ? LP_BIG : 0.0);
    double *sa = work;
    double *sb = work + LP_MAX_ROWS * (D - 1);
    int     i  = 0;
    while (i < m) {
      int found = -1;
      for (int base = i; base < m; base += 64) {
        const int r    = base + lane;
        bool      viol = false;
        if (r < m) {
          const double *ar = a + r * D;
          double        v  = 0;
#pragma unroll
          for (int j = 0; j < D; ++j) v += ar[j] * x[j];
          viol = !(v <= b[r] + LP_TOL);
        }
        const unsigned long long mk = __ballot(viol);
        if (mk) {
          found = base + __ffsll((long long)mk) - 1;
          break;
        }
      }
      if (found < 0) break;
      i                = found;
      const double *ai = a + i * D;
      double        av[D];
#pragma unroll
      for (int j = 0; j < D; ++j) av[j] = ai[j];
      int    k  = 0;
      double mx = dabs(av[0]), ak = av[0], ck = cv[0];
#pragma unroll
      for (int j = 1; j < D; ++j)
        if (dabs(av[j]) > mx) {
          mx = dabs(av[j]);
          k  = j;
          ak = av[j];
          ck = cv[j];
        }
      if (mx < LP_TINY) return false;
      const double inv = 1.0 / ak;
      for (int r = lane; r < i; r += 64) {
        const double *ar = a + r * D;
        const double  f  = ar[k] * inv;
#pragma unroll
        for (int q = 0; q < D - 1; ++q) {
          const int j          = q < k ? q : q + 1;
          sa[r * (D - 1) + q] = ar[j] - f * ai[j];
        }
        sb[r] = b[r] - f * b[i];
      }
      wave_lds_sync();
      double cc[D - 1];
      {
        const double f = ck * inv;
#pragma unroll
        for (int q = 0; q < D - 1; ++q) {
          const double cj = q < k ? cv[q] : cv[q + 1];
          const double aj = q < k ? av[q] : av[q + 1];
          cc[q]           = cj - f * aj;
        }
      }
      double xs[D - 1];
      if (!SeidelW<D - 1>::solve(sa, sb, i, cc, xs, work + LP_MAX_ROWS * D)) return false;
      double acc = b[i];
      double xv[D];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        // value of coordinate j of the lifted point when j != k: xs[j] below k, xs[j - 1] above
        const double lo = j < D - 1 ? xs[j < D - 1 ? j : 0] : 0.0;
        const double hi = j > 0 ? xs[j > 0 ? j - 1 : 0] : 0.0;
        const double v  = j < k ? lo : hi;
        const double na = acc - av[j] * v;
        acc             = j != k ? na : acc;
        xv[j]           = v;
      }
      const double xk = acc * inv;
#pragma unroll
      for (int j = 0; j < D; ++j) x[j] = j == k ? xk : xv[j];
      wave_lds_sync();  // the sub-problem arrays are rebuilt by the next violation
      ++i;
    }
    return true;
  }
};
template <>
struct SeidelW<1> {
  __device__ static bool solve(const double *a, const double *b, int m, const double *c, double *x,
                               double *) {
    const int lane = threadIdx.x & 63;
    double    lo = -LP_BIG, hi = LP_BIG;
    bool      bad = false;
    for (int i = lane; i < m; i += 64) {
      if (a[i] > LP_TINY) {
        const double v = b[i] / a[i];
        if (v < hi) hi = v;
      } else if (a[i] < -LP_TINY) {
        const double v = b[i] / a[i];
        if (v > lo) lo = v;
      } else if (b[i] < -LP_TOL) {
        bad = true;
      }
    }
    for (int d = 32; d >= 1; d >>= 1) {
      const double oh = __shfl_xor(hi, d, 64), ol = __shfl_xor(lo, d, 64);
      hi = oh < hi ? oh : hi;
      lo = ol > lo ? ol : lo;
    }
    if (__ballot(bad)) return false;
    if (lo > hi + LP_TOL) return false;
    if (lo > hi) lo = hi = 0.5 * (lo + hi);
    if (c[0] > 0)
      x[0] = lo;
    else if (c[0] < 0)
      x[0] = hi;
    else
      x[0] = lo > 0 ? lo : (hi < 0 ? hi : 0.0);
    return true;
  }
};
// linprog<D> for a whole wave (same row permutation, normalisation and return convention)
template <int D>
__device__ __forceinline__ double linprog_wave(const double *c, int m, const double *A, const double *rhsv,
                                            double *x, double *work, int *perm) {
  const int lane = threadIdx.x & 63;
  for (int j = 0; j < D; ++j) x[j] = 0.0;
  if (m <= 0) {
    double mx = 0;
    for (int j = 0; j < D; ++j) mx = dabs(c[j]) > mx ? dabs(c[j]) : mx;
    return mx > 0.0 ? -INFINITY : 0.0;
  }
  const int M  = m + 2 * D;
  double   *a  = work;
  double   *bb = work + LP_MAX_ROWS * D;
  if (lane == 0) {
    for (int i = 0; i < 2 * D; ++i)
      for (int j = 0; j < D; ++j) a[i * D + j] = 0.0;
    for (int j = 0; j < D; ++j) {
      a[(2 * j) * D + j]     = 1.0;
      bb[2 * j]              = LP_BOX;
      a[(2 * j + 1) * D + j] = -1.0;
      bb[2 * j + 1]          = LP_BOX;
    }
    for (int i = 0; i < m; ++i) perm[i] = i;
    unsigned long long s = 0x9E3779B97F4A7C15ULL;
    for (int i = m - 1; i > 0; --i) {
      s           = s * 6364136223846793005ULL + 1442695040888963407ULL;
      const int j = (int)((s >> 33) % (unsigned long long)(i + 1));
      const int t = perm[i];
      perm[i]     = perm[j];
      perm[j]     = t;
    }
  }
  wave_lds_sync();
  for (int i = lane; i < m; i += 64) {
    const double *src = A + perm[i] * D;
    const double  rhs = rhsv[perm[i]];
    double        nn  = 0;
    for (int j = 0; j < D; ++j) nn += src[j] * src[j];
    nn          = sogm_det::sqrt_rn(nn);
    double *dst = a + (2 * D + i) * D;
    if (nn > 0) {
      for (int j = 0; j < D; ++j) dst[j] = src[j] / nn;
      bb[2 * D + i] = rhs / nn;
    } else {
      for (int j = 0; j < D; ++j) dst[j] = 0;
      bb[2 * D + i] = rhs;
    }
  }
  wave_lds_sync();
  double xs[D];
  if (!SeidelW<D>::solve(a, bb, M, c, xs, work + LP_MAX_ROWS * (D + 1))) return INFINITY;
  for (int j = 0; j < D; ++j) x[j] = xs[j];
  for (int j = 0; j < D; ++j)
    if (dabs(xs[j]) > 0.99 * LP_BOX) return -INFINITY;
  double v = 0;
  for (int j = 0; j < D; ++j) v += c[j] * xs[j];
  return v;
}

// ---------------------------------------------------------------------------------------------
// MVIE (firi.hpp:44-236) — lane 0 only
// ---------------------------------------------------------------------------------------------
__device__ inline bool smoothedL1(double mu, double x, double &f, double &df) {
  if (x < 0.0) return false;
  if (x > mu) {
    f  = x - 0.5 * mu;
    df = 1.0;
    return true;
  }
  const double xdmu = x / mu, sqrxdmu = xdmu * xdmu, mumxd2 = mu - 0.5 * x;
  f  = mumxd2 * sqrxdmu * xdmu;
  df = sqrxdmu * ((-0.5) * xdmu + 3.0 * mumxd2 / mu);
  return true;
}

struct MvieData {
  int    M;
  double smoothEps, penaltyWt;
  // this lane's faces (lane and lane + 64); a face beyond M is flagged off
  double a0[3], a1[3];
  bool   on0, on1;
};

// 64-lane butterfly sum: every lane ends with the same total; the association order
// ((l, l^32), (.., ^16), ...) is what oracle/corridor_oracle.cpp::tree_sum64 replays.
// Levels 32 and 16 cross the 16-lane rows (ds_bpermute); after them every lane of a class l mod 16 holds
// the same value, so levels 8 and 4 can take their partner with a DPP row rotate ((l + 8) mod 16 is in
// class (l mod 16) ^ 8, likewise for 4) and levels 2, 1 with DPP quad permutes — same pairs, same
// (commutative) additions, no LDS crossbar round trip on four of the six levels.
template <int CTRL>
__device__ inline double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo     = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi     = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ inline double bfly_sum(double v) {
  v += __shfl_xor(v, 32, 64);
  v += __shfl_xor(v, 16, 64);
  v += dpp_f64<0x128>(v);  // row_ror:8
  v += dpp_f64<0x124>(v);  // row_ror:4
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  return v;
}

// The ten partial sums of costMVIE reduced together, stage by stage (ten independent chains per stage: no DPP
// hazard stalls, no serial bpermute round trips).  Lanes >= M hold exact zeros, so for M <= 16 the two cross-row
// stages only add zeros to lanes 0..15 and are skipped; lane 0's value (the same in every lane of the full
// butterfly, additions being commutative) is broadcast instead.  Same association order, same bits.
__device__ inline double first_lane_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo     = __builtin_amdgcn_readfirstlane(lo);
  hi     = __builtin_amdgcn_readfirstlane(hi);
  return __hiloint2double(hi, lo);
}
__device__ inline void bfly_sum10(double acc[10], int M) {
  if (M > 16) {  // wave-uniform
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] += __shfl_xor(acc[k], 32, 64);
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] += __shfl_xor(acc[k], 16, 64);
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] += dpp_f64<0x128>(acc[k]);  // row_ror:8
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] += dpp_f64<0x124>(acc[k]);  // row_ror:4
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] += dpp_f64<0x4E>(acc[k]);  // quad_perm [2,3,0,1]
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] += dpp_f64<0xB1>(acc[k]);  // quad_perm [1,0,3,2]
  if (M <= 16) {
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = first_lane_f64(acc[k]);
  }
}

#ifdef SOGM_PROFILE_MVIE
__device__ unsigned long long g_mvie_prof[2];  // profiling build only: ticks (100 MHz) and calls of costMVIE
#endif
// one face's contribution to cost and gradient (firi.hpp:105-122)
__device__ inline void mvie_face(const double a[3], const double L[3][3], const double *p,
                                 double smoothEps, double acc[10]) {
  double AL[3];
  for (int j = 0; j < 3; ++j) AL[j] = (a[0] * L[0][j] + a[1] * L[1][j]) + a[2] * L[2][j];
  const double normAL = sogm_det::sqrt_rn((AL[0] * AL[0] + AL[1] * AL[1]) + AL[2] * AL[2]);
  const double adj[3] = {AL[0] / normAL, AL[1] / normAL, AL[2] / normAL};
  const double Ap     = (a[0] * p[0] + a[1] * p[1]) + a[2] * p[2];
  const double viola  = (normAL + Ap) - 1.0;
  double       c, dc;
  if (smoothedL1(smoothEps, viola, c, dc)) {
    acc[0] += c;
    const double vec[3] = {dc * a[0], dc * a[1], dc * a[2]};
    for (int j = 0; j < 3; ++j) acc[1 + j] += vec[j];
    for (int j = 0; j < 3; ++j) acc[4 + j] += adj[j] * vec[j];
    acc[7] += adj[0] * vec[1];
    acc[8] += adj[1] * vec[2];
    acc[9] += adj[0] * vec[2];
  }
}

// costMVIE (firi.hpp:74-140), evaluated by the whole wave: one face per lane (two when M > 64),
// per-lane partial sums, butterfly reduction.  Every lane returns the same cost and gradient.
__device__ __forceinline__ double costMVIE(const MvieData &D, const double *x, double *g) {
  const double *p = x, *rtd = x + 3, *cde = x + 6;
  double       *gdp = g, *gdrtd = g + 3, *gdcde = g + 6;
  double L[3][3];
  L[0][0] = rtd[0] * rtd[0] + DBL_EPSILON;
  L[0][1] = 0.0;
  L[0][2] = 0.0;
  L[1][0] = cde[0];
  L[1][1] = rtd[1] * rtd[1] + DBL_EPSILON;
  L[1][2] = 0.0;
  L[2][0] = cde[2];
  L[2][1] = cde[1];
  L[2][2] = rtd[2] * rtd[2] + DBL_EPSILON;
  double acc[10];
  for (int k = 0; k < 10; ++k) acc[k] = 0.0;
  if (D.on0) mvie_face(D.a0, L, p, D.smoothEps, acc);
  if (D.on1) mvie_face(D.a1, L, p, D.smoothEps, acc);
  bfly_sum10(acc, D.M);
  double cost = acc[0];
  for (int j = 0; j < 3; ++j) {
    gdp[j]   = acc[1 + j];
    gdrtd[j] = acc[4 + j];
    gdcde[j] = acc[7 + j];
  }
  cost *= D.penaltyWt;
  for (int j = 0; j < 3; ++j) {
    gdp[j] *= D.penaltyWt;
    gdrtd[j] *= D.penaltyWt;
    gdcde[j] *= D.penaltyWt;
  }
  cost -= sogm_det::log(L[0][0]) + sogm_det::log(L[1][1]) + sogm_det::log(L[2][2]);
  gdrtd[0] -= 1.0 / L[0][0];
  gdrtd[1] -= 1.0 / L[1][1];
  gdrtd[2] -= 1.0 / L[2][2];
  gdrtd[0] *= 2.0 * rtd[0];
  gdrtd[1] *= 2.0 * rtd[1];
  gdrtd[2] *= 2.0 * rtd[2];
  return cost;
}

__device__ inline double dotn9(const double *a, const double *b) {
  double s = 0;
  for (int i = 0; i < 9; ++i) s += a[i] * b[i];
  return s;
}
__device__ inline double ninf9(const double *v) {
  double mx = 0;
  for (int i = 0; i < 9; ++i) mx = dabs(v[i]) > mx ? dabs(v[i]) : mx;
  return mx;
}

__device__ __forceinline__ int lineSearchLO(const MvieData &D, double *x, double &f, double *g, double &stp,
                            const double *s, const double *xp, const double *gp, double stpmin,
                            double stpmax) {
  const double f_dec = 1.0e-4, s_curv = 0.9, machine_prec = 1.0e-16;
  const int    max_linesearch = 64;
  int          count = 0;
  bool         brackt = false, touched = false;
  double       mu = 0.0, nu = stpmax;
  if (!(stp > 0.0)) return -1;
  const double dginit = dotn9(gp, s);
  if (0.0 < dginit) return -2;
  const double finit = f, dgtest = f_dec * dginit, dstest = s_curv * dginit;
  while (true) {
    for (int i = 0; i < 9; ++i) x[i] = xp[i] + stp * s[i];
#ifdef SOGM_PROFILE_MVIE
    const long long tc0 = wall_clock64();
#endif
    f = costMVIE(D, x, g);
#ifdef SOGM_PROFILE_MVIE
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&g_mvie_prof[0], (unsigned long long)(wall_clock64() - tc0));
      atomicAdd(&g_mvie_prof[1], 1ull);
    }
#endif
    ++count;
    if (f != f || f == INFINITY || f == -INFINITY) return -3;
    if (f > finit + stp * dgtest) {
      nu     = stp;
      brackt = true;
    } else {
      if (dotn9(g, s) < dstest)
        mu = stp;
      else
        return count;
    }
    if (max_linesearch <= count) return -4;
    if (brackt && (nu - mu) < machine_prec * nu) return -5;
    if (brackt)
      stp = 0.5 * (mu + nu);
    else
      stp *= 2.0;
    if (stp < stpmin) return -6;
    if (stp > stpmax) {
      if (touched) return -7;
      touched = true;
      stp     = stpmax;
    }
  }
}

// L-BFGS (lbfgs.hpp lbfgs_optimize with the parameters of firi.hpp:191-199), executed replicated
// by every lane of the wave (uniform control flow; only costMVIE is lane-parallel).  Everything that
// is indexed dynamically lives in LDS with lane 0 as the single writer, so nothing spills to scratch:
//   lm[0..162) = s history, lm[162..324) = y history, lm[340..358) = alpha, lm[358..376) = y.s
#define LBFGS_FENCE()                                        \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)
__device__ __forceinline__ int lbfgsMVIE(const MvieData &D, double *x, double *lm, int *n_iter, int *n_eval) {
  const int    n = 9, m = 18, past = 3;
  const double g_epsilon = 0.0, delta = 1.0e-7, min_step = 1.0e-32, max_step = 1.0e+20,
               cautious = 1.0e-6;
  double       xp[9], g[9], gp[9], d[9];
  double       pf0 = 0, pf1 = 0, pf2 = 0;  // pf[k % 3]
  double      *lm_s = lm, *lm_y = lm + m * n, *lm_alpha = lm + 2 * m * n + 16,
              *lm_ys = lm + 2 * m * n + 16 + m;
  const bool   writer = (threadIdx.x & 63) == 0;
  if (writer) {
    for (int i = 0; i < 2 * m * n; ++i) lm[i] = 0;
    for (int i = 0; i < 2 * m; ++i) lm_alpha[i] = 0;
  }
  LBFGS_FENCE();
  double fx = costMVIE(D, x, g);
  pf0       = fx;
  for (int i = 0; i < n; ++i) d[i] = -g[i];
  int          ret;
  const double xn0 = ninf9(x);
  if (ninf9(g) / (1.0 > xn0 ? 1.0 : xn0) < g_epsilon) {
    ret = 0;
  } else {
    double step = 1.0 / sogm_det::sqrt_rn(dotn9(d, d));
    int    k = 1, end = 0, bound = 0;
    while (true) {
      for (int i = 0; i < n; ++i) {
        xp[i] = x[i];
        gp[i] = g[i];
      }
      const int ls = lineSearchLO(D, x, fx, g, step, d, xp, gp, min_step, max_step);
      *n_iter += 1;
      *n_eval += ls > 0 ? ls : 0;
      if (ls < 0) {
        for (int i = 0; i < n; ++i) {
          x[i] = xp[i];
          g[i] = gp[i];
        }
        ret = ls;
        break;
      }
      const double xn = ninf9(x);
      if (ninf9(g) / (1.0 > xn ? 1.0 : xn) < g_epsilon) {
        ret = 0;
        break;
      }
      const int km = k % past;
      if (past <= k) {
        const double afx  = dabs(fx);
        const double pfk  = km == 0 ? pf0 : (km == 1 ? pf1 : pf2);
        const double rate = dabs(pfk - fx) / (1.0 > afx ? 1.0 : afx);
        if (rate < delta) {
          ret = 1;
          break;
        }
      }
      if (km == 0) pf0 = fx;
      else if (km == 1) pf1 = fx;
      else pf2 = fx;
      ++k;
      double *se = lm_s + end * n, *ye = lm_y + end * n;
      double  sv[9], yv[9];
      for (int i = 0; i < n; ++i) {
        sv[i] = x[i] - xp[i];
        yv[i] = g[i] - gp[i];
      }
      const double ys = dotn9(yv, sv);
      const double yy = dotn9(yv, yv);
      if (writer) {
        for (int i = 0; i < n; ++i) {
          se[i] = sv[i];
          ye[i] = yv[i];
        }
        lm_ys[end] = ys;
      }
      LBFGS_FENCE();
      for (int i = 0; i < n; ++i) d[i] = -g[i];
      const double cau = dotn9(sv, sv) * sogm_det::sqrt_rn(dotn9(gp, gp)) * cautious;
      if (ys > cau) {
        ++bound;
        bound = m < bound ? m : bound;
        end   = (end + 1) % m;
        int j = end;
        for (int i = 0; i < bound; ++i) {
          j                  = (j + m - 1) % m;
          const double alpha = dotn9(lm_s + j * n, d) / lm_ys[j];
          if (writer) lm_alpha[j] = alpha;
          for (int q = 0; q < n; ++q) d[q] += (-alpha) * lm_y[j * n + q];
        }
        LBFGS_FENCE();
        for (int q = 0; q < n; ++q) d[q] *= ys / yy;
        for (int i = 0; i < bound; ++i) {
          const double beta = dotn9(lm_y + j * n, d) / lm_ys[j];
          const double al   = lm_alpha[j];
          for (int q = 0; q < n; ++q) d[q] += (al - beta) * lm_s[j * n + q];
          j = (j + 1) % m;
        }
      }
      step = 1.0;
    }
  }
  return ret;
}

__device__ void jacobiEig3(double S[3][3], double V[3][3], double w[3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 32; ++sweep) {
    const double off = dabs(S[0][1]) + dabs(S[0][2]) + dabs(S[1][2]);
    const double dia = dabs(S[0][0]) + dabs(S[1][1]) + dabs(S[2][2]);
    if (off <= 1e-300 || off <= 1e-17 * dia) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (S[p][q] == 0.0) continue;
        const double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
        const double t =
            (theta >= 0 ? 1.0 : -1.0) / (dabs(theta) + sogm_det::sqrt_rn(theta * theta + 1.0));
        const double c = 1.0 / sogm_det::sqrt_rn(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double skp = S[k][p], skq = S[k][q];
          S[k][p] = c * skp - s * skq;
          S[k][q] = s * skp + c * skq;
        }
        for (int k = 0; k < 3; ++k) {
          const double spk = S[p][k], sqk = S[q][k];
          S[p][k] = c * spk - s * sqk;
          S[q][k] = s * spk + c * sqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < 3; ++i) w[i] = S[i][i];
}

// LDS scratch handed to lane-0 solvers
struct SolverScratch {
  double *lp_work;  // LP_WORK_DOUBLES
  int    *perm;     // LP_MAX_ROWS
  double *rows;     // LP_MAX_ROWS * 5   (normalised Alp rows + blp / mvie A)
  double *lm;       // 2 * 18 * 9
};

// maxVolInsEllipsoid (firi.hpp:146-236); hPoly: M x 4 in LDS.  Called by the WHOLE wave: the
// deepest-point LP and the final 3x3 SVD run on lane 0, the L-BFGS runs replicated on all lanes
// with the cost evaluated one face per lane.  R, p, r are meaningful on lane 0 only.
__device__ __forceinline__ bool maxVolInsEllipsoid(const double *hPoly, int M, double R[3][3], double p[3],
                                   double r[3], const SolverScratch &sc, long long *dbg) {
  const int lane = threadIdx.x & 63;
  double   *Alp  = sc.rows;                    // M x 4
  double   *blp  = sc.rows + LP_MAX_ROWS * 4;  // M
  double   *sh   = sc.lm + 2 * 18 * 9;         // 16 doubles of hand-off space after the history
  // deepest interior point: rows one per lane, LP solved by the whole wave
  for (int i = lane; i < M; i += 64) {
    const double *h  = hPoly + i * 4;
    const double  hn = sogm_det::sqrt_rn((h[0] * h[0] + h[1] * h[1]) + h[2] * h[2]);
    for (int j = 0; j < 3; ++j) Alp[i * 4 + j] = h[j] / hn;
    Alp[i * 4 + 3] = 1.0;
    blp[i]         = -h[3] / hn;
  }
  wave_lds_sync();
  const double clp[4] = {0, 0, 0, -1.0};
  double       xlp[4];
  const double maxdepth = -linprog_wave<4>(clp, M, Alp, blp, xlp, sc.lp_work, sc.perm);
  wave_lds_sync();
  if (lane == 0) {
    const bool   ok = !(!(maxdepth > 0.0) || maxdepth == INFINITY || maxdepth == -INFINITY);
    sh[12]          = ok ? 1.0 : 0.0;
    if (ok) {
      const double interior[3] = {xlp[0], xlp[1], xlp[2]};
      // A = Alp / (blp - Alp interior), overwriting Alp's first three columns (stride 4)
      for (int i = 0; i < M; ++i) {
        double      *a   = Alp + i * 4;
        const double den = blp[i] - ((a[0] * interior[0] + a[1] * interior[1]) + a[2] * interior[2]);
        for (int j = 0; j < 3; ++j) a[j] = a[j] / den;
      }
      double Q[3][3], L[3][3];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          Q[i][j] = (R[i][0] * (r[0] * r[0]) * R[j][0] + R[i][1] * (r[1] * r[1]) * R[j][1]) +
                    R[i][2] * (r[2] * r[2]) * R[j][2];
      // chol3d (firi.hpp:44-55)
      L[0][0] = sogm_det::sqrt_rn(Q[0][0]);
      L[1][0] = 0.5 * (Q[0][1] + Q[1][0]) / L[0][0];
      L[1][1] = sogm_det::sqrt_rn(Q[1][1] - L[1][0] * L[1][0]);
      L[2][0] = 0.5 * (Q[0][2] + Q[2][0]) / L[0][0];
      L[2][1] = (0.5 * (Q[1][2] + Q[2][1]) - L[2][0] * L[1][0]) / L[1][1];
      L[2][2] = sogm_det::sqrt_rn(Q[2][2] - L[2][0] * L[2][0] - L[2][1] * L[2][1]);
      for (int j = 0; j < 3; ++j) sh[j] = p[j] - interior[j];
      sh[3] = sogm_det::sqrt_rn(L[0][0]);
      sh[4] = sogm_det::sqrt_rn(L[1][1]);
      sh[5] = sogm_det::sqrt_rn(L[2][2]);
      sh[6] = L[1][0];
      sh[7] = L[2][1];
      sh[8] = L[2][0];
      for (int j = 0; j < 3; ++j) sh[9 + j] = interior[j];
    }
  }
  __syncthreads();
  if (sh[12] == 0.0) {
    __syncthreads();
    return false;
  }
  MvieData D;
  D.M         = M;
  D.smoothEps = 1.0e-2;
  D.penaltyWt = 1.0e+3;
  D.on0       = lane < M;
  D.on1       = lane + 64 < M;
  for (int j = 0; j < 3; ++j) {
    D.a0[j] = D.on0 ? Alp[lane * 4 + j] : 0.0;
    D.a1[j] = D.on1 ? Alp[(lane + 64) * 4 + j] : 0.0;
  }
  double x[9];
  for (int j = 0; j < 9; ++j) x[j] = sh[j];
  const double interior[3] = {sh[9], sh[10], sh[11]};
  __syncthreads();
  int             n_it = 0, n_ev = 0;
  const long long tl0 = wall_clock64();
  const int       ret = lbfgsMVIE(D, x, sc.lm, &n_it, &n_ev);
  if (dbg && lane == 0) {
    dbg[3] = n_it;
    dbg[4] = n_ev;
    dbg[9] = wall_clock64() - tl0;
  }
  if (lane == 0) {
    double L[3][3];
    for (int j = 0; j < 3; ++j) p[j] = x[j] + interior[j];
    L[0][0] = x[3] * x[3];
    L[0][1] = 0.0;
    L[0][2] = 0.0;
    L[1][0] = x[6];
    L[1][1] = x[4] * x[4];
    L[1][2] = 0.0;
    L[2][0] = x[8];
    L[2][1] = x[7];
    L[2][2] = x[5] * x[5];
    double S[3][3], V[3][3], w[3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        S[i][j] = (L[i][0] * L[j][0] + L[i][1] * L[j][1]) + L[i][2] * L[j][2];
    jacobiEig3(S, V, w);
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2 - a; ++b)
        if (w[ord[b]] < w[ord[b + 1]]) {
          const int t = ord[b];
          ord[b]      = ord[b + 1];
          ord[b + 1]  = t;
        }
    double U[3][3], Sg[3];
    for (int c = 0; c < 3; ++c) {
      Sg[c] = sogm_det::sqrt_rn(w[ord[c]] > 0 ? w[ord[c]] : 0.0);
      for (int k = 0; k < 3; ++k) U[k][c] = V[k][ord[c]];
    }
    const double det = U[0][0] * (U[1][1] * U[2][2] - U[1][2] * U[2][1]) -
                       U[0][1] * (U[1][0] * U[2][2] - U[1][2] * U[2][0]) +
                       U[0][2] * (U[1][0] * U[2][1] - U[1][1] * U[2][0]);
    if (det < 0.0) {
      for (int k = 0; k < 3; ++k) {
        R[k][0] = U[k][1];
        R[k][1] = U[k][0];
        R[k][2] = U[k][2];
      }
      r[0] = Sg[1];
      r[1] = Sg[0];
      r[2] = Sg[2];
    } else {
      for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) R[k][c] = U[k][c];
      r[0] = Sg[0];
      r[1] = Sg[1];
      r[2] = Sg[2];
    }
  }
  __syncthreads();
  return ret >= 0;
}

// checkCorridorValidity (baseline.cpp:191-204); poly rows h0 x + h1 y + h2 z + h3 <= 0; lane 0
__device__ bool corridorValid(const double *polyA, int mA, const double *polyB, int mB,
                              const SolverScratch &sc) {
  const double c[3] = {0, 0, 0};
  double       x[3];
  double      *A = sc.rows, *b = sc.rows + LP_MAX_ROWS * 4;
  for (int i = 0; i < mA + mB; ++i) {
    const double *h = i < mA ? polyA + i * 4 : polyB + (i - mA) * 4;
    A[i * 3 + 0]    = h[0];
    A[i * 3 + 1]    = h[1];
    A[i * 3 + 2]    = h[2];
    b[i]            = -h[3];
  }
  const double v = linprog<3>(c, mA + mB, A, b, x, sc.lp_work, sc.perm);
  return !(v == INFINITY || v == -INFINITY);
}

// checkGoalReachability (baseline.cpp:143-182); lane 0
__device__ bool goalReachable(const double *poly, int m, const double *start, double *goal,
                              const SolverScratch &sc) {
  if (m <= 0) return true;
  double mx = -INFINITY;
  for (int i = 0; i < m; ++i) {
    const double *h = poly + i * 4;
    const double  v = dot3(h, goal) + h[3] * 1.0;
    mx              = v > mx ? v : mx;
  }
  if (mx <= 0) return true;
  double *A = sc.rows, *b = sc.rows + LP_MAX_ROWS * 4;
  for (int i = 0; i < m; ++i) {
    const double *h = poly + i * 4;
    A[i * 3 + 0]    = h[0];
    A[i * 3 + 1]    = h[1];
    A[i * 3 + 2]    = h[2];
    b[i]            = -h[3];
  }
  double c[3] = {-goal[0] + start[0], -goal[1] + start[1], -goal[2] + start[2]};
  double gmax[3], gmin[3];
  linprog<3>(c, m, A, b, gmax, sc.lp_work, sc.perm);
  for (int j = 0; j < 3; ++j) c[j] = goal[j] - start[j];
  linprog<3>(c, m, A, b, gmin, sc.lp_work, sc.perm);
  for (int j = 0; j < 3; ++j) goal[j] = 0.5 * (gmax[j] + gmin[j]);
  return false;
}

// whole-wave versions of the two predicates (k_corridor_finalize): same rows, same LPs, solved by linprog_wave
__device__ bool corridorValidW(const double *polyA, int mA, const double *polyB, int mB,
                               const SolverScratch &sc) {
  const int    lane = threadIdx.x & 63;
  const double c[3] = {0, 0, 0};
  double       x[3];
  double      *A = sc.rows, *b = sc.rows + LP_MAX_ROWS * 4;
  for (int i = lane; i < mA + mB; i += 64) {
    const double *h = i < mA ? polyA + i * 4 : polyB + (i - mA) * 4;
    A[i * 3 + 0]    = h[0];
    A[i * 3 + 1]    = h[1];
    A[i * 3 + 2]    = h[2];
    b[i]            = -h[3];
  }
  wave_lds_sync();
  const double v = linprog_wave<3>(c, mA + mB, A, b, x, sc.lp_work, sc.perm);
  wave_lds_sync();
  return !(v == INFINITY || v == -INFINITY);
}
__device__ bool goalReachableW(const double *poly, int m, const double *start, double *goal,
                               const SolverScratch &sc) {
  const int lane = threadIdx.x & 63;
  if (m <= 0) return true;
  double mx = -INFINITY;
  for (int i = 0; i < m; ++i) {
    const double *h = poly + i * 4;
    const double  v = dot3(h, goal) + h[3] * 1.0;
    mx              = v > mx ? v : mx;
  }
  if (mx <= 0) return true;
  double *A = sc.rows, *b = sc.rows + LP_MAX_ROWS * 4;
  for (int i = lane; i < m; i += 64) {
    const double *h = poly + i * 4;
    A[i * 3 + 0]    = h[0];
    A[i * 3 + 1]    = h[1];
    A[i * 3 + 2]    = h[2];
    b[i]            = -h[3];
  }
  wave_lds_sync();
  double c[3] = {-goal[0] + start[0], -goal[1] + start[1], -goal[2] + start[2]};
  double gmax[3], gmin[3];
  linprog_wave<3>(c, m, A, b, gmax, sc.lp_work, sc.perm);
  wave_lds_sync();
  for (int j = 0; j < 3; ++j) c[j] = goal[j] - start[j];
  linprog_wave<3>(c, m, A, b, gmin, sc.lp_work, sc.perm);
  wave_lds_sync();
  for (int j = 0; j < 3; ++j) goal[j] = 0.5 * (gmax[j] + gmin[j]);
  return false;
}

// wave arg-min of (value, index): smaller value wins, ties -> smaller index
__device__ inline void wave_argmin(double &v, int &idx) {
  for (int d = 32; d >= 1; d >>= 1) {
    const double ov = __shfl_xor(v, d, 64);
    const int    oi = __shfl_xor(idx, d, 64);
    if (ov < v || (ov == v && oi < idx)) {
      v   = ov;
      idx = oi;
    }
  }
}

// Local box of segment `seg` (baseline.cpp:300-324): box[0..2] = llc, box[3..5] = lhc, w[0..5] = the two waypoints
__device__ inline void segment_box(const SogmPlannerParams &pp, const double *sp, const double *rt, int seg,
                                   double *box, double *w) {
  double w0[3], w1[3];
  for (int k = 0; k < 3; ++k) {
    w0[k] = rt[seg * 6 + k];
    w1[k] = rt[(seg + 1) * 6 + k];
  }
  if (w0[2] < 0) w0[2] = 0.1;  // baseline.cpp:314
  if (w1[2] < 0) w1[2] = 0.1;
  double lower[3]  = {-4 + sp[0], -4 + sp[1], -1 + sp[2]};
  double higher[3] = {4 + sp[0], 4 + sp[1], 1 + sp[2]};
  if (lower[2] < 0) lower[2] = 0;
  if (higher[2] > 4) higher[2] = 4;
  for (int k = 0; k < 3; ++k) {
    const double mxw = w0[k] > w1[k] ? w0[k] : w1[k];
    const double mnw = w0[k] < w1[k] ? w0[k] : w1[k];
    const double hi  = mxw + pp.init_range;
    const double lo  = mnw - pp.init_range;
    box[3 + k]       = hi < higher[k] ? hi : higher[k];  // lhc
    box[k]           = lo > lower[k] ? lo : lower[k];    // llc
    w[k]             = w0[k];
    w[3 + k]         = w1[k];
  }
}

}  // namespace

// =================================================================================================
// Kernel P: obstacle points of one (segment, agent) — map.cpp:480-518 / risk_base.cpp:295-337.
// The only corridor stage that reads the SOGM: once it has run the grid may be cleared for the next update.
// Four waves; a lane takes four consecutive cells of the box scan per step (4 x slices loads in flight per lane,
// 1024 cells per workgroup step), and an order-preserving wave scan + cross-wave offsets keep the reference's
// point order (x fastest, then y, z; a cell's time slices in ascending order).
// =================================================================================================
__global__ __launch_bounds__(256) void k_corridor_points(MapView m, SogmPlannerParams pp, CorridorWorkspace ws,
                                                        const double *__restrict__ start_pva,
                                                        const double *__restrict__ t_start,
                                                        const double *__restrict__ route,
                                                        const int32_t *__restrict__ route_len, int route_cap,
                                                        int agent0) {
  const int seg   = blockIdx.x;
  const int agent = blockIdx.y + agent0;
  const int tid   = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rl    = route_len[agent];
  const int slot  = agent * SOGM_MAX_PIECES + seg;
  if (seg >= rl - 1 || seg >= SOGM_MAX_PIECES) return;
  __shared__ double s_box[6], s_w[6];
  __shared__ int    s_wtot[4];
  const GridGeom &g    = m.g;
  const float    *pose = m.poses + agent * 3;
  const int       cap  = pp.pc_capacity;
  double         *pc   = ws.pc + (size_t)slot * cap * 3;
  if (tid == 0) segment_box(pp, start_pva + agent * 9, route + (size_t)agent * route_cap * 6, seg, s_box, s_w);
  __syncthreads();
  int N = 0;
  {
    const double stamp = m.stamps[agent];
    const double tr    = (double)g.dt;
    const double t1    = t_start[agent] + seg * pp.corridor_tau;
    const double t2    = t_start[agent] + (seg + 1) * pp.corridor_tau;
    int          js    = (int)floor((t1 - stamp) / tr);
    int          je    = (int)ceil((t2 - stamp) / tr);
    js                 = js < 0 ? 0 : js;
    js                 = js > g.T ? g.T : js;
    je                 = je > g.T ? g.T : je;
    je                 = je < 0 ? 0 : je;
    if (je > g.T - 1) je = g.T - 1;
    int lx = (int)((s_box[0] - pose[0] + g.rx) / g.res);
    int ly = (int)((s_box[1] - pose[1] + g.ry) / g.res);
    int lz = (int)((s_box[2] - pose[2] + g.rz) / g.res);
    int hx = (int)((s_box[3] - pose[0] + g.rx) / g.res);
    int hy = (int)((s_box[4] - pose[1] + g.ry) / g.res);
    int hz = (int)((s_box[5] - pose[2] + g.rz) / g.res);
    hx     = min(hx, g.L - 1);
    hy     = min(hy, g.W - 1);
    hz     = min(hz, g.H - 1);
    lx     = max(lx, 0);
    ly     = max(ly, 0);
    lz     = max(lz, 0);
    const int nx = hx - lx + 1, ny = hy - ly + 1, nz = hz - lz + 1;
    if (nx > 0 && ny > 0 && nz > 0 && js <= je) {
      const int   cells = nx * ny * nz;
      const void *grid0 = m.slab(agent, 0);
      int         base  = 0;
      for (int c0 = 0; c0 < cells; c0 += 1024) {
        int      cnt[4], vi[4];
        unsigned mask[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = c0 + 4 * tid + q;
          cnt[q]      = 0;
          mask[q]     = 0;
          vi[q]       = 0;
          if (c < cells) {
            const int x = lx + c % nx;
            const int y = ly + (c / nx) % ny;
            const int z = lz + c / (nx * ny);
            vi[q]       = x + y * g.L + z * g.L * g.W;
            for (int j = js; j <= je; ++j) {
              const float thr = g.map_kind == SOGM_MAP_FAKE ? g.risk_threshold
                                                             : g.risk_threshold - g.decay_voxel * (float)j;
              if (cell_ld(grid0, (size_t)j * g.V + vi[q], g.half) > thr) {
                ++cnt[q];
                mask[q] |= 1u << (j - js);
              }
            }
          }
        }
        const int mine = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        int       incl = mine;
        for (int d = 1; d < 64; d <<= 1) {
          const int up = __shfl_up(incl, d, 64);
          if (lane >= d) incl += up;
        }
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int t = s_wtot[w];
          woff += w < wave ? t : 0;
          total += t;
        }
        __syncthreads();  // s_wtot is rewritten by the next step
        int off = base + woff + incl - mine;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (cnt[q]) {
            float fx, fy, fz;
            g.corner_of(vi[q], pose, fx, fy, fz);
            for (int j = 0; j < 32 && (mask[q] >> j); ++j)
              if ((mask[q] >> j) & 1u) {
                if (off < cap) {
                  pc[off * 3 + 0] = (double)fx;
                  pc[off * 3 + 1] = (double)fy;
                  pc[off * 3 + 2] = (double)fz;
                }
                ++off;
              }
          }
        base += total;
      }
      N = base;
    }
  }
  if (tid == 0) ws.seg_npts[slot] = N;
}

namespace {
}  // namespace

// =================================================================================================
// Kernel A: one workgroup per (segment, agent)
// =================================================================================================
__global__ __launch_bounds__(64) void k_corridor_segment(
    MapView m, SogmPlannerParams pp, CorridorWorkspace ws, const double *__restrict__ start_pva,
    const double *__restrict__ t_start, const double *__restrict__ route,
    const int32_t *__restrict__ route_len, int route_cap, int agent0) {
  const int seg   = blockIdx.x;
  const int agent = blockIdx.y + agent0;
  const int lane  = threadIdx.x;
  const int rl    = route_len[agent];
  const int slot  = agent * SOGM_MAX_PIECES + seg;
  if (seg >= rl - 1 || seg >= SOGM_MAX_PIECES) {
    if (lane == 0) ws.seg_state[slot] = -2;  // no such segment
    return;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double        *s_lp    = (double *)smem;                    // LP_WORK_DOUBLES
  double        *s_rows  = s_lp + LP_WORK_DOUBLES;            // LP_MAX_ROWS * 5
  double        *s_lm    = s_rows + LP_MAX_ROWS * 5;          // 324 history + 16 hand-off + 36 alpha/ys
  double        *s_fH    = s_lm + 2 * 18 * 9 + 16 + 36;       // FIRI_MAX_H * 4
  double        *s_poly  = s_fH + FIRI_MAX_H * 4;             // FIRI_MAX_H * 4
  double        *s_small = s_poly + FIRI_MAX_H * 4;           // 96 doubles of shared small state
  int           *s_perm  = (int *)(s_small + 96);             // LP_MAX_ROWS
  int           *s_int   = s_perm + LP_MAX_ROWS;              // 16 ints
  unsigned char *s_flag  = (unsigned char *)(s_int + 16);     // pc_capacity bytes
  SolverScratch  sc{s_lp, s_perm, s_rows, s_lm};

  // shared small state layout
  double *s_fwd  = s_small;       // 9  forward
  double *s_fa   = s_small + 9;   // 3  fwd_a
  double *s_fb   = s_small + 12;  // 3  fwd_b
  double *s_p    = s_small + 15;  // 3
  double *s_fh   = s_small + 18;  // 4  current plane
  double *s_bd   = s_small + 22;  // 24 bd
  double *s_fB   = s_small + 46;  // 18 forwardB
  double *s_fD   = s_small + 64;  // 6  forwardD
  double *s_dD   = s_small + 70;  // 6  distDs
  double *s_box  = s_small + 76;  // 6  llc, lhc
  double *s_w    = s_small + 82;  // 6  w0, w1

  const double   *rt    = route + (size_t)agent * route_cap * 6;
  const double   *sp    = start_pva + agent * 9;
  const int       cap   = pp.pc_capacity;
  double         *pc    = ws.pc + (size_t)slot * cap * 3;
  double         *fpc   = ws.fpc + (size_t)slot * cap * 3;
  double         *tang  = ws.tang + (size_t)slot * cap * 4;
  double         *distR = ws.distr + (size_t)slot * cap;

  long long *dbg = ws.seg_dbg + (size_t)slot * 16;
  const long long tk0 = wall_clock64();
  if (lane == 0) {
    for (int k = 0; k < 16; ++k) dbg[k] = 0;
    segment_box(pp, sp, rt, seg, s_box, s_w);
    // getInitCorridor (baseline.cpp:127-141) with the local box
    for (int i = 0; i < 24; ++i) s_bd[i] = 0;
    for (int k = 0; k < 3; ++k) {
      s_bd[k * 4 + k]       = 1.0;
      s_bd[(k + 3) * 4 + k] = -1.0;
      s_bd[k * 4 + 3]       = -s_box[3 + k];
      s_bd[(k + 3) * 4 + 3] = s_box[k];
    }
  }
  __syncthreads();

  // obstacle points: written by k_corridor_points (the only stage of the corridor generation that reads the SOGM)
  int N = ws.seg_npts[slot];
  int overflow = 0;
  if (N > cap) {
    N        = cap;
    overflow = 1;
  }
  __syncthreads();

  if (lane == 0) {
    dbg[0] = N;
    dbg[5] = wall_clock64() - tk0;
  }
  // ---------------- firi::firi (firi.hpp:238-365) ----------------
  const double epsilon = 1.0e-6;
  const int    M       = 6;
  int          nH      = 0;
  bool         seed_ok = true;
  if (lane == 0) {
    int ok = 1;
    for (int i = 0; i < M; ++i) {
      const double *h = s_bd + i * 4;
      if (dot3(h, s_w) + h[3] > 0.0 || dot3(h, s_w + 3) + h[3] > 0.0) ok = 0;
    }
    s_int[0] = ok;
  }
  __syncthreads();
  seed_ok = s_int[0] != 0;

  if (seed_ok) {
    // lane-0 private ellipsoid state (R, p, r)
    double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    double p[3]    = {0.5 * (s_w[0] + s_w[3]), 0.5 * (s_w[1] + s_w[4]), 0.5 * (s_w[2] + s_w[5])};
    double r[3]    = {1, 1, 1};
    for (int loop = 0; loop < pp.firi_iterations; ++loop) {
      if (lane == 0) {
        double forward[3][3], backward[3][3];
        for (int k = 0; k < 3; ++k)
          for (int j = 0; j < 3; ++j) {
            forward[k][j]    = (1.0 / r[k]) * R[j][k];
            backward[k][j]   = R[k][j] * r[j];
            s_fwd[k * 3 + j] = forward[k][j];
          }
        for (int i = 0; i < M; ++i) {
          const double *h = s_bd + i * 4;
          for (int j = 0; j < 3; ++j)
            s_fB[i * 3 + j] =
                (h[0] * backward[0][j] + h[1] * backward[1][j]) + h[2] * backward[2][j];
          s_fD[i] = h[3] + dot3(h, p);
        }
        const double da[3] = {s_w[0] - p[0], s_w[1] - p[1], s_w[2] - p[2]};
        const double db[3] = {s_w[3] - p[0], s_w[4] - p[1], s_w[5] - p[2]};
        for (int k = 0; k < 3; ++k) {
          s_fa[k] = dot3(forward[k], da);
          s_fb[k] = dot3(forward[k], db);
          s_p[k]  = p[k];
        }
        for (int i = 0; i < M; ++i) {
          const double *fb = s_fB + i * 3;
          s_dD[i]          = dabs(s_fD[i]) / sogm_det::sqrt_rn(dot3(fb, fb));
        }
      }
      __syncthreads();
      // per-point: ellipsoid frame, tangent plane with the a/b fix-ups (firi.hpp:274-305)
      double lmin = INFINITY;
      int    lidx = 0x7fffffff;
      {
        double fwd[9], fa[3], fb[3], pp3[3];
        for (int k = 0; k < 9; ++k) fwd[k] = s_fwd[k];
        for (int k = 0; k < 3; ++k) {
          fa[k]  = s_fa[k];
          fb[k]  = s_fb[k];
          pp3[k] = s_p[k];
        }
        for (int i = lane; i < N; i += 64) {
          const double d[3] = {pc[i * 3] - pp3[0], pc[i * 3 + 1] - pp3[1], pc[i * 3 + 2] - pp3[2]};
          double       q[3];
          for (int k = 0; k < 3; ++k) q[k] = dot3(fwd + k * 3, d);
          double t[4];
          double dr = sogm_det::sqrt_rn(dot3(q, q));
          t[3]      = -dr;
          for (int k = 0; k < 3; ++k) t[k] = q[k] / dr;
          if (dot3(t, fa) + t[3] > epsilon) {
            const double delta[3] = {q[0] - fa[0], q[1] - fa[1], q[2] - fa[2]};
            const double s        = dot3(delta, fa) / dot3(delta, delta);
            for (int k = 0; k < 3; ++k) t[k] = fa[k] - s * delta[k];
            dr   = sogm_det::sqrt_rn(dot3(t, t));
            t[3] = -dr;
            for (int k = 0; k < 3; ++k) t[k] /= dr;
          }
          if (dot3(t, fb) + t[3] > epsilon) {
            const double delta[3] = {q[0] - fb[0], q[1] - fb[1], q[2] - fb[2]};
            const double s        = dot3(delta, fb) / dot3(delta, delta);
            for (int k = 0; k < 3; ++k) t[k] = fb[k] - s * delta[k];
            dr   = sogm_det::sqrt_rn(dot3(t, t));
            t[3] = -dr;
            for (int k = 0; k < 3; ++k) t[k] /= dr;
          }
          if (dot3(t, fa) + t[3] > epsilon) {
            const double u[3] = {fa[0] - q[0], fa[1] - q[1], fa[2] - q[2]};
            const double v[3] = {fb[0] - q[0], fb[1] - q[1], fb[2] - q[2]};
            double       n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2],
                                 u[0] * v[1] - u[1] * v[0]};
            const double nn   = sogm_det::sqrt_rn(dot3(n, n));
            if (nn > 0)
              for (int k = 0; k < 3; ++k) n[k] /= nn;
            for (int k = 0; k < 3; ++k) t[k] = n[k];
            t[3]            = -dot3(t, fa);
            const double sg = t[3] > 0.0 ? -1.0 : 1.0;
            for (int k = 0; k < 4; ++k) t[k] *= sg;
          }
          for (int k = 0; k < 3; ++k) fpc[i * 3 + k] = q[k];
          for (int k = 0; k < 4; ++k) tang[i * 4 + k] = t[k];
          distR[i]  = dr;
          s_flag[i] = 1;
          if (dr < lmin) {
            lmin = dr;
            lidx = i;
          }
        }
      }
      wave_argmin(lmin, lidx);
      // greedy plane selection (firi.hpp:307-349)
      double minSqrR = lmin;
      int    pcMinId = lidx == 0x7fffffff ? 0 : lidx;
      double minSqrD = INFINITY;
      int    bdMinId = 0;
      unsigned bdFlags = 0x3f;
      for (int j = 0; j < M; ++j)
        if (s_dD[j] < minSqrD) {
          minSqrD = s_dD[j];
          bdMinId = j;
        }
      nH             = 0;
      bool completed = false;
      for (int i = 0; !completed && i < (M + N); ++i) {
        __syncthreads();
        if (lane == 0) {
          double *fh = s_fH + (nH < FIRI_MAX_H ? nH : FIRI_MAX_H - 1) * 4;
          if (minSqrD < minSqrR) {
            for (int k = 0; k < 3; ++k) fh[k] = s_fB[bdMinId * 3 + k];
            fh[3] = s_fD[bdMinId];
          } else {
            for (int k = 0; k < 4; ++k) fh[k] = tang[pcMinId * 4 + k];
            s_flag[pcMinId] = 0;
          }
          for (int k = 0; k < 4; ++k) s_fh[k] = fh[k];
        }
        if (minSqrD < minSqrR) bdFlags &= ~(1u << bdMinId);  // uniform across lanes
        __syncthreads();
        completed = true;
        minSqrD   = INFINITY;
        for (int j = 0; j < M; ++j)
          if (bdFlags & (1u << j)) {
            completed = false;
            if (minSqrD > s_dD[j]) {
              bdMinId = j;
              minSqrD = s_dD[j];
            }
          }
        const double fh0 = s_fh[0], fh1 = s_fh[1], fh2 = s_fh[2], fh3 = s_fh[3];
        double       lm = INFINITY;
        int          li = 0x7fffffff;
        int          open = 0;
        for (int j = lane; j < N; j += 64) {
          if (s_flag[j]) {
            const double *q = fpc + j * 3;
            if (((fh0 * q[0] + fh1 * q[1]) + fh2 * q[2]) + fh3 > -epsilon) {
              s_flag[j] = 0;
            } else {
              open = 1;
              if (lm > distR[j]) {
                lm = distR[j];
                li = j;
              }
            }
          }
        }
        wave_argmin(lm, li);
        if (__any(open)) completed = false;
        minSqrR = lm;
        pcMinId = li == 0x7fffffff ? 0 : li;
        ++nH;
      }
      __syncthreads();
      if (nH > FIRI_MAX_H) {
        nH       = FIRI_MAX_H;
        overflow = 1;
      }
      // hPoly = forwardH * forward, offset back by p (firi.hpp:351-355)
      for (int i = lane; i < nH; i += 64) {
        const double *fh = s_fH + i * 4;
        double        h[4];
        for (int j = 0; j < 3; ++j)
          h[j] = (fh[0] * s_fwd[0 * 3 + j] + fh[1] * s_fwd[1 * 3 + j]) + fh[2] * s_fwd[2 * 3 + j];
        h[3] = fh[3] - dot3(h, s_p);
        for (int k = 0; k < 4; ++k) s_poly[i * 4 + k] = h[k];
      }
      __syncthreads();
      if (lane == 0) {
        dbg[1 + (loop > 0)] = nH;
        dbg[6 + 2 * (loop > 0)] = wall_clock64() - tk0;
      }
      if (loop == pp.firi_iterations - 1) break;
      {
        const int       mm  = nH < LP_MAX_ROWS - 8 ? nH : LP_MAX_ROWS - 8;
        const long long tm0 = wall_clock64();
        maxVolInsEllipsoid(s_poly, mm, R, p, r, sc, dbg);
        if (lane == 0) dbg[7] = wall_clock64() - tm0;
      }
      __syncthreads();
    }
  }

  // ---------------- ShrinkCorridor + checkCorridorValidity ----------------
  int nf = seed_ok ? nH : 0;
  if (nf > pp.max_faces) {
    nf       = pp.max_faces;
    overflow = 1;
  }
  {
    const double path[3] = {s_w[3] - s_w[0], s_w[4] - s_w[1], s_w[5] - s_w[2]};
    for (int f = lane; f < nf; f += 64) {  // one face per lane
      double      *h   = s_poly + f * 4;
      const double nrm = sogm_det::sqrt_rn(dot3(h, h));
      if (pp.fake_planner) {
        const double pn = sogm_det::sqrt_rn(dot3(path, path));
        if (dot3(h, path) / nrm / pn > 0.8) continue;
        if (dabs(h[2]) / nrm > 0.8) continue;
      }
      h[3] += nrm * pp.shrink_size;
    }
  }
  wave_lds_sync();
  const bool valid = corridorValidW(s_poly, nf, nullptr, 0, sc);  // whole wave
  {
    double *out = ws.polys + (size_t)slot * pp.max_faces * 4;
    for (int i = lane; i < nf * 4; i += 64) out[i] = s_poly[i];
  }
  if (lane == 0) {
    ws.seg_nfaces[slot] = nf;
    ws.seg_state[slot]  = overflow ? -3 : (valid ? 1 : 0);
    ws.seg_npts[slot]   = N;
    dbg[10]             = wall_clock64() - tk0;
  }
}

// =================================================================================================
// Kernel B: per agent bookkeeping (baseline_fake.cpp:364-414 / baseline.cpp:362-403)
// =================================================================================================
__global__ __launch_bounds__(64) void k_corridor_finalize(
    SogmPlannerParams pp, CorridorWorkspace ws, const double *__restrict__ start_pva,
    const double *__restrict__ route, const int32_t *__restrict__ route_len, int route_cap,
    double *__restrict__ out_polys, int32_t *__restrict__ out_nfaces,
    int32_t *__restrict__ out_npoly, double *__restrict__ out_goal, int agent0) {
  const int agent = blockIdx.x + agent0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double       *s_lp   = (double *)smem;
  double       *s_rows = s_lp + LP_WORK_DOUBLES;
  int          *s_perm = (int *)(s_rows + LP_MAX_ROWS * 5);
  SolverScratch sc{s_lp, s_perm, s_rows, nullptr};
  // The sequential bookkeeping below is executed by all 64 lanes with identical data (uniform control flow);
  // the LPs inside are solved by the whole wave, outputs are written by lane 0 / copied lane-parallel.
  const bool    w0    = threadIdx.x == 0;
  const int     MF    = pp.max_faces;
  const double *polys = ws.polys + (size_t)agent * SOGM_MAX_PIECES * MF * 4;
  const int    *nfs   = ws.seg_nfaces + agent * SOGM_MAX_PIECES;
  const int    *state = ws.seg_state + agent * SOGM_MAX_PIECES;
  const double *rt    = route + (size_t)agent * route_cap * 6;
  const double *sp    = start_pva + agent * 9;
  const int     rl    = route_len[agent];
  int           npoly = 0;
  if (w0) {
    for (int i = 0; i < SOGM_MAX_PIECES; ++i) out_nfaces[agent * SOGM_MAX_PIECES + i] = 0;
    for (int i = 0; i < 6; ++i) out_goal[agent * 6 + i] = 0;
    out_npoly[agent] = 0;
  }
  if (!pp.fake_planner && rl < 2) return;
  if (rl < 1) return;
  // corridors are kept up to the first invalid one (the loop `break`s, baseline.cpp:355-358)
  for (int i = 0; i < rl - 1 && i < SOGM_MAX_PIECES; ++i) {
    if (state[i] != 1) break;
    ++npoly;
  }
  if (npoly == 0) return;
  for (int i = 0; i + 1 < npoly; ++i) {
    if (!corridorValidW(polys + (size_t)i * MF * 4, nfs[i], polys + (size_t)(i + 1) * MF * 4,
                        nfs[i + 1], sc)) {
      if (i < 2) return;
      npoly = pp.fake_planner ? i + 1 : i;
      break;
    }
  }
  if (pp.fake_planner ? npoly == 0 : npoly <= 1) return;
  double gpos[3], gvel[3];
  int    gi = npoly - 1;
  for (int k = 0; k < 3; ++k) {
    gpos[k] = rt[gi * 6 + k];
    gvel[k] = rt[gi * 6 + 3 + k];
  }
  bool do_scan = true;
  if (!pp.fake_planner)
    do_scan = !goalReachableW(polys + (size_t)(npoly - 1) * MF * 4, nfs[npoly - 1], sp, gpos, sc);
  if (do_scan) {
    for (int it = npoly - 1; it != 0; --it) {
      if (goalReachableW(polys + (size_t)it * MF * 4, nfs[it], sp, gpos, sc)) {
        npoly         = it + 1;
        const int idx = npoly - 1;
        for (int k = 0; k < 3; ++k) {
          gpos[k] = rt[idx * 6 + k];
          gvel[k] = rt[idx * 6 + 3 + k];
        }
        break;
      }
    }
  }
  for (int i = 0; i < npoly; ++i) {
    if (w0) out_nfaces[agent * SOGM_MAX_PIECES + i] = nfs[i];
    const double *src = polys + (size_t)i * MF * 4;
    double       *dst = out_polys + ((size_t)agent * SOGM_MAX_PIECES + i) * MF * 4;
    for (int k = threadIdx.x; k < nfs[i] * 4; k += 64) dst[k] = src[k];
  }
  if (w0) {
    for (int k = 0; k < 3; ++k) {
      out_goal[agent * 6 + k]     = gpos[k];
      out_goal[agent * 6 + 3 + k] = gvel[k];
    }
    out_npoly[agent] = npoly;
  }
}

size_t corridor_segment_lds(int pc_capacity) {
  return sizeof(double) * (LP_WORK_DOUBLES + LP_MAX_ROWS * 5 + 2 * 18 * 9 + 16 + 36 + 2 * FIRI_MAX_H * 4 + 96) +
         sizeof(int) * (LP_MAX_ROWS + 16) + (size_t)pc_capacity;
}

// ------------------------------------------------------------------------------------------------
// ParticleATC::isSafeAfterOpt (traj_coordinator/src/particles.cpp:223-283): one workgroup per
// (other agent's record, ego agent).  Set A = the new trajectory's control points, set B = the record's
// control points from the piece that contains t_now onwards; separable iff the feasibility LP
// n.a + d >= 1, n.b + d <= -1 (utils/separator/src/separator_glpk.cpp:75-190, zero objective) has a
// solution — solved with the same Seidel LP as the corridor checks.
// ------------------------------------------------------------------------------------------------
#define DECONFLICT_MAX_ROWS (LP_MAX_ROWS - 8)
__global__ __launch_bounds__(64) void k_safe_after_opt(const double *__restrict__ cpts,
                                                       const int32_t *__restrict__ npoly,
                                                       const SogmTrajRecord *__restrict__ rec, int n_rec,
                                                       const int32_t *__restrict__ ego_ids,
                                                       const double *__restrict__ t_now,
                                                       int32_t *__restrict__ out_safe, int agent0) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  double *s_lp   = s_dyn;                   // LP_WORK_DOUBLES
  double *s_rows = s_lp + LP_WORK_DOUBLES;  // LP_MAX_ROWS * 5
  int    *s_perm = (int *)(s_rows + LP_MAX_ROWS * 5);
  const int a = blockIdx.y + agent0, i = blockIdx.x;
  const int M = npoly[a];
  if (M <= 0) return;  // nothing optimised for this agent
  const SogmTrajRecord &r = rec[i];
  if (r.n_pieces <= 0 || r.drone_id == ego_ids[a]) return;
  double time_end = r.time_start;
  for (int k = 0; k < r.n_pieces; ++k) time_end += r.duration[k];
  const double now = t_now[a];
  if (!(r.time_start < now && now < time_end)) return;
  double t     = now - r.time_start;  // Bezier::locatePiece (bernstein.hpp:164-172)
  int    piece = r.n_pieces - 1;
  for (int k = 0; k < r.n_pieces; ++k) {
    t -= r.duration[k];
    if (t < 0) {
      piece = k;
      break;
    }
  }
  const int nA = 5 * M, nB = (r.n_pieces - piece) * 5;
  if (nA + nB > DECONFLICT_MAX_ROWS) {  // LP capacity: treated as "not separable" (oracle does the same)
    if (threadIdx.x == 0) out_safe[a] = 0;
    return;
  }
  double       *A = s_rows, *b = s_rows + LP_MAX_ROWS * 4;
  const double *ca = cpts + (size_t)a * SOGM_MAX_PIECES * 15, *cb = r.cpts + piece * 15;
  // Disjoint bounding boxes are separated by an axis-aligned plane (the LP is feasible): most pairs of
  // a swarm end here without touching the LP.  64-lane min/max over the two point sets.
  {
    double lo[6], hi[6];
    for (int k = 0; k < 6; ++k) {
      lo[k] = INFINITY;
      hi[k] = -INFINITY;
    }
    for (int q = threadIdx.x; q < nA; q += 64)
      for (int k = 0; k < 3; ++k) {
        lo[k] = fmin(lo[k], ca[q * 3 + k]);
        hi[k] = fmax(hi[k], ca[q * 3 + k]);
      }
    for (int q = threadIdx.x; q < nB; q += 64)
      for (int k = 0; k < 3; ++k) {
        lo[3 + k] = fmin(lo[3 + k], cb[q * 3 + k]);
        hi[3 + k] = fmax(hi[3 + k], cb[q * 3 + k]);
      }
    bool apart = false;
    for (int k = 0; k < 6; ++k)
      for (int d = 32; d >= 1; d >>= 1) {
        lo[k] = fmin(lo[k], __shfl_xor(lo[k], d, 64));
        hi[k] = fmax(hi[k], __shfl_xor(hi[k], d, 64));
      }
    for (int k = 0; k < 3; ++k) apart = apart || hi[k] < lo[3 + k] || hi[3 + k] < lo[k];
    if (apart) return;  // wave-uniform
    // more candidate normals (face / body diagonals, the line between the box centres), same list and
    // order as oracle/deconflict_oracle.cpp: disjoint projections = separable, no LP needed
    double dirs[11][3] = {{1, 1, 0}, {1, -1, 0}, {1, 0, 1}, {1, 0, -1}, {0, 1, 1}, {0, 1, -1},
                          {1, 1, 1}, {1, 1, -1}, {1, -1, 1}, {1, -1, -1}, {0, 0, 0}};
    for (int k = 0; k < 3; ++k) dirs[10][k] = 0.5 * (lo[3 + k] + hi[3 + k]) - 0.5 * (lo[k] + hi[k]);
#pragma unroll
    for (int q = 0; q < 11; ++q) {  // unrolled: the normals are immediates, not a scratch-memory table
      const double n0 = dirs[q][0], n1 = dirs[q][1], n2 = dirs[q][2];
      double loA = INFINITY, hiA = -INFINITY, loB = INFINITY, hiB = -INFINITY;
      for (int i = threadIdx.x; i < nA; i += 64) {
        const double v = (n0 * ca[i * 3] + n1 * ca[i * 3 + 1]) + n2 * ca[i * 3 + 2];
        loA = fmin(loA, v);
        hiA = fmax(hiA, v);
      }
      for (int i = threadIdx.x; i < nB; i += 64) {
        const double v = (n0 * cb[i * 3] + n1 * cb[i * 3 + 1]) + n2 * cb[i * 3 + 2];
        loB = fmin(loB, v);
        hiB = fmax(hiB, v);
      }
      for (int d = 32; d >= 1; d >>= 1) {
        loA = fmin(loA, __shfl_xor(loA, d, 64));
        hiA = fmax(hiA, __shfl_xor(hiA, d, 64));
        loB = fmin(loB, __shfl_xor(loB, d, 64));
        hiB = fmax(hiB, __shfl_xor(hiB, d, 64));
      }
      if (hiA < loB || hiB < loA) return;  // wave-uniform
    }
  }
  for (int q = threadIdx.x; q < nA + nB; q += 64) {
    if (q < nA) {
      A[q * 4 + 0] = -ca[q * 3 + 0];
      A[q * 4 + 1] = -ca[q * 3 + 1];
      A[q * 4 + 2] = -ca[q * 3 + 2];
      A[q * 4 + 3] = -1.0;
    } else {
      const int e  = q - nA;
      A[q * 4 + 0] = cb[e * 3 + 0];
      A[q * 4 + 1] = cb[e * 3 + 1];
      A[q * 4 + 2] = cb[e * 3 + 2];
      A[q * 4 + 3] = 1.0;
    }
    b[q] = -1.0;
  }
  __syncthreads();
  {
    const double c[4] = {0, 0, 0, 0};
    double       x[4];
    const double v = linprog_wave<4>(c, nA + nB, A, b, x, s_lp, s_perm);  // the whole wave solves the LP
    if (threadIdx.x == 0 && (v == INFINITY || v == -INFINITY)) out_safe[a] = 0;  // every writer writes 0
  }
}

__global__ void k_fill_i32(int32_t *p, int n, int32_t v, int agent0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[agent0 + i] = v;
}

int launch_deconflict(int n_agents, const double *cpts, const int32_t *npoly, const SogmTrajRecord *rec,
                      int n_rec, const int32_t *ego_ids, const double *t_now, int32_t *out_safe,
                      hipStream_t st, int agent0) {
  hipLaunchKernelGGL(k_fill_i32, dim3((n_agents + 63) / 64), dim3(64), 0, st, out_safe, n_agents, 1, agent0);
  if (n_rec > 0) {
    const size_t lds = sizeof(double) * (LP_WORK_DOUBLES + LP_MAX_ROWS * 5) + sizeof(int) * LP_MAX_ROWS;
    hipLaunchKernelGGL(k_safe_after_opt, dim3(n_rec, n_agents), dim3(64), lds, st, cpts, npoly, rec, n_rec,
                       ego_ids, t_now, out_safe, agent0);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_corridor(const MapView &m, const SogmPlannerParams &pp, const CorridorWorkspace &ws,
                    int n_agents, const double *start_pva, const double *t_start,
                    const double *route, const int32_t *route_len, int route_cap,
                    double *out_polys, int32_t *out_nfaces, int32_t *out_npoly, double *out_goal,
                    hipStream_t st, int agent0, hipEvent_t ev_map_read) {
  hipLaunchKernelGGL(k_corridor_points, dim3(SOGM_MAX_PIECES, n_agents), dim3(256), 0, st, m, pp, ws, start_pva,
                     t_start, route, route_len, route_cap, agent0);
  if (hipGetLastError() != hipSuccess) return -1;
  // nothing after this point reads the SOGM
  if (ev_map_read && hipEventRecord(ev_map_read, st) != hipSuccess) return -1;
  const size_t ldsA = corridor_segment_lds(pp.pc_capacity);
  hipLaunchKernelGGL(k_corridor_segment, dim3(SOGM_MAX_PIECES, n_agents), dim3(64), ldsA, st, m,
                     pp, ws, start_pva, t_start, route, route_len, route_cap, agent0);
  if (hipGetLastError() != hipSuccess) return -1;
  const size_t ldsB = sizeof(double) * (LP_WORK_DOUBLES + LP_MAX_ROWS * 5) + sizeof(int) * LP_MAX_ROWS;
  hipLaunchKernelGGL(k_corridor_finalize, dim3(n_agents), dim3(64), ldsB, st, pp, ws, start_pva,
                     route, route_len, route_cap, out_polys, out_nfaces, out_npoly, out_goal, agent0);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace sogm

// debug aid (not part of include/sogm_abi.h): resident workgroups per CU of the segment kernel
extern "C" int sogm_debug_corridor_occupancy(int pc_capacity) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)sogm::k_corridor_segment, 64,
                                                   sogm::corridor_segment_lds(pc_capacity)) != hipSuccess)
    return -1;
  return n;
}

#ifdef SOGM_PROFILE_MVIE
extern "C" int sogm_debug_mvie_prof(unsigned long long *out2) {
  if (hipMemcpyFromSymbol(out2, HIP_SYMBOL(sogm::g_mvie_prof), 16) != hipSuccess) return -1;
  unsigned long long z[2] = {0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(sogm::g_mvie_prof), z, 16);
  return 0;
}
#endif
