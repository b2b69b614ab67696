// sogm_corridor.hip — batched safe-corridor generation (FIRI polytopes) on the SOGM, gfx950.
//
// Reference: the corridor stage of FakeBaselinePlanner::replan / BaselinePlanner::replan
// (plan_manager/src/baseline_fake.cpp:300-414, baseline.cpp:296-403) =
//   getObstaclePoints (plan_env/src/map.cpp:480-518 / risk_base.cpp:295-337)
//   firi::firi + maxVolInsEllipsoid + costMVIE (plan_manager/include/sfc_gen/firi.hpp:44-365)
//   lbfgs::lbfgs_optimize + Lewis-Overton line search (plan_manager/include/sfc_gen/lbfgs.hpp)
//   sdlp::linprog<3|4> (traj_utils/include/traj_utils/sdlp.hpp) — the projective Seidel LP, whole-wave
//   ShrinkCorridor / checkCorridorValidity / checkCorridorIntersect / checkGoalReachability.
//
// Mapping to CDNA4: one 64-lane wave per (agent, path segment) — 7 segments x 128 agents = 896 independent
// problems fill the 256 CUs (4 waves per CU, 39.8 KB LDS each).  Inside a wave the point-cloud work is lane-parallel
// (obstacle-point extraction with an order-preserving wave scan, ellipsoid-frame transform, tangent planes, the
// greedy plane selection with a wave arg-min that breaks ties by index exactly like the reference's sequential scan);
// the LPs are solved by the whole wave (sdlp's projective algorithm, see linprog_wave), the MVIE cost is evaluated one
// face per lane and summed in the reference's order, the 9-variable L-BFGS and the 3x3 Jacobi run replicated /
// on lane 0 with their working set in LDS.  The per-agent bookkeeping (first invalid corridor, adjacent
// intersections, goal projection) is done by the wave that finishes an agent's last segment (dataflow replan) or by
// a second small kernel (grouped path).  All fp64, operation order identical to the CPU oracle — which follows the
// reference's (-ffp-contract=off, include/sogm_detmath.h for log) — so polytopes match bit for bit.
// No dense contraction -> no MFMA.
#include <hip/hip_runtime.h>

#include <cfloat>

#include "../../include/sogm_detmath.h"
#include "sogm_planner.hpp"

namespace sogm {
namespace {

#define LP_MAX_ROWS 153  // planes of one LP incl. sdlp's plane 0: 2 * max_faces(64) + 1, or 144 deconfliction rows + 8 box rows + 1; keeps the segment kernel at 4 workgroups per CU (LDS <= 40 KB)
#define LP_WORK_DOUBLES (14 * LP_MAX_ROWS)  // planes of the four recursion levels: (5 + 4 + 3 + 2) per row
#define FIRI_MAX_H 128  // planes selected before truncation to max_faces

__device__ inline double dot3(const double *a, const double *b) {
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
__device__ inline double dabs(double x) { return x < 0 ? -x : x; }

__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------
// sdlp::linprog<d> (traj_utils/include/traj_utils/sdlp.hpp:709-787) executed by a whole wave.
//
// Hohmeyer's projective Seidel LP: planes carry d+1 homogeneous coefficients, plane 0 is "x_d >= 0", the
// objective is n.x / d.x; a violated plane recurses into the problem on that plane with the coordinate of its
// largest coefficient eliminated (linfracprog<d>, :526-662), the 1-D problem is a wedge on the projective line
// (wedge / lp_base_case, :260-446).  All 64 lanes call with identical arguments; the arithmetic and every
// decision are those of the sequential code (same operation order as oracle/lp_oracle.cpp, bit for bit):
//   * sdlp's doubly linked list (next/prev, shared by all recursion levels) is the array ord[position] ->
//     plane; move_to_front (:132-150) of the plane at position q rotates ord[1..q] by one, and "continue with
//     the successor of the returned plane" is position q + 1 in either of its branches;
//   * opt does not change between two violated planes, so "the next violated plane in list order" is found 64
//     positions at a time with a ballot (first set bit = the sequential scan's hit); the same holds for the
//     wedge, whose state (cw, ccw) only changes at an "offensive" plane;
//   * the planes in front of the violated one are projected one per lane (:604-618);
//   * objective vectors and optima live in registers (compile-time indices, select chains for imax).
// Deviation from the reference (documented in DESIGN.md): the insertion order is a fixed LCG Fisher-Yates
// permutation of the row count instead of sdlp's process-global mt19937_64 (call-history dependent).
// ---------------------------------------------------------------------------------------------
#define SDLP_EPS 1.0e-12
enum { SDLP_MINIMUM = 0, SDLP_INFEASIBLE = 1, SDLP_UNBOUNDED = 2, SDLP_AMBIGUOUS = 3 };

// lp_no_con<d> (:97-129) incl. unit<d> (:76-94)
template <int D>
__device__ __forceinline__ int lp_no_con(const double (&nv)[D + 1], const double (&dv)[D + 1],
                                         double (&opt)[D + 1]) {
  double n_dot_d = 0.0, d_dot_d = 0.0;
#pragma unroll
  for (int i = 0; i <= D; ++i) {
    n_dot_d += nv[i] * dv[i];
    d_dot_d += dv[i] * dv[i];
  }
  if (d_dot_d < SDLP_EPS * SDLP_EPS) {
    n_dot_d = 0.0;
    d_dot_d = 1.0;
  }
#pragma unroll
  for (int i = 0; i <= D; ++i) opt[i] = -nv[i] + dv[i] * n_dot_d / d_dot_d;
  double mag = 0.0;
#pragma unroll
  for (int i = 0; i <= D; ++i) mag += opt[i] * opt[i];
  if (mag < (D + 1) * SDLP_EPS * SDLP_EPS) {
    opt[D] = 1.0;
    return SDLP_AMBIGUOUS;
  }
  mag = 1.0 / sogm_det::sqrt_rn(mag);
#pragma unroll
  for (int i = 0; i <= D; ++i) opt[i] *= mag;
  return SDLP_MINIMUM;
}

// move_to_front (:132-150) on the position array: the plane at position q goes to position 1
__device__ __forceinline__ void lp_move_to_front(int *ord, int q) {
  if (q > 1) {  // q == 0: plane 0; q == 1: already next[0]
    const int lane = threadIdx.x & 63;
    const int iq   = ord[q];
    const int r0 = 1 + lane, r1 = 65 + lane, r2 = 129 + lane;  // LP_MAX_ROWS <= 193 positions
    const int v0 = r0 < q ? ord[r0] : 0;
    const int v1 = r1 < q ? ord[r1] : 0;
    const int v2 = r2 < q ? ord[r2] : 0;
    wave_lds_sync();
    if (r0 < q) ord[r0 + 1] = v0;
    if (r1 < q) ord[r1 + 1] = v1;
    if (r2 < q) ord[r2 + 1] = v2;
    if (lane == 0) ord[1] = iq;
    wave_lds_sync();
  }
}

__device__ __forceinline__ double dot2(const double a[2], const double b[2]) { return a[0] * b[0] + a[1] * b[1]; }
__device__ __forceinline__ double cross2(const double a[2], const double b[2]) { return a[0] * b[1] - a[1] * b[0]; }
// unit2 (:61-73); b may alias a
__device__ __forceinline__ bool unit2(const double a[2], double b[2]) {
  const double a0 = a[0], a1 = a[1];
  const double mag = sogm_det::sqrt_rn(a0 * a0 + a1 * a1);
  if (mag < 2.0 * SDLP_EPS) return true;
  b[0] = a0 / mag;
  b[1] = a1 / mag;
  return false;
}

// lp_min_lin_rat (:152-258)
__device__ __forceinline__ void lp_min_lin_rat(bool degen, const double cw[2], const double ccw[2],
                                               const double nv[2], const double dv[2], double opt[2]) {
  const double d_cw = dot2(cw, dv), d_ccw = dot2(ccw, dv);
  const double n_cw = dot2(cw, nv), n_ccw = dot2(ccw, nv);
  bool take_cw;
  if (degen) {
    take_cw = n_cw / d_cw < n_ccw / d_ccw;
  } else if (dabs(d_cw) > 2.0 * SDLP_EPS && dabs(d_ccw) > 2.0 * SDLP_EPS) {
    if (d_cw * d_ccw > 0.0) {
      take_cw = n_cw / d_cw < n_ccw / d_ccw;
    } else {
      if (d_cw > 0.0) {
        opt[0] = -dv[1];
        opt[1] = dv[0];
      } else {
        opt[0] = dv[1];
        opt[1] = -dv[0];
      }
      return;
    }
  } else if (dabs(d_cw) > 2.0 * SDLP_EPS) {
    take_cw = n_ccw * d_cw > 0.0;
  } else if (dabs(d_ccw) > 2.0 * SDLP_EPS) {
    take_cw = !(n_cw * d_ccw > 2.0 * SDLP_EPS);
  } else {
    take_cw = cross2(dv, nv) > 0.0;
  }
  opt[0] = take_cw ? cw[0] : ccw[0];
  opt[1] = take_cw ? cw[1] : ccw[1];
}

// first position in [p, count) whose lane predicate holds (-1 if none); pred(r) is evaluated one position per lane
template <class F>
__device__ __forceinline__ int lp_first(int p, int count, F pred) {
  const int lane = threadIdx.x & 63;
  for (int base = p; base < count; base += 64) {
    const int                r  = base + lane;
    const bool               ok = r < count ? pred(r) : false;
    const unsigned long long mk = __ballot(ok);
    if (mk) return base + __ffsll((long long)mk) - 1;
  }
  return -1;
}

template <int D>
struct Lfp {
  // halves: LDS, stride D+1, indexed by plane; the list is ord[0..count).  work: planes of the lower levels.
  __device__ __forceinline__ static int solve(const double *halves, int count, const double (&nv_in)[D + 1],
                              const double (&dv_in)[D + 1], double (&opt)[D + 1], double *work, int *ord) {
    const int lane = threadIdx.x & 63;
    // the objective by value: a select between two entries of the CALLER's array would be folded into an
    // indexed load before inlining and demote that array to scratch memory
    double nv[D + 1], dv[D + 1];
#pragma unroll
    for (int j = 0; j <= D; ++j) {
      nv[j] = nv_in[j];
      dv[j] = dv_in[j];
    }
    double    val  = 0.0;
#pragma unroll
    for (int j = 0; j <= D; ++j) val += dv[j] * dv[j];
    const bool d_vec_zero = val < (D + 1) * SDLP_EPS * SDLP_EPS;
    int        status     = lp_no_con<D>(nv, dv, opt);
    if (count <= 0) return status;
    double *new_halves = work;  // [LP_MAX_ROWS][D]
    int     p          = 0;
    while (p < count) {
      const int q = lp_first(p, count, [&](int r) {
        const double *pl = halves + ord[r] * (D + 1);
        double        v  = 0.0;
#pragma unroll
        for (int j = 0; j <= D; ++j) v += opt[j] * pl[j];
        return v < -(D + 1) * SDLP_EPS;
      });
      if (q < 0) break;
      const int     i  = ord[q];
      const double *pi = halves + i * (D + 1);
      double        pv[D + 1];
#pragma unroll
      for (int j = 0; j <= D; ++j) pv[j] = pi[j];
      int    imax = 0;  // findimax (:449-464); the imax-th entries of the plane and of both objective vectors
      double rmax = dabs(pv[0]), pmax = pv[0], nmax = nv[0], dmax = dv[0];  // ride along (no indexed access)
#pragma unroll
      for (int j = 1; j <= D; ++j) {
        const double ab = dabs(pv[j]);
        if (ab > rmax) {
          imax = j;
          rmax = ab;
          pmax = pv[j];
          nmax = nv[j];
          dmax = dv[j];
        }
      }
      if (i != 0) {  // project the planes in front of i (:604-618), one per lane
        const double fac = 1.0 / pmax;
        for (int r = lane; r < q; r += 64) {
          const int     j    = ord[r];
          const double *old  = halves + j * (D + 1);
          const double  crit = old[imax] * fac;
          double       *np   = new_halves + j * D;
#pragma unroll
          for (int l = 0; l < D; ++l) {
            const int k = l < imax ? l : l + 1;
            np[l]       = old[k] - (l < imax ? pv[l] : pv[l + 1]) * crit;
          }
        }
      }
      wave_lds_sync();
      double nn[D], nd[D];
      if (d_vec_zero) {  // vector_down (:485-507)
        double ve = 0.0, ee = 0.0;
#pragma unroll
        for (int j = 0; j <= D; ++j) {
          ve += nv[j] * pv[j];
          ee += pv[j] * pv[j];
        }
        const double fac = ve / ee;
#pragma unroll
        for (int l = 0; l < D; ++l) {
          nn[l] = (l < imax ? nv[l] : nv[l + 1]) - (l < imax ? pv[l] : pv[l + 1]) * fac;
          nd[l] = 0.0;
        }
      } else {  // plane_down (:509-524) for numerator and denominator
        const double critn = nmax / pmax;
        const double critd = dmax / pmax;
#pragma unroll
        for (int l = 0; l < D; ++l) {
          const double e = l < imax ? pv[l] : pv[l + 1];
          nn[l]          = (l < imax ? nv[l] : nv[l + 1]) - e * critn;
          nd[l]          = (l < imax ? dv[l] : dv[l + 1]) - e * critd;
        }
      }
      double nopt[D];
      status = Lfp<D - 1>::solve(new_halves, q, nn, nd, nopt, work + LP_MAX_ROWS * D, ord);
      if (status == SDLP_INFEASIBLE) return status;
      // vector_up (:466-483) then the inline unit (:641-651)
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j <= D; ++j) {
        const double lo = nopt[j < D ? j : 0];       // low_vector[j]     (used when j < imax)
        const double hi = nopt[j > 0 ? j - 1 : 0];   // low_vector[j - 1] (used when j > imax)
        const double v  = j < imax ? lo : hi;
        const double na = acc - pv[j] * v;
        acc             = j != imax ? na : acc;
        opt[j]          = v;
      }
      acc /= pmax;
#pragma unroll
      for (int j = 0; j <= D; ++j) opt[j] = j == imax ? acc : opt[j];
      double mag = 0.0;
#pragma unroll
      for (int j = 0; j <= D; ++j) mag += opt[j] * opt[j];
      mag = 1.0 / sogm_det::sqrt_rn(mag);
#pragma unroll
      for (int j = 0; j <= D; ++j) opt[j] *= mag;
      lp_move_to_front(ord, q);
      p = q + 1;
    }
    return status;
  }
};

// linfracprog<1> (:664-684) = lp_base_case (:378-446) over wedge (:260-375); halves stride 2
template <>
struct Lfp<1> {
  __device__ __forceinline__ static int solve(const double *halves, int count, const double (&nv)[2], const double (&dv)[2],
                              double (&opt)[2], double *, int *ord) {
    if (count <= 0) return lp_no_con<1>(nv, dv, opt);
    const double e2 = 2.0 * SDLP_EPS;
    double       cw[2], ccw[2];
    bool         degen = false;
    {  // the first plane of the list that is not (numerically) zero spans the initial half circle
      const int q0 = lp_first(0, count, [&](int r) {
        const double *h = halves + 2 * ord[r];
        return !(sogm_det::sqrt_rn(h[0] * h[0] + h[1] * h[1]) < e2);
      });
      if (q0 < 0) return lp_no_con<1>(nv, dv, opt);  // wedge: UNBOUNDED
      const double *h = halves + 2 * ord[q0];
      unit2(h, ccw);
      cw[0]  = ccw[1];
      cw[1]  = -ccw[0];
      ccw[0] = -cw[0];
      ccw[1] = -cw[1];
    }
    int p = 0;
    while (p < count) {
      const int q = lp_first(p, count, [&](int r) {
        const double *h    = halves + 2 * ord[r];
        const double  d_cw = dot2(cw, h), d_ccw = dot2(ccw, h);
        if (d_ccw >= e2) return d_cw <= -e2;
        if (d_cw >= e2) return d_ccw <= -e2;
        if (d_ccw <= -e2 && d_cw <= -e2) return true;
        return d_cw <= -e2 || d_ccw <= -e2 || cross2(cw, h) < 0.0;
      });
      if (q < 0) break;
      const double h[2]  = {halves[2 * ord[q]], halves[2 * ord[q] + 1]};
      const double d_cw = dot2(cw, h), d_ccw = dot2(ccw, h);
      if (d_ccw >= e2) {
        cw[0] = h[1];
        cw[1] = -h[0];
        unit2(cw, cw);
      } else if (d_cw >= e2) {
        ccw[0] = -h[1];
        ccw[1] = h[0];
        unit2(ccw, ccw);
      } else if (d_ccw <= -e2 && d_cw <= -e2) {
        return SDLP_INFEASIBLE;
      } else {
        if (d_cw <= -e2)
          unit2(ccw, cw);
        else if (d_ccw <= -e2)
          unit2(cw, ccw);
        degen = true;
      }
      lp_move_to_front(ord, q);
      p = q + 1;
      if (degen) break;
    }
    if (degen) {
      while (p < count) {
        const int q = lp_first(p, count, [&](int r) {
          const double *h = halves + 2 * ord[r];
          return dot2(cw, h) < -e2 || dot2(ccw, h) < -e2;
        });
        if (q < 0) break;
        const double h[2] = {halves[2 * ord[q]], halves[2 * ord[q] + 1]};
        const double d_cw = dot2(cw, h), d_ccw = dot2(ccw, h);
        if (d_cw < -e2) {
          if (d_ccw < -e2) return SDLP_INFEASIBLE;
          cw[0] = ccw[0];
          cw[1] = ccw[1];
        } else {
          ccw[0] = cw[0];
          ccw[1] = cw[1];
        }
        p = q + 1;
      }
    }
    // lp_base_case (:403-445)
    if (dabs(cross2(nv, dv)) < 2.0 * SDLP_EPS * SDLP_EPS) {
      if (dot2(nv, nv) < 2.0 * SDLP_EPS * SDLP_EPS || dot2(dv, dv) > 2.0 * SDLP_EPS * SDLP_EPS) {
        opt[0] = cw[0];
        opt[1] = cw[1];
        return SDLP_AMBIGUOUS;
      }
      if (!degen && cross2(cw, nv) <= 0.0 && cross2(nv, ccw) <= 0.0) {
        opt[0] = -nv[0];
        opt[1] = -nv[1];
      } else if (dot2(nv, cw) > dot2(nv, ccw)) {
        opt[0] = ccw[0];
        opt[1] = ccw[1];
      } else {
        opt[0] = cw[0];
        opt[1] = cw[1];
      }
      return SDLP_MINIMUM;
    }
    lp_min_lin_rat(degen, cw, ccw, nv, dv, opt);
    return SDLP_MINIMUM;
  }
};

// linprog<D> (:709-787): min c^T x s.t. A[i][0..D) x <= rhs[i]  (A row-major, stride D, in LDS).
// work: LP_WORK_DOUBLES doubles (LDS), ord: LP_MAX_ROWS ints (LDS); rows < LP_MAX_ROWS.
// Returns +inf infeasible, -inf unbounded / optimum at infinity, else the minimum.  Whole wave.
template <int D>
__device__ __forceinline__ double linprog_wave(const double *c, int rows, const double *A, const double *rhsv,
                                               double *x, double *work, int *ord) {
  const int lane = threadIdx.x & 63;
  for (int j = 0; j < D; ++j) x[j] = 0.0;
  if (rows <= 0) {
    double mx = 0;
    for (int j = 0; j < D; ++j) mx = dabs(c[j]) > mx ? dabs(c[j]) : mx;
    return mx > 0.0 ? -INFINITY : 0.0;
  }
  const int m      = rows + 1;
  double   *halves = work;  // [LP_MAX_ROWS][D + 1]
  if (lane == 0) {
    ord[0] = 0;
    for (int i = 0; i < rows; ++i) ord[1 + i] = i + 1;
    unsigned long long s = 0x9E3779B97F4A7C15ULL;  // the fixed insertion order (oracle: fixed_permutation)
    for (int i = rows - 1; i > 0; --i) {
      s           = s * 6364136223846793005ULL + 1442695040888963407ULL;
      const int j = (int)((s >> 33) % (unsigned long long)(i + 1));
      const int t = ord[1 + i];
      ord[1 + i]  = ord[1 + j];
      ord[1 + j]  = t;
    }
    for (int j = 0; j < D; ++j) halves[j] = 0.0;
    halves[D] = 1.0;
  }
  for (int i = 1 + lane; i < m; i += 64) {  // halves.col(i) = (-A_i, b_i) normalised (:737-740)
    const double *src = A + (i - 1) * D;
    double        h[D + 1];
#pragma unroll
    for (int j = 0; j < D; ++j) h[j] = -src[j];
    h[D]      = rhsv[i - 1];
    double nn = 0.0;
#pragma unroll
    for (int j = 0; j <= D; ++j) nn += h[j] * h[j];
    nn          = sogm_det::sqrt_rn(nn);
    double *dst = halves + i * (D + 1);
#pragma unroll
    for (int j = 0; j <= D; ++j) dst[j] = nn > 0.0 ? h[j] / nn : h[j];
  }
  wave_lds_sync();
  double nv[D + 1], dv[D + 1], opt[D + 1];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    nv[j] = c[j];
    dv[j] = 0.0;
  }
  nv[D] = 0.0;
  dv[D] = 1.0;
  const int status = Lfp<D>::solve(halves, m, nv, dv, opt, work + LP_MAX_ROWS * (D + 1), ord);
  double    minimum = INFINITY;
  if (status != SDLP_INFEASIBLE) {
    if (opt[D] != 0.0 && status != SDLP_UNBOUNDED) {
      minimum = 0.0;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        x[j] = opt[j] / opt[D];
        minimum += c[j] * x[j];
      }
    }
    if (opt[D] == 0.0 || status == SDLP_UNBOUNDED) {
#pragma unroll
      for (int j = 0; j < D; ++j) x[j] = opt[j];
      minimum = -INFINITY;
    }
  }
  return minimum;
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

#ifdef SOGM_PROFILE_MVIE
__device__ unsigned long long g_mvie_prof[2];  // profiling build only: ticks (100 MHz) and calls of costMVIE
__device__ unsigned long long g_lbfgs_prof[5];  // line-search ticks, direction-update ticks, iterations, history entries, ticks of the two loops alone
#endif
__device__ inline double lane_f64(double v, int l) {  // value of lane l (l wave-uniform)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo     = __builtin_amdgcn_readlane(lo, l);
  hi     = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
// N independent IEEE fp64 divisions x[i] / y[i], stage by stage: the instruction sequence the compiler expands "x / y" to
// (v_div_scale of the denominator and of the numerator, v_rcp, two Newton steps on the reciprocal, the quotient, its
// residual, v_div_fmas, v_div_fixup — LLVM's LowerFDIV64), so every quotient has the bits of the plain division; written
// out because the expansion passes its scale flag through VCC, which makes the compiler emit N divisions as N chains of
// thirteen dependent instructions one after the other — on a wave that is alone on its SIMD and issues in order, N times
// ~110 cycles.  Stage-wise the chains overlap (the flags live in ordinary scalar registers until their v_div_fmas).
template <int N>
__device__ __forceinline__ void div_n(const double (&x)[N], const double (&y)[N], double (&q)[N]) {
  double s0[N], s1[N], r[N], e[N], m[N];
  bool   fl[N], unused;
#pragma unroll
  for (int i = 0; i < N; ++i) s0[i] = __builtin_amdgcn_div_scale(x[i], y[i], false, &unused);  // the denominator, scaled
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = __builtin_amdgcn_rcp(s0[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) s1[i] = __builtin_amdgcn_div_scale(x[i], y[i], true, &fl[i]);  // the numerator, scaled
#pragma unroll
  for (int i = 0; i < N; ++i) e[i] = __builtin_fma(-s0[i], r[i], 1.0);
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = __builtin_fma(r[i], e[i], r[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) e[i] = __builtin_fma(-s0[i], r[i], 1.0);
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = __builtin_fma(r[i], e[i], r[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) m[i] = s1[i] * r[i];
#pragma unroll
  for (int i = 0; i < N; ++i) e[i] = __builtin_fma(-s0[i], m[i], s1[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) q[i] = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(e[i], r[i], m[i], fl[i]), y[i], x[i]);
}

// one face's terms of cost and gradient (firi.hpp:105-122); false when smoothedL1 rejects the face.  The face's five
// divisions — A L / ||A L|| and the two of smoothedL1's cubic branch (firi.hpp's x / mu and 3 (mu - x / 2) / mu, evaluated
// whether or not the branch is the lane's: some lane of the wave takes it) — go through div_n together.
__device__ inline bool mvie_face(const double a[3], const double L[3][3], const double *p,
                                 double smoothEps, double t[10]) {
  double AL[3];
  for (int j = 0; j < 3; ++j) AL[j] = (a[0] * L[0][j] + a[1] * L[1][j]) + a[2] * L[2][j];
  const double normAL = sogm_det::sqrt_rn((AL[0] * AL[0] + AL[1] * AL[1]) + AL[2] * AL[2]);
  const double Ap     = (a[0] * p[0] + a[1] * p[1]) + a[2] * p[2];
  const double viola  = (normAL + Ap) - 1.0;
  const double mu = smoothEps, mumxd2 = mu - 0.5 * viola;
  const double nx[5] = {AL[0], AL[1], AL[2], viola, 3.0 * mumxd2}, ny[5] = {normAL, normAL, normAL, mu, mu};
  double       qd[5];
  div_n<5>(nx, ny, qd);
  const double adj[3] = {qd[0], qd[1], qd[2]};
  double       c, dc;
  // smoothedL1(mu, viola, c, dc) with its two quotients taken from above
  if (viola < 0.0) return false;
  if (viola > mu) {
    c  = viola - 0.5 * mu;
    dc = 1.0;
  } else {
    const double xdmu = qd[3], sqrxdmu = xdmu * xdmu;
    c  = mumxd2 * sqrxdmu * xdmu;
    dc = sqrxdmu * ((-0.5) * xdmu + qd[4]);
  }
  t[0]                = c;
  const double vec[3] = {dc * a[0], dc * a[1], dc * a[2]};
  for (int j = 0; j < 3; ++j) t[1 + j] = vec[j];
  for (int j = 0; j < 3; ++j) t[4 + j] = adj[j] * vec[j];
  t[7] = adj[0] * vec[1];
  t[8] = adj[1] * vec[2];
  t[9] = adj[0] * vec[2];
  return true;
}

// costMVIE (firi.hpp:74-140), evaluated by the whole wave: one face per lane (two when M > 64).  The ten sums
// over faces (cost, gdp, gdrtd, gdcde) are accumulated IN THE REFERENCE'S ORDER — one running sum per quantity,
// faces in index order, only the faces smoothedL1 accepts (:111-122): the accepted faces' terms are compacted in
// face order into LDS (`terms`, >= 10 * M doubles: the LP work area, idle during the L-BFGS), lanes 0..9 each
// run one of the ten chains (usually a handful of dependent adds: few faces touch the ellipsoid), the totals are
// broadcast.  Every lane returns the same cost and gradient, bit-identical to the sequential loop.
__device__ __forceinline__ double costMVIE(const MvieData &D, const double *x, double *g, double *terms) {
  const int     lane = threadIdx.x & 63;
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
  // log and reciprocal of the three diagonal entries (the barrier term, used at the end): one entry per lane (lanes
  // 0..2; the others repeat entry 0) instead of three chains one after the other on the single wave of a SIMD, and
  // issued here, where the face terms' own dependent chains leave issue slots free; same functions, same inputs
  const double Ld = lane == 1 ? L[1][1] : (lane == 2 ? L[2][2] : L[0][0]);
  const double lg = sogm_det::log(Ld), rc = 1.0 / Ld;
  double t0[10], t1[10];
  bool   a0 = false, a1 = false;
  if (D.on0) a0 = mvie_face(D.a0, L, p, D.smoothEps, t0);
  if (D.on1) a1 = mvie_face(D.a1, L, p, D.smoothEps, t1);
  const unsigned long long m0 = __ballot(a0), m1 = D.M > 64 ? __ballot(a1) : 0ull;
  const unsigned long long below = (1ull << lane) - 1ull;
  const int                n0 = __popcll(m0), n = n0 + __popcll(m1);
  if (a0) {
    double *dst = terms + __popcll(m0 & below) * 10;
#pragma unroll
    for (int k = 0; k < 10; ++k) dst[k] = t0[k];
  }
  if (a1) {
    double *dst = terms + (n0 + __popcll(m1 & below)) * 10;
#pragma unroll
    for (int k = 0; k < 10; ++k) dst[k] = t1[k];
  }
  wave_lds_sync();
  double sum = 0.0;
  if (lane < 10) {
    // the running sum in face order, its LDS reads four at a time (read one, wait, add one was a chain of LDS latencies
    // as long as the number of faces the ellipsoid touches); a row beyond n re-reads row n - 1 and is not added
    const double *src = terms + lane;
    for (int r = 0; r < n; r += 4) {
      double t4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t4[u] = src[(r + u < n ? r + u : n - 1) * 10];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double t = sum + t4[u];
        sum            = r + u < n ? t : sum;
      }
    }
  }
  wave_lds_sync();  // the next evaluation overwrites `terms`
  double acc[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = lane_f64(sum, k);
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
  cost -= lane_f64(lg, 0) + lane_f64(lg, 1) + lane_f64(lg, 2);
  gdrtd[0] -= lane_f64(rc, 0);
  gdrtd[1] -= lane_f64(rc, 1);
  gdrtd[2] -= lane_f64(rc, 2);
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
  // (max |v_i| from +0: v_max_f64 with the |x| operand modifier gives the value of "dabs(v[i]) > mx ? dabs(v[i]) : mx" for
  //  every finite input — a compare, a sign flip and four selects per element otherwise, twice per L-BFGS iteration on
  //  a wave that issues in order)
  double mx = 0;
  for (int i = 0; i < 9; ++i) mx = __builtin_fmax(mx, __builtin_fabs(v[i]));
  return mx;
}

#ifdef SOGM_PROFILE_MVIE
#define MVIE_PROF_ARG , long long *prof
#define MVIE_PROF_PASS , prof
#else
#define MVIE_PROF_ARG
#define MVIE_PROF_PASS
#endif
__device__ __forceinline__ int lineSearchLO(const MvieData &D, double *x, double &f, double *g, double &stp,
                            const double *s, const double *xp, const double *gp, double stpmin,
                            double stpmax, double *terms MVIE_PROF_ARG) {
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
    f = costMVIE(D, x, g, terms);
#ifdef SOGM_PROFILE_MVIE
    prof[0] += wall_clock64() - tc0;
    prof[1] += 1;
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
//   lm[0..162) = s history, lm[162..324) = y history
#define LBFGS_FENCE()                                        \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)
__device__ __forceinline__ int lbfgsMVIE(const MvieData &D, double *x, double *lm, double *terms, int *n_iter,
                                         int *n_eval) {
  const int    n = 9, m = 18, past = 3;
  const double g_epsilon = 0.0, delta = 1.0e-7, min_step = 1.0e-32, max_step = 1.0e+20,
               cautious = 1.0e-6;
  double       xp[9], g[9], gp[9], d[9];
  double       pf0 = 0, pf1 = 0, pf2 = 0;  // pf[k % 3]
  const int    lane   = threadIdx.x & 63;
  const bool   writer = lane == 0;
  // s / y history in LDS (lane 0 writes an entry per iteration); the per-entry scalars y.s and alpha live in lane j
  // of two registers: written with a lane compare, read back with readlane, so the two-loop recursion contains no
  // LDS store and its history reads can be issued ahead of the dependent dot product / division / update.
  // (Also tried: the whole history in lane registers — 9.1 vs 8.7 us per iteration, 36 more AGPRs; prefetching the
  // next entry's vector one entry ahead — 9.3 us; a Markstein-corrected multiply by the stored reciprocal of y.s in
  // place of the division — 8.6 us: the compiler already starts the divisor's part of the division early.)
  double      *lm_s = lm, *lm_y = lm + m * n;
  double       ys_keep = 0.0, al_keep = 0.0;
#ifdef SOGM_PROFILE_MVIE
  long long prof[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
  if (writer)
    for (int i = 0; i < 2 * m * n; ++i) lm[i] = 0;
  LBFGS_FENCE();
  double fx = costMVIE(D, x, g, terms);
  pf0       = fx;
  for (int i = 0; i < n; ++i) d[i] = -g[i];
  int          ret;
  // (g_epsilon is 0 here (firi.hpp:193): "||g||inf / max(1, ||x||inf) < 0" cannot hold — the quotient of a non-negative
  //  number and one >= 1 is non-negative or NaN — so with the constant the two norms and the division fold away;
  //  written so that a positive g_epsilon would bring the test back)
  auto grad_small = [&]() __attribute__((always_inline)) -> bool {
    if (!(g_epsilon > 0.0)) return false;
    const double xn = ninf9(x);
    return ninf9(g) / (1.0 > xn ? 1.0 : xn) < g_epsilon;
  };
  if (grad_small()) {
    ret = 0;
  } else {
    double step = 1.0 / sogm_det::sqrt_rn(dotn9(d, d));
    int    k = 1, end = 0, bound = 0;
    while (true) {
      for (int i = 0; i < n; ++i) {
        xp[i] = x[i];
        gp[i] = g[i];
      }
#ifdef SOGM_PROFILE_MVIE
      const long long tl0 = wall_clock64();
#endif
      const int ls = lineSearchLO(D, x, fx, g, step, d, xp, gp, min_step, max_step, terms MVIE_PROF_PASS);
#ifdef SOGM_PROFILE_MVIE
      prof[2] += wall_clock64() - tl0;
      prof[4] += 1;
      const long long tr0 = wall_clock64();
#endif
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
      if (grad_small()) {
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
      double  sv[9], yv[9];
      for (int i = 0; i < n; ++i) {
        sv[i] = x[i] - xp[i];
        yv[i] = g[i] - gp[i];
      }
      const double ys = dotn9(yv, sv);
      const double yy = dotn9(yv, yv);
      if (writer) {
        double *se = lm_s + end * n, *ye = lm_y + end * n;
        for (int i = 0; i < n; ++i) {
          se[i] = sv[i];
          ye[i] = yv[i];
        }
      }
      if (lane == end) ys_keep = ys;  // lm_ys[end]
      LBFGS_FENCE();
      for (int i = 0; i < n; ++i) d[i] = -g[i];
      const double cau = dotn9(sv, sv) * sogm_det::sqrt_rn(dotn9(gp, gp)) * cautious;
      // every lane holds the same values: make the bookkeeping provably wave-uniform (scalar branches, and
      // readlane needs a scalar lane index)
      if (__builtin_amdgcn_readfirstlane((int)(ys > cau))) {
#ifdef SOGM_PROFILE_MVIE
        const long long tw0 = wall_clock64();
#endif
        ++bound;
        bound = m < bound ? m : bound;
        end   = end + 1 == m ? 0 : end + 1;
        int j = end;
        // The two-loop recursion with d spread over the lanes: lane (row r, i) — i = lane & 15 < 9, every 16-lane DPP row
        // alike — holds d[i] and reads ITS element of s_j and y_j (two LDS reads per entry instead of eighteen); the nine
        // products of a dot product are one instruction, the update two, and the sum is taken in dotn9's order,
        // ((0 + p0) + p1) + ... + p8, by nine dependent v_fmac_f64_dpp with row_newbcast:k (acc += lane k's product x 1.0:
        // the product by one is exact, so each is the plain addition, rounded once) — every lane ends with the same sum.
        // 85 -> ~40 instructions per entry for a wave that is alone on its SIMD and issues in order; same operations on
        // the same values as the replicated form (polytopes stay bit-identical to the oracle's).
        const int ql  = (lane & 15) < n ? (lane & 15) : 0;
        double    dl  = d[0];
#pragma unroll
        for (int q = 1; q < n; ++q) dl = ql == q ? d[q] : dl;
        double one = 1.0;
        asm volatile("" : "+v"(one));
#define LBFGS_BC(K) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
        auto ordered_sum = [&](double p) __attribute__((always_inline)) -> double {
          double acc = 0.0;
          asm volatile("s_nop 1\n\t"  // (a VGPR written by the VALU needs two wait states before a DPP read)
                       LBFGS_BC(0) LBFGS_BC(1) LBFGS_BC(2) LBFGS_BC(3) LBFGS_BC(4) LBFGS_BC(5) LBFGS_BC(6) LBFGS_BC(7)
                       LBFGS_BC(8)
                       : "+v"(acc)
                       : "v"(p), "v"(one));
          return acc;
        };
#undef LBFGS_BC
        // (the next entry's two elements are read while the current entry's chain — product, ordered sum, division, update —
        //  runs: the address does not depend on it, and read at the top of its own trip the LDS latency stood in front of
        //  every entry.  The read past the last entry of a loop is of a valid slot and unused.)
        int    jn = j == 0 ? m - 1 : j - 1;
        double ns = lm_s[jn * n + ql], ny = lm_y[jn * n + ql];
        double hsq = 0.0, hyq = 0.0;
        for (int i = 0; i < bound; ++i) {
          j   = jn;
          hsq = ns;
          hyq = ny;
          jn  = j == 0 ? m - 1 : j - 1;
          ns  = lm_s[jn * n + ql];
          ny  = lm_y[jn * n + ql];
          const double alpha = ordered_sum(hsq * dl) / lane_f64(ys_keep, j);
          if (lane == j) al_keep = alpha;  // lm_alpha[j]
          dl += (-alpha) * hyq;
        }
        dl *= ys / yy;
        jn = j;  // the second loop starts at the entry the first one ended with: its elements are still in hsq / hyq
        ns = hsq;
        ny = hyq;
        for (int i = 0; i < bound; ++i) {
          j   = jn;
          hsq = ns;
          hyq = ny;
          jn  = j + 1 == m ? 0 : j + 1;
          ns  = lm_s[jn * n + ql];
          ny  = lm_y[jn * n + ql];
          const double beta = ordered_sum(hyq * dl) / lane_f64(ys_keep, j);
          const double al   = lane_f64(al_keep, j);
          dl += (al - beta) * hsq;
        }
#pragma unroll
        for (int q = 0; q < n; ++q) d[q] = lane_f64(dl, q);
#ifdef SOGM_PROFILE_MVIE
        prof[6] += wall_clock64() - tw0;
#endif
      }
      step = 1.0;
#ifdef SOGM_PROFILE_MVIE
      prof[3] += wall_clock64() - tr0;
      prof[5] += bound;
#endif
    }
  }
#ifdef SOGM_PROFILE_MVIE
  if (writer) {
    atomicAdd(&g_mvie_prof[0], (unsigned long long)prof[0]);
    atomicAdd(&g_mvie_prof[1], (unsigned long long)prof[1]);
    atomicAdd(&g_lbfgs_prof[0], (unsigned long long)prof[2]);
    atomicAdd(&g_lbfgs_prof[1], (unsigned long long)prof[3]);
    atomicAdd(&g_lbfgs_prof[2], (unsigned long long)prof[4]);
    atomicAdd(&g_lbfgs_prof[3], (unsigned long long)prof[5]);
    atomicAdd(&g_lbfgs_prof[4], (unsigned long long)prof[6]);
  }
#endif
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
#ifdef SOGM_PROFILE_MVIE
  const long long tcl0 = clock64();
#endif
  const int       ret = lbfgsMVIE(D, x, sc.lm, sc.lp_work, &n_it, &n_ev);  // LP work area: idle by now
  if (dbg && lane == 0) {
    dbg[3] = n_it;
    dbg[4] = n_ev;
    dbg[9] = wall_clock64() - tl0;
#ifdef SOGM_PROFILE_MVIE
    dbg[11] = clock64() - tcl0;  // shader-clock cycles of the same interval: dbg[11] / dbg[9] x 100 MHz = shader clock
#endif
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

// checkCorridorValidity (baseline.cpp:191-204; poly rows h0 x + h1 y + h2 z + h3 <= 0) and
// checkGoalReachability (baseline.cpp:143-182), executed by the whole wave (linprog_wave)
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
// NW = waves of the calling workgroup (4: k_corridor_points; 1: the dataflow kernel, where the segment's own wave
// extracts its points).  Same cells, same order, same output for either.
template <int NW>
__device__ __forceinline__ void corridor_points_body(const MapView &m, const SogmPlannerParams &pp,
                                                     const CorridorWorkspace &ws, const double *start_pva,
                                                     const double *t_start, const double *route,
                                                     const int32_t *route_len, int route_cap, int agent, int seg) {
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
      for (int c0 = 0; c0 < cells; c0 += 256 * NW) {
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
            const int pi = g.phys(x, y, z);
            for (int j = js; j <= je; ++j) {
              const float thr = g.map_kind == SOGM_MAP_FAKE ? g.risk_threshold
                                                             : g.risk_threshold - g.decay_voxel * (float)j;
              if (cell_ld(grid0, (size_t)j * g.V + pi, g.half) > thr) {
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
        for (int w = 0; w < NW; ++w) {
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

__global__ __launch_bounds__(256) void k_corridor_points(MapView m, SogmPlannerParams pp, CorridorWorkspace ws,
                                                        const double *__restrict__ start_pva,
                                                        const double *__restrict__ t_start,
                                                        const double *__restrict__ route,
                                                        const int32_t *__restrict__ route_len, int route_cap,
                                                        int agent0) {
  corridor_points_body<4>(m, pp, ws, start_pva, t_start, route, route_len, route_cap, blockIdx.y + agent0,
                          blockIdx.x);
}

namespace {
}  // namespace

// =================================================================================================
// Kernel A: one workgroup per (segment, agent)
// =================================================================================================
// MB = capacity of the boundary block (6 in the replan path: getInitCorridor's box), DIRECT = the standalone
// firi::firi entry (sogm_firi_batched): bd / points / a / b / r come from the caller instead of the route and the
// map, and the polytope is returned as firi leaves it (no ShrinkCorridor, no validity LP).
struct FiriDirect {
  const double  *bd;        // [n][n_bd][4]
  int            n_bd;
  const double  *pc;        // packed xyz
  const int32_t *pc_range;  // [n][2]
  const double  *a, *b;     // [n][3]
  double        *r;         // [n][3] in / out
  int            iterations;
  double        *hpoly;     // [n][max_faces][4]
  int32_t       *nfaces;    // [n]
  int32_t       *status;    // [n] 1 ok, 0 a or b outside bd (firi returns false), -3 over capacity
  int            max_faces;
  int            first;     // problem index of slot 0 in this launch
  double         epsilon;
};

template <int MB>
__host__ __device__ constexpr int firi_small_doubles() { return 34 + 9 * MB; }

template <int MB, bool DIRECT>
__device__ __forceinline__ void corridor_segment_body(const MapView &m, const SogmPlannerParams &pp,
                                                      const CorridorWorkspace &ws, const double *start_pva,
                                                      const double *t_start, const double *route,
                                                      const int32_t *route_len, int route_cap, int agent, int seg,
                                                      char *smem, const FiriDirect &fd) {
  const int lane  = threadIdx.x;
  const int slot  = DIRECT ? agent : agent * SOGM_MAX_PIECES + seg;
  if constexpr (!DIRECT) {
    const int rl = route_len[agent];
    if (seg >= rl - 1 || seg >= SOGM_MAX_PIECES) {
      if (lane == 0) ws.seg_state[slot] = -2;  // no such segment
      return;
    }
  }
  constexpr int  SMALL   = MB == 6 ? 96 : firi_small_doubles<MB>();
  double        *s_lp    = (double *)smem;                    // LP_WORK_DOUBLES
  double        *s_rows  = s_lp + LP_WORK_DOUBLES;            // LP_MAX_ROWS * 5
  double        *s_lm    = s_rows + LP_MAX_ROWS * 5;          // 324 history + 16 hand-off + 36 alpha/ys
  double        *s_fH    = s_lm + 2 * 18 * 9 + 16 + 36;       // FIRI_MAX_H * 4
  double        *s_poly  = s_fH + FIRI_MAX_H * 4;             // FIRI_MAX_H * 4
  double        *s_small = s_poly + FIRI_MAX_H * 4;           // SMALL doubles of shared small state
  // "point not yet covered" flags of the greedy selection: one BIT per obstacle point, 64 per word — the 64 lanes of
  // a trip of the point loops share one word, which lane 0 rewrites from a ballot (no atomics, 2 KiB for 16 k points)
  unsigned long long *s_fw = (unsigned long long *)(s_small + SMALL);  // (pc_capacity + 63) / 64 words
  int           *s_perm  = (int *)(s_fw + (pp.pc_capacity + 63) / 64);   // LP_MAX_ROWS
  int           *s_int   = s_perm + LP_MAX_ROWS;              // 16 ints
  SolverScratch  sc{s_lp, s_perm, s_rows, s_lm};

  // shared small state layout
  double *s_fwd  = s_small;                 // 9  forward
  double *s_fa   = s_small + 9;             // 3  fwd_a
  double *s_fb   = s_small + 12;            // 3  fwd_b
  double *s_p    = s_small + 15;            // 3
  double *s_fh   = s_small + 18;            // 4  current plane
  double *s_bd   = s_small + 22;            // 4 MB  bd
  double *s_fB   = s_bd + 4 * MB;           // 3 MB  forwardB
  double *s_fD   = s_fB + 3 * MB;           // MB    forwardD
  double *s_dD   = s_fD + MB;               // MB    distDs
  double *s_box  = s_dD + MB;               // 6  llc, lhc
  double *s_w    = s_box + 6;               // 6  w0, w1 (a, b)

  const double   *rt    = DIRECT ? nullptr : route + (size_t)agent * route_cap * 6;
  const double   *sp    = DIRECT ? nullptr : start_pva + agent * 9;
  const int       cap   = pp.pc_capacity;
  const int       prob  = DIRECT ? fd.first + slot : 0;
  const double   *pc    = DIRECT ? fd.pc + (size_t)fd.pc_range[prob * 2] * 3 : ws.pc + (size_t)slot * cap * 3;
  double         *fpc   = ws.fpc + (size_t)slot * cap * 3;
  double         *tang  = ws.tang + (size_t)slot * cap * 4;
  double         *distR = ws.distr + (size_t)slot * cap;

  long long *dbg = ws.seg_dbg + (size_t)slot * 16;
  const long long tk0 = wall_clock64();
  const int M = DIRECT ? fd.n_bd : 6;
  if (lane == 0) {
    for (int k = 0; k < 12; ++k) dbg[k] = 0;  // (12-15: the flight's own stamps of this slot, k_flight_light)
    if constexpr (DIRECT) {
      for (int i = 0; i < 4 * M; ++i) s_bd[i] = fd.bd[(size_t)prob * M * 4 + i];
      for (int k = 0; k < 3; ++k) {
        s_w[k]     = fd.a[prob * 3 + k];
        s_w[3 + k] = fd.b[prob * 3 + k];
      }
    } else {
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
  }
  __syncthreads();

  // obstacle points: written by k_corridor_points (the only stage of the corridor generation that reads the SOGM)
  int N = DIRECT ? fd.pc_range[prob * 2 + 1] - fd.pc_range[prob * 2] : ws.seg_npts[slot];
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
  const double epsilon = DIRECT ? fd.epsilon : 1.0e-6;
  int          nH      = 0;
  const int    n_iter  = DIRECT ? fd.iterations : pp.firi_iterations;
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
    if constexpr (DIRECT)
      for (int k = 0; k < 3; ++k) r[k] = fd.r[prob * 3 + k];
    for (int loop = 0; loop < n_iter; ++loop) {
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
          {  // every point starts uncovered: the active lanes of this trip are exactly the points i < N of its word
            const unsigned long long act = __ballot(1);
            if (lane == 0) s_fw[i >> 6] = act;
          }
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
      unsigned bdFlags = M >= 32 ? 0xffffffffu : (1u << M) - 1u;
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
            s_fw[pcMinId >> 6] &= ~(1ull << (pcMinId & 63));
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
          const unsigned long long word = s_fw[j >> 6];  // uniform over the trip's lanes
          bool                     covered = false;
          if ((word >> lane) & 1ull) {
            const double *q = fpc + j * 3;
            if (((fh0 * q[0] + fh1 * q[1]) + fh2 * q[2]) + fh3 > -epsilon) {
              covered = true;
            } else {
              open = 1;
              if (lm > distR[j]) {
                lm = distR[j];
                li = j;
              }
            }
          }
          const unsigned long long clr = __ballot(covered);
          if (clr != 0ull && lane == 0) s_fw[j >> 6] = word & ~clr;
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
      if (loop == n_iter - 1) break;
      {
        const int       mm  = nH < LP_MAX_ROWS - 9 ? nH : LP_MAX_ROWS - 9;
        const long long tm0 = wall_clock64();
        maxVolInsEllipsoid(s_poly, mm, R, p, r, sc, dbg);
        if (lane == 0) dbg[7] = wall_clock64() - tm0;
      }
      __syncthreads();
    }
    if constexpr (DIRECT)
      if (lane == 0)
        for (int k = 0; k < 3; ++k) fd.r[prob * 3 + k] = r[k];
  }

  if constexpr (DIRECT) {
    int nf = seed_ok ? nH : 0;
    if (nf > fd.max_faces) {
      nf       = fd.max_faces;
      overflow = 1;
    }
    double *out = fd.hpoly + (size_t)prob * fd.max_faces * 4;
    for (int i = lane; i < nf * 4; i += 64) out[i] = s_poly[i];
    if (lane == 0) {
      fd.nfaces[prob] = nf;
      fd.status[prob] = overflow ? -3 : (seed_ok ? 1 : 0);
    }
    return;
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

__global__ __launch_bounds__(64) void k_corridor_segment(
    MapView m, SogmPlannerParams pp, CorridorWorkspace ws, const double *__restrict__ start_pva,
    const double *__restrict__ t_start, const double *__restrict__ route,
    const int32_t *__restrict__ route_len, int route_cap, int agent0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  corridor_segment_body<6, false>(m, pp, ws, start_pva, t_start, route, route_len, route_cap, blockIdx.y + agent0,
                                  blockIdx.x, smem, FiriDirect{});
}

#define FIRI_DIRECT_BD_MAX 32
__global__ __launch_bounds__(64) void k_firi_direct(MapView m, SogmPlannerParams pp, CorridorWorkspace ws,
                                                    FiriDirect fd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  corridor_segment_body<FIRI_DIRECT_BD_MAX, true>(m, pp, ws, nullptr, nullptr, nullptr, nullptr, 0, blockIdx.x, 0, smem,
                                                  fd);
}

// =================================================================================================
// Kernel B: per agent bookkeeping (baseline_fake.cpp:364-414 / baseline.cpp:362-403)
// =================================================================================================
__device__ __forceinline__ void corridor_finalize_body(const SogmPlannerParams &pp, const CorridorWorkspace &ws,
                                                       const double *start_pva, const double *route,
                                                       const int32_t *route_len, int route_cap, double *out_polys,
                                                       int32_t *out_nfaces, int32_t *out_npoly, double *out_goal,
                                                       int agent, const SolverScratch &sc) {
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
    if (state[i] != 1) {
      // a corridor that hit a capacity the reference does not have (pc_capacity points, 128 selected planes,
      // max_faces) counts as invalid: make that visible
      if (state[i] == -3 && w0 && ws.counters) atomicAdd(&ws.counters[SOGM_CNT_CORRIDOR_CAPACITY], 1ull);
      break;
    }
    ++npoly;
  }
  if (npoly == SOGM_MAX_PIECES && rl - 1 > SOGM_MAX_PIECES && w0 && ws.counters)
    atomicAdd(&ws.counters[SOGM_CNT_PIECES_CAPACITY], 1ull);  // route longer than 16 pieces: truncated
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

__global__ __launch_bounds__(64) void k_corridor_finalize(
    SogmPlannerParams pp, CorridorWorkspace ws, const double *__restrict__ start_pva,
    const double *__restrict__ route, const int32_t *__restrict__ route_len, int route_cap,
    double *__restrict__ out_polys, int32_t *__restrict__ out_nfaces,
    int32_t *__restrict__ out_npoly, double *__restrict__ out_goal, int agent0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double       *s_lp   = (double *)smem;
  double       *s_rows = s_lp + LP_WORK_DOUBLES;
  int          *s_perm = (int *)(s_rows + LP_MAX_ROWS * 5);
  SolverScratch sc{s_lp, s_perm, s_rows, nullptr};
  corridor_finalize_body(pp, ws, start_pva, route, route_len, route_cap, out_polys, out_nfaces, out_npoly, out_goal,
                         blockIdx.x + agent0, sc);
}

// =================================================================================================
// Dataflow kernel C (sogm_replan, pipelining modes 0 / 2 / 3): ONE persistent launch for the whole corridor stage.
// A workgroup (one wave) takes tickets; ticket k is segment k % 16 of the (k / 16)-th agent whose A* search has
// finished (k_astar publishes agents in completion order), so an agent's corridors start the moment ITS search is
// done — not when the slowest search of a group is.  The wave extracts the segment's obstacle points itself, runs
// FIRI / MVIE / shrink / validity, and the wave that completes an agent's last segment does the per-agent
// bookkeeping (adjacent intersections, goal projection) and publishes the agent to the QP kernel's ready list.
// Waiting is a bounded spin (s_sleep polling of one word); a timeout raises FlowCtl::err and every kernel of the
// tick drains.
// =================================================================================================
// Called by ALL lanes of a wave (uniform control flow; every value that steers a branch goes through
// readfirstlane so that the compiler sees a scalar condition): returns the published value, or -1 on failure.
__global__ __launch_bounds__(64) void k_corridor_flow(MapView m, SogmPlannerParams pp, CorridorWorkspace ws,
                                                      FlowCtl fc, const double *start_pva, const double *t_start,
                                                      const double *route, const int32_t *route_len,
                                                      int route_cap, double *out_polys, int32_t *out_nfaces,
                                                      int32_t *out_npoly, double *out_goal, int n_agents) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane  = threadIdx.x;
  const int total = n_agents * SOGM_MAX_PIECES;
  for (;;) {
    const int k = flow_ticket(&fc.hdr[FLOW_C_TICKET]);
    if (k >= total) break;
    const int agent = flow_wait_slot(fc.a_ready + k / SOGM_MAX_PIECES, &fc.hdr[FLOW_ERR]);
    if (agent < 0) break;  // timed out / another kernel failed: drain
    __threadfence();        // the search's outputs (route, route_len) were published before the ready slot
    const int seg = k % SOGM_MAX_PIECES;
    if (seg == 0 && lane == 0) fc.ts[agent * 8 + 2] = wall_clock64();
    if (seg < route_len[agent] - 1) {
      corridor_points_body<1>(m, pp, ws, start_pva, t_start, route, route_len, route_cap, agent, seg);
      __threadfence_block();
      __syncthreads();
    }
    corridor_segment_body<6, false>(m, pp, ws, start_pva, t_start, route, route_len, route_cap, agent, seg, smem,
                                    FiriDirect{});
    __syncthreads();
    __threadfence();  // this segment's polytope is visible before its completion is counted
    const int last = flow_ticket(&fc.seg_done[agent]) == SOGM_MAX_PIECES - 1;
    if (last) {
      __threadfence();  // the other segments' polytopes (their completions were counted before ours)
      double       *s_lp   = (double *)smem;
      double       *s_rows = s_lp + LP_WORK_DOUBLES;
      int          *s_perm = (int *)(s_rows + LP_MAX_ROWS * 5);  // the head of s_lm: free between segments
      SolverScratch sc{s_lp, s_perm, s_rows, nullptr};
      corridor_finalize_body(pp, ws, start_pva, route, route_len, route_cap, out_polys, out_nfaces, out_npoly,
                             out_goal, agent, sc);
      __syncthreads();
      if (lane == 0) fc.ts[agent * 8 + 3] = wall_clock64();
      __threadfence();
      if (lane == 0) {
        const int r = atomicAdd(&fc.hdr[FLOW_Q_READY_N], 1);
        __hip_atomic_store(fc.q_ready + r, agent, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
  }
}

size_t corridor_segment_lds(int pc_capacity) {
  return sizeof(double) * (LP_WORK_DOUBLES + LP_MAX_ROWS * 5 + 2 * 18 * 9 + 16 + 36 + 2 * FIRI_MAX_H * 4 + 96) +
         sizeof(int) * (LP_MAX_ROWS + 16) + 8 * (((size_t)pc_capacity + 63) / 64);
}
size_t firi_direct_lds(int pc_capacity) {
  return sizeof(double) * (LP_WORK_DOUBLES + LP_MAX_ROWS * 5 + 2 * 18 * 9 + 16 + 36 + 2 * FIRI_MAX_H * 4 +
                           firi_small_doubles<FIRI_DIRECT_BD_MAX>()) +
         sizeof(int) * (LP_MAX_ROWS + 16) + 8 * (((size_t)pc_capacity + 63) / 64);
}

// ------------------------------------------------------------------------------------------------
// ParticleATC::isSafeAfterOpt (traj_coordinator/src/particles.cpp:223-283): one workgroup per
// (other agent's record, ego agent).  Set A = the new trajectory's control points, set B = the record's
// control points from the piece that contains t_now onwards; separable iff the feasibility LP
// n.a + d >= 1, n.b + d <= -1 (utils/separator/src/separator_glpk.cpp:75-190, zero objective) has a
// solution — solved with the same sdlp LP as the corridor checks, the four variables boxed at +-1e4 (8 extra
// rows): the projective LP reports a feasible point at infinity (two crossing segments: the plane through both)
// as "-inf", GLPK's affine model calls that infeasible; with the box the result is finite or +inf.
// ------------------------------------------------------------------------------------------------
#define DECONFLICT_MAX_ROWS (LP_MAX_ROWS - 9)  // 144 point rows + 8 box rows + sdlp's plane 0
// One (new trajectory of agent a, record r) pair, executed by one wave: true = NOT separable (unsafe).
// ca: the agent's 5 M control points; s_lp / s_rows / s_perm: the wave's LP scratch in LDS.
__device__ __forceinline__ bool deconflict_pair_unsafe(const double *ca, int M, const SogmTrajRecord &r, int ego_id,
                                                       double now, double *s_lp, double *s_rows, int *s_perm,
                                                       unsigned long long *counters) {
  if (M <= 0) return false;  // nothing optimised for this agent
  if (r.n_pieces <= 0 || r.drone_id == ego_id) return false;
  double time_end = r.time_start;
  for (int k = 0; k < r.n_pieces; ++k) time_end += r.duration[k];
  if (!(r.time_start < now && now < time_end)) return false;
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
    if ((threadIdx.x & 63) == 0 && counters) atomicAdd(&counters[SOGM_CNT_DECONFLICT_CAPACITY], 1ull);
    return true;
  }
  double       *A = s_rows, *b = s_rows + LP_MAX_ROWS * 4;
  const double *cb = r.cpts + piece * 15;
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
    if (apart) return false;  // wave-uniform
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
      if (hiA < loB || hiB < loA) return false;  // wave-uniform
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
  if (threadIdx.x < 8) {  // x_k <= 1e4, -x_k <= 1e4
    const int q = nA + nB + threadIdx.x, k = threadIdx.x >> 1;
    for (int j = 0; j < 4; ++j) A[q * 4 + j] = j == k ? ((threadIdx.x & 1) ? -1.0 : 1.0) : 0.0;
    b[q] = 1.0e4;
  }
  wave_lds_sync();
  const double c[4] = {0, 0, 0, 0};
  double       x[4];
  const double v = linprog_wave<4>(c, nA + nB + 8, A, b, x, s_lp, s_perm);  // the whole wave solves the LP
  wave_lds_sync();
  return v == INFINITY || v == -INFINITY;
}

// The cheap part of deconflict_pair_unsafe with ONE LANE PER RECORD (the dataflow replan's finishing kernel: the
// wave-per-pair form above walks the swarm table record by record, two dependent global round trips each, and
// those round trips are the first thing a streaming clear beside it stretches).  Same decisions bit for bit: the
// activity tests, the bounding boxes and the 11 candidate-normal projections are minima / maxima of the same
// products, whatever the order.  sA: the agent's nA control points (LDS), boxA lo[3] hi[3], prA[q][2] = (lo, hi) of
// A's projection on the ten fixed normals.  Returns true when the pair still needs the full test (the LP, or the
// capacity rule), i.e. whenever deconflict_pair_unsafe would not return false before its LP assembly.
__device__ inline bool deconflict_prefilter(const double *sA, int nA, const double *boxA, const double (*prA)[2],
                                            const SogmTrajRecord &r, int ego_id, double now) {
  const int np = r.n_pieces;
  if (np <= 0 || r.drone_id == ego_id) return false;
  const double ts = r.time_start;
  double       time_end = ts;
  for (int k = 0; k < np; ++k) time_end += r.duration[k];
  if (!(ts < now && now < time_end)) return false;
  double t     = now - ts;
  int    piece = np - 1;
  for (int k = 0; k < np; ++k) {
    t -= r.duration[k];
    if (t < 0) {
      piece = k;
      break;
    }
  }
  const int nB = (np - piece) * 5;
  if (nA + nB > DECONFLICT_MAX_ROWS) return true;  // the capacity rule (and its counter) live in the full test
  const double *cb = r.cpts + piece * 15;
  const double  dirs[10][3] = {{1, 1, 0}, {1, -1, 0}, {1, 0, 1}, {1, 0, -1}, {0, 1, 1},
                               {0, 1, -1}, {1, 1, 1}, {1, 1, -1}, {1, -1, 1}, {1, -1, -1}};
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  double plo[10], phi[10];
#pragma unroll
  for (int q = 0; q < 10; ++q) {
    plo[q] = INFINITY;
    phi[q] = -INFINITY;
  }
#pragma unroll 4
  for (int i = 0; i < nB; ++i) {
    const double x = cb[i * 3], y = cb[i * 3 + 1], z = cb[i * 3 + 2];
    lo[0] = fmin(lo[0], x);
    hi[0] = fmax(hi[0], x);
    lo[1] = fmin(lo[1], y);
    hi[1] = fmax(hi[1], y);
    lo[2] = fmin(lo[2], z);
    hi[2] = fmax(hi[2], z);
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      const double v = (dirs[q][0] * x + dirs[q][1] * y) + dirs[q][2] * z;
      plo[q]         = fmin(plo[q], v);
      phi[q]         = fmax(phi[q], v);
    }
  }
  for (int k = 0; k < 3; ++k)
    if (boxA[3 + k] < lo[k] || hi[k] < boxA[k]) return false;  // disjoint bounding boxes
#pragma unroll
  for (int q = 0; q < 10; ++q)
    if (prA[q][1] < plo[q] || phi[q] < prA[q][0]) return false;
  // the line between the box centres
  double n[3];
  for (int k = 0; k < 3; ++k) n[k] = 0.5 * (lo[k] + hi[k]) - 0.5 * (boxA[k] + boxA[3 + k]);
  double loA = INFINITY, hiA = -INFINITY, loB = INFINITY, hiB = -INFINITY;
  for (int i = 0; i < nA; ++i) {
    const double v = (n[0] * sA[i * 3] + n[1] * sA[i * 3 + 1]) + n[2] * sA[i * 3 + 2];
    loA            = fmin(loA, v);
    hiA            = fmax(hiA, v);
  }
  for (int i = 0; i < nB; ++i) {
    const double v = (n[0] * cb[i * 3] + n[1] * cb[i * 3 + 1]) + n[2] * cb[i * 3 + 2];
    loB            = fmin(loB, v);
    hiB            = fmax(hiB, v);
  }
  if (hiA < loB || hiB < loA) return false;
  return true;
}

__global__ __launch_bounds__(64) void k_safe_after_opt(const double *__restrict__ cpts,
                                                       const int32_t *__restrict__ npoly,
                                                       const SogmTrajRecord *__restrict__ rec, int n_rec,
                                                       const int32_t *__restrict__ ego_ids,
                                                       const double *__restrict__ t_now,
                                                       int32_t *__restrict__ out_safe, int agent0,
                                                       unsigned long long *counters) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  double *s_lp   = s_dyn;                   // LP_WORK_DOUBLES
  double *s_rows = s_lp + LP_WORK_DOUBLES;  // LP_MAX_ROWS * 5
  int    *s_perm = (int *)(s_rows + LP_MAX_ROWS * 5);
  const int a = blockIdx.y + agent0, i = blockIdx.x;
  if (deconflict_pair_unsafe(cpts + (size_t)a * SOGM_MAX_PIECES * 15, npoly[a], rec[i], ego_ids[a], t_now[a], s_lp,
                             s_rows, s_perm, counters) &&
      threadIdx.x == 0)
    out_safe[a] = 0;  // every writer writes 0
}

// What the finishing role does for one agent (one wave): ParticleATC::isSafeAfterOpt against every record of the swarm
// (when a swarm is set), then the agent's SogmTrajRecord / ok flag and the publication — shared by k_finish_flow
// (sogm_replan) and k_flight_light (sogm_flight_run).  Returns bit 0 = safe, bit 1 = replan() returned true.
__device__ __forceinline__ int finish_agent(const FinishArgs &f, int a, double *s_lp, double *s_rows, int *s_perm, int lane) {
  const int M    = f.npoly[a];
  int       safe = 1;
  const bool solved = f.ret[a] != 0 && M > 0 && (f.status[a] == 1 || f.status[a] == 2);
  if (f.swarm && solved) {  // unsolved agents fail anyway: the check cannot change their outcome
    const double *ca  = f.cpts + (size_t)a * SOGM_MAX_PIECES * 15;
    const int     ego = f.swarm_ego[a];
    const double  now = f.swarm_now[a];
    const int     nA  = 5 * M;
    // set A once per agent: its points in LDS (the LP scratch is idle until a pair needs the LP), its box and
    // its projections on the ten fixed normals as wave-uniform values
    double *sA = s_lp;
    double  boxA[6], prA[10][2];
    auto stage_A = [&]() {
      for (int q = lane; q < nA * 3; q += 64) sA[q] = ca[q];
      wave_lds_sync();
    };
    stage_A();
    {
      const double dirs[10][3] = {{1, 1, 0}, {1, -1, 0}, {1, 0, 1}, {1, 0, -1}, {0, 1, 1},
                                  {0, 1, -1}, {1, 1, 1}, {1, 1, -1}, {1, -1, 1}, {1, -1, -1}};
      double lo[13], hi[13];
#pragma unroll
      for (int k = 0; k < 13; ++k) {
        lo[k] = INFINITY;
        hi[k] = -INFINITY;
      }
      for (int q = lane; q < nA; q += 64) {
        const double x = sA[q * 3], y = sA[q * 3 + 1], z = sA[q * 3 + 2];
        lo[0] = fmin(lo[0], x);
        hi[0] = fmax(hi[0], x);
        lo[1] = fmin(lo[1], y);
        hi[1] = fmax(hi[1], y);
        lo[2] = fmin(lo[2], z);
        hi[2] = fmax(hi[2], z);
#pragma unroll
        for (int k = 0; k < 10; ++k) {
          const double v = (dirs[k][0] * x + dirs[k][1] * y) + dirs[k][2] * z;
          lo[3 + k]      = fmin(lo[3 + k], v);
          hi[3 + k]      = fmax(hi[3 + k], v);
        }
      }
#pragma unroll
      for (int k = 0; k < 13; ++k)
        for (int d = 32; d >= 1; d >>= 1) {
          lo[k] = fmin(lo[k], __shfl_xor(lo[k], d, 64));
          hi[k] = fmax(hi[k], __shfl_xor(hi[k], d, 64));
        }
      for (int k = 0; k < 3; ++k) {
        boxA[k]     = lo[k];
        boxA[3 + k] = hi[k];
      }
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        prA[k][0] = lo[3 + k];
        prA[k][1] = hi[3 + k];
      }
    }
    // one lane per record decides whether the pair needs the full test; the few that do go through it one at a
    // time, in record order (first unsafe pair ends the check, as the sequential loop did)
    for (int i0 = 0; i0 < f.n_swarm && safe; i0 += 64) {
      const int  i    = i0 + lane;
      const bool need = i < f.n_swarm && deconflict_prefilter(sA, nA, boxA, prA, f.swarm[i], ego, now);
      unsigned long long m = __ballot(need);
      bool               lp_ran = false;
      while (m != 0 && safe) {
        const int j = __builtin_ctzll(m);
        m &= m - 1;
        if (deconflict_pair_unsafe(ca, M, f.swarm[i0 + j], ego, now, s_lp, s_rows, s_perm, f.counters)) safe = 0;
        lp_ran = true;
      }
      if (lp_ran && safe && i0 + 64 < f.n_swarm) stage_A();  // the LP used the scratch that held set A
    }
  }
#ifdef SOGM_FLOW_DEBUG
  if (lane == 0) f.out_safe[a] = 101;
#else
  if (lane == 0 && f.out_safe) f.out_safe[a] = safe;
#endif
  // BezierTraj record (plan_manager.cpp:364-399); n_pieces = 0 marks "replan() returned false".
  // Publication (sogm_planner_set_publish): a successful replan's record also goes — in the same stores — into
  // the host's own table (latest wins; a failed replan keeps executing the previous trajectory, :176-196) and the
  // agent's current record into the next tick's swarm table, HERE, where the stores overlap with the other agents'
  // chains, instead of in a store-heavy kernel after the replan (beside the streaming clear that kernel took 2 ms).
  const bool      good = solved && safe != 0;
  SogmTrajRecord *dst[3] = {f.out + a, (good && f.pub_own) ? f.pub_own + a : nullptr,
                            (good && f.pub_own && f.pub_table) ? f.pub_table + a : nullptr};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    SogmTrajRecord *r = dst[k];
    if (!r) continue;  // wave-uniform
    for (int i = lane; i < SOGM_MAX_PIECES; i += 64) r->duration[i] = (good && i < M) ? f.corridor_tau : 0.0;
    for (int i = lane; i < SOGM_MAX_PIECES * 15; i += 64)
      r->cpts[i] = (good && i < M * 15) ? f.cpts[(size_t)a * SOGM_MAX_PIECES * 15 + i] : 0.0;
    if (lane == 0) {
      r->drone_id   = f.drone_ids[a];
      r->time_start = f.t_start[a];
      r->n_pieces   = good ? M : 0;
    }
  }
  if (!good && f.pub_own && f.pub_table) {  // the table still lists the trajectory the agent goes on executing
    constexpr int W   = (int)(sizeof(SogmTrajRecord) / 16);
    const uint4  *src = reinterpret_cast<const uint4 *>(f.pub_own + a);
    uint4        *t   = reinterpret_cast<uint4 *>(f.pub_table + a);
    for (int w = lane; w < W; w += 64) t[w] = src[w];
  }
  if (lane == 0) f.out_ok[a] = good ? 1 : 0;
  return (safe ? 1 : 0) | (good ? 2 : 0);
}

// where this replan ended (baseline_fake.cpp: :292 no path, :405-419 corridors, :447 QP, :455 unsafe); one lane
__device__ __forceinline__ void finish_count(const FinishArgs &f, int a, bool safe) {
  if (!f.counters) return;
  const int M = f.npoly[a];
  int       k = SOGM_CNT_REPLAN_OK;
  if (f.ret[a] == 0) k = SOGM_CNT_FAIL_SEARCH;
  else if (M <= 0) k = SOGM_CNT_FAIL_CORRIDOR;
  else if (!(f.status[a] == 1 || f.status[a] == 2)) k = SOGM_CNT_FAIL_QP;
  else if (!safe) k = SOGM_CNT_FAIL_UNSAFE;
  atomicAdd(&f.counters[k], 1ull);
}
// Dataflow kernel F (sogm_replan): ONE persistent launch; a wave takes the agent whose QP finished ticket-th, runs
// finish_agent (what k_safe_after_opt + k_pack_records do in the grouped path) and hands the agent to the pre-stamp.
__global__ __launch_bounds__(64) void k_finish_flow(FlowCtl fc, FinishArgs f, int n_agents) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  double   *s_lp   = s_dyn;
  double   *s_rows = s_lp + LP_WORK_DOUBLES;
  int      *s_perm = (int *)(s_rows + LP_MAX_ROWS * 5);
  const int lane   = threadIdx.x;
  for (;;) {
    const int k = flow_ticket(&fc.hdr[FLOW_F_TICKET]);
    if (k >= n_agents) break;
    const int a = flow_wait_slot(fc.f_ready + k, &fc.hdr[FLOW_ERR]);
    if (a < 0) break;
    __threadfence();
    const int  code = finish_agent(f, a, s_lp, s_rows, s_perm, lane);
    const bool safe = (code & 1) != 0;
    if (fc.p_ready) {  // the agent's record is final: hand it to the pre-stamp
      __threadfence();
      if (lane == 0) {
        const int r = atomicAdd(&fc.hdr[FLOW_P_READY_N], 1);
        __hip_atomic_store(fc.p_ready + r, a, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (lane == 0) {
      fc.ts[a * 8 + 6] = wall_clock64();
      finish_count(f, a, safe);
    }
  }
}

// Flight kernel L (sogm_flight_run): role-less one-wave workgroups over the corridor + finish work queue.  A wave takes a
// ticket (one atomicAdd), waits for the descriptor at that position and does what it says.  WK_CORRIDOR: one of the 16
// segment slots of an agent whose search is done, as in k_corridor_flow; the wave that completes the agent's last slot
// finalises its corridors and queues the QP.  WK_FINISH: finish_agent against table ver(k - 2), the record into ver(k) and
// the flight log, then the tick's accounting and the agent's next map head.  17 descriptors per (agent, tick).
__global__ __launch_bounds__(64) void k_flight_light(MapView m, SogmPlannerParams pp, CorridorWorkspace ws, FlightCtl fl,
                                                     FlightLightDev d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int      lane  = threadIdx.x;
  // ONE FIFO of 17 descriptors per agent-tick (16 corridor segments + the finish); a worker whose ticket lies beyond the
  // flight's last descriptor leaves at once.  (Tried at the end of round 5 and taken back: a second, priority queue for the
  // finishes — 50 us of work that end an agent's tick; in the FIFO they wait 0.1-0.2 ms behind corridor segments — which
  // every worker looked at first and whose descriptor count per queue is not known in advance, so that the idle workers
  // stayed resident until the call's end.  Single-process flights: records identical, finish 0.2 -> 0.05 ms per agent-tick,
  // tick 6.99 -> 6.83 ms.  The two-rank flight of tests/test_exchange_gpu.py (two-tick calls): from the second call on two
  // corridor segments per call stayed unprocessed for 3 s — with every exit rule tried (finished count, epoch word, both
  // queues' known totals), with the queue's head / tail never reset, and NOT with all but 64 of the idle workers leaving
  // early.  Unexplained; profiles/EXPERIMENTS.md.)
  const unsigned total = (unsigned)fl.n_agents * (unsigned)fl.n_ticks * (SOGM_MAX_PIECES + 1);
  int           *err   = &fl.hdr[FL_ERR];
  fl_wg_started(fl, 2);
  long long      c1_prev = 0;
  int            kind_prev = 0;
  for (;;) {
    const unsigned t = (unsigned)flow_ticket(&fl.hdr[FL_LW_HEAD]);
    if (t >= total) break;
    const long long c0   = wall_clock64();
    const int       desc = wq_take(fl.lw, t, err);
    if (desc < 0) break;
    __threadfence();
    const long long c1 = wall_clock64();
    if (lane == 0) {  // wave time by activity (sogm_flight_stats)
      if (c1_prev) {
        atomicAdd(&fl.prof[kind_prev == WK_FINISH ? 8 : 7], (unsigned long long)(c0 - c1_prev));
        atomicAdd(&fl.prof[kind_prev == WK_FINISH ? 15 : 14], 1ull);
      }
      atomicAdd(&fl.prof[6], (unsigned long long)(c1 - c0));
    }
    const int kind = desc >> 28, a = desc & 0xFFFF;
    c1_prev   = c1;
    kind_prev = kind;
    if (kind == WK_FINISH) {
      const int k  = fl.tick_of[a];
      const int kl = k - fl.first_tick;
      FinishArgs f = d.fin;
      f.swarm      = d.tables ? d.tables + (size_t)((k - fl.lag) & 3) * d.n_total : nullptr;
      f.pub_table  = d.tables ? d.tables + (size_t)(k & 3) * d.n_total + d.agent0 : nullptr;
      f.out        = d.log_records + (size_t)kl * fl.n_agents;
      f.out_ok     = d.log_ok + (size_t)kl * fl.n_agents;
      double *s_lp = (double *)smem, *s_rows = s_lp + LP_WORK_DOUBLES;
      int    *s_perm = (int *)(s_rows + LP_MAX_ROWS * 5);
      const int code = finish_agent(f, a, s_lp, s_rows, s_perm, lane);
      __syncthreads();
      __threadfence();  // the record (own, ver(k), log) is out before the tick counts as finished
      if (lane == 0) {
        const long long now = wall_clock64();
        long long      *ts  = fl.ts + (size_t)a * FL_TS, *acc = fl.acc + (size_t)a * 8;
        ts[6]               = now;
        acc[0] += ts[14] - ts[10];                       // gate wait: the overlay parked behind the finished marks
        acc[1] += (ts[11] - ts[7]) - (ts[14] - ts[10]);  // map: from the head's publication to the complete map, less that
        acc[2] += ts[1] - ts[11];                      // search queue + A*
        acc[3] += ts[3] - ts[1];                       // corridor queue + corridors
        acc[4] += ts[5] - ts[3];                       // QP queue + QP
        acc[5] += now - ts[5];                         // finish queue + finish
        acc[6] += now - ts[7];                         // the whole chain
        acc[7] += 1;
        if (fl.ts_log) {
          long long *lg = fl.ts_log + ((size_t)kl * fl.n_agents + a) * FL_TS;
          for (int q = 0; q < FL_TS; ++q) lg[q] = ts[q];
        }
        finish_count(f, a, (code & 1) != 0);
        // The gate of the staleness rule — tick j reads table ver(j - 2), so every agent must have finished tick j - 2 — sits
        // in FRONT OF THE OVERLAY, the only phase of a map that reads the table (k_flight_map): an agent goes on to its next
        // map head at once and builds reset / bits / marks while it waits; the overlay of a map that reaches the gate early
        // is PARKED there, not queued, and the finish that completes the awaited tick queues the parked overlays.  (First
        // version: the whole head parked at the gate.  A gate then released a burst of 20-60 heads into the admission order
        // and the map workers at once, and the laggards arriving just then — the very agents the next gate waits for —
        // queued behind it: parked + admission + map 1.9 ms per tick of the flight's critical path.)
        const int A_      = fl.n_agents;
        const int done_kl = __hip_atomic_fetch_add(&fl.tick_done[kl], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1;
        if (atomicAdd(&fl.hdr[FL_FINISHED], 1) + 1 == A_ * fl.n_ticks)  // the call's last agent-tick: the map kernel's waves may go
          __hip_atomic_store(&fl.hdr[FL_END], fl.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (kl + 1 < fl.n_ticks) {  // the agent's next tick: its map head, now
          fl.tick_of[a] = k + 1;
          ts[7]         = now;
          // one of the last finishers of its tick: its next map goes through the urgent lane
          const bool urgent = fl.n_urgent > 0 && done_kl > A_ - fl.n_urgent;
          fl.urgent[a]      = urgent ? 1 : 0;
          fl_publish(urgent ? fl.u_ring : fl.m_ring, fl.ring_mask, &fl.hdr[urgent ? FL_U_READY : FL_M_READY], a);
        }
        // tick k is complete: the overlays of tick k + 2 parked so far may go.  (Several ranks with the exchange behind the
        // call: the gate is the all-gather of ver(k), and the kernel behind that collective releases them — k_flight_xsignal.)
        if (done_kl == A_ && kl + fl.lag < fl.n_ticks && !fl.xready) fl_gate_release(fl, kl + fl.lag);
      }
      continue;
    }
    // ---- WK_CORRIDOR: segment slot `seg` of agent a ----
    const int seg = (desc >> 16) & 0xFFF;
    if (seg == 0 && lane == 0) fl.ts[a * FL_TS + 2] = wall_clock64();
    // diagnostics (sogm_debug_corridor_stats, tools/soak_flight.py): when this slot's descriptor was taken, its obstacle points
    // were out, its segment was done, and on which compute unit (HW_ID | XCC_ID << 32)
    long long *sdbg = ws.seg_dbg + ((size_t)a * SOGM_MAX_PIECES + seg) * 16;
    if (lane == 0) {
      sdbg[12] = c1;
      sdbg[13] = 0;
      sdbg[14] = 0;
      sdbg[15] = (long long)(unsigned)__builtin_amdgcn_s_getreg(63492) | ((long long)(unsigned)__builtin_amdgcn_s_getreg(63508) << 32);
    }
    if (seg < d.route_len[a] - 1) {
      corridor_points_body<1>(m, pp, ws, d.start_pva, d.t_start, d.route, d.route_len, d.route_cap, a, seg);
      __threadfence_block();
      __syncthreads();
    }
    if (lane == 0) sdbg[13] = wall_clock64();
    corridor_segment_body<6, false>(m, pp, ws, d.start_pva, d.t_start, d.route, d.route_len, d.route_cap, a, seg, smem,
                                    FiriDirect{});
    __syncthreads();
    if (lane == 0) sdbg[14] = wall_clock64();
    __threadfence();
    const int last = (flow_ticket(&fl.seg_done[a]) & (SOGM_MAX_PIECES - 1)) == SOGM_MAX_PIECES - 1;
    if (last) {
      __threadfence();
      double       *s_lp   = (double *)smem;
      double       *s_rows = s_lp + LP_WORK_DOUBLES;
      int          *s_perm = (int *)(s_rows + LP_MAX_ROWS * 5);
      SolverScratch sc{s_lp, s_perm, s_rows, nullptr};
      corridor_finalize_body(pp, ws, d.start_pva, d.route, d.route_len, d.route_cap, d.out_polys, d.out_nfaces, d.out_npoly,
                             d.out_goal, a, sc);
      __syncthreads();
      if (lane == 0) {
        fl.ts[a * FL_TS + 3] = wall_clock64();
        fl_publish(fl.q_ring, fl.ring_mask, &fl.hdr[FL_Q_READY], a);
      }
    }
    __syncthreads();
  }
}

// residency gate of the dataflow replan: returns when every A* workgroup has started (see k_astar)
__global__ void k_flow_gate(FlowCtl fc, int expected) {
  if (threadIdx.x != 0) return;
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(&fc.hdr[FLOW_A_RESIDENT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > FLOW_TIMEOUT_TICKS) {
      atomicExch(&fc.hdr[FLOW_ERR], 1);
      break;
    }
  }
}



// sdlp::linprog<d> for a batch of independent LPs: one wave per problem (sogm_linprog_batched)
template <int D>
__global__ __launch_bounds__(64) void k_linprog(const double *__restrict__ c, const double *__restrict__ A,
                                                const double *__restrict__ b,
                                                const int32_t *__restrict__ row_range,
                                                double *__restrict__ out_x, double *__restrict__ out_min) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  double   *s_lp   = s_dyn;                   // LP_WORK_DOUBLES
  double   *s_rows = s_lp + LP_WORK_DOUBLES;  // LP_MAX_ROWS * 5
  int      *s_ord  = (int *)(s_rows + LP_MAX_ROWS * 5);
  const int p = blockIdx.x, lane = threadIdx.x;
  const int r0 = row_range[2 * p], rows = row_range[2 * p + 1] - r0;
  if (rows >= LP_MAX_ROWS || rows < 0) {  // capacity: NaN, never a silent answer
    if (lane == 0) {
      out_min[p] = __builtin_nan("");
      for (int j = 0; j < D; ++j) out_x[p * D + j] = __builtin_nan("");
    }
    return;
  }
  double *sA = s_rows, *sb = s_rows + LP_MAX_ROWS * 4;
  for (int i = lane; i < rows; i += 64) {
    for (int j = 0; j < D; ++j) sA[i * D + j] = A[(size_t)(r0 + i) * D + j];
    sb[i] = b[r0 + i];
  }
  wave_lds_sync();
  double cv[D], x[D];
  for (int j = 0; j < D; ++j) cv[j] = c[p * D + j];
  const double v = linprog_wave<D>(cv, rows, sA, sb, x, s_lp, s_ord);
  if (lane == 0) {
    out_min[p] = v;
    for (int j = 0; j < D; ++j) out_x[p * D + j] = x[j];
  }
}

__global__ void k_fill_i32(int32_t *p, int n, int32_t v, int agent0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[agent0 + i] = v;
}

int launch_deconflict(int n_agents, const double *cpts, const int32_t *npoly, const SogmTrajRecord *rec,
                      int n_rec, const int32_t *ego_ids, const double *t_now, int32_t *out_safe,
                      hipStream_t st, int agent0, unsigned long long *counters) {
  hipLaunchKernelGGL(k_fill_i32, dim3((n_agents + 63) / 64), dim3(64), 0, st, out_safe, n_agents, 1, agent0);
  if (n_rec > 0) {
    const size_t lds = sizeof(double) * (LP_WORK_DOUBLES + LP_MAX_ROWS * 5) + sizeof(int) * LP_MAX_ROWS;
    hipLaunchKernelGGL(k_safe_after_opt, dim3(n_rec, n_agents), dim3(64), lds, st, cpts, npoly, rec, n_rec,
                       ego_ids, t_now, out_safe, agent0, counters);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_flow_gate(const FlowCtl &fc, int expected, hipStream_t st) {
  hipLaunchKernelGGL(k_flow_gate, dim3(1), dim3(64), 0, st, fc, expected);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_corridor_flow(const MapView &m, const SogmPlannerParams &pp, const CorridorWorkspace &ws,
                         const FlowCtl &fc, int n_agents, int n_workgroups, const double *start_pva,
                         const double *t_start, const double *route, const int32_t *route_len, int route_cap,
                         double *out_polys, int32_t *out_nfaces, int32_t *out_npoly, double *out_goal,
                         hipStream_t st) {
  hipLaunchKernelGGL(k_corridor_flow, dim3(n_workgroups), dim3(64), corridor_segment_lds(pp.pc_capacity), st, m, pp,
                     ws, fc, start_pva, t_start, route, route_len, route_cap, out_polys, out_nfaces, out_npoly,
                     out_goal, n_agents);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_finish_flow(const FlowCtl &fc, int n_agents, int n_workgroups, double corridor_tau, const int32_t *ret,
                       const int32_t *npoly, const int32_t *status, const double *cpts, const SogmTrajRecord *swarm,
                       int n_swarm, const int32_t *swarm_ego, const double *swarm_now, const double *t_start,
                       const int32_t *drone_ids, SogmTrajRecord *out, int32_t *out_ok, int32_t *out_safe,
                       unsigned long long *counters, hipStream_t st, SogmTrajRecord *pub_own,
                       SogmTrajRecord *pub_table) {
  const size_t lds = sizeof(double) * (LP_WORK_DOUBLES + LP_MAX_ROWS * 5) + sizeof(int) * LP_MAX_ROWS;
  const FinishArgs f{corridor_tau, ret, npoly, status, cpts, swarm, n_swarm, swarm_ego, swarm_now, t_start, drone_ids,
                     out, out_ok, out_safe, counters, pub_own, pub_table};
  hipLaunchKernelGGL(k_finish_flow, dim3(n_workgroups), dim3(64), lds, st, fc, f, n_agents);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_flight_light(const MapView &m, const SogmPlannerParams &pp, const CorridorWorkspace &ws, const FlightCtl &fl,
                        const FlightLightDev &d, int n_workgroups, hipStream_t st) {
  hipLaunchKernelGGL(k_flight_light, dim3(n_workgroups), dim3(64), corridor_segment_lds(pp.pc_capacity), st, m, pp, ws, fl, d);
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

extern "C" int sogm_linprog_batched(int d, const double *c, const double *A, const double *b,
                                    const int32_t *row_range, int n, double *out_x, double *out_min,
                                    void *stream) {
  if ((d != 3 && d != 4) || n < 0 || (n > 0 && (!c || !row_range || !out_x || !out_min)))
    return SOGM_ERR_INVALID_ARG;
  if (n == 0) return SOGM_OK;
  const size_t lds = sizeof(double) * (LP_WORK_DOUBLES + LP_MAX_ROWS * 5) + sizeof(int) * LP_MAX_ROWS;
  hipStream_t  st  = (hipStream_t)stream;
  if (d == 3)
    hipLaunchKernelGGL(sogm::k_linprog<3>, dim3(n), dim3(64), lds, st, c, A, b, row_range, out_x, out_min);
  else
    hipLaunchKernelGGL(sogm::k_linprog<4>, dim3(n), dim3(64), lds, st, c, A, b, row_range, out_x, out_min);
  SOGM_HIP_CHECK(hipGetLastError());
  return SOGM_OK;
}

extern "C" int sogm_firi_batched(const double *bd, int n_bd, const double *pc_xyz, const int32_t *pc_range,
                                 const double *a, const double *b, double *r, int iterations, double epsilon, int n,
                                 int max_points, int max_faces, double *out_hpoly, int32_t *out_nfaces,
                                 int32_t *out_status, void *stream) {
  if (n < 0 || n_bd < 1 || n_bd > FIRI_DIRECT_BD_MAX || iterations < 1 || max_points < 1 || max_points > 16384 ||
      max_faces < 1 || max_faces > FIRI_MAX_H ||
      (n > 0 && (!bd || !pc_range || !a || !b || !r || !out_hpoly || !out_nfaces || !out_status)))
    return SOGM_ERR_INVALID_ARG;
  if (n == 0) return SOGM_OK;
  hipStream_t st = (hipStream_t)stream;
  // per-problem scratch (ellipsoid-frame points, tangent planes, distances, debug words): stream-ordered
  const size_t per = sizeof(double) * 8 * (size_t)max_points + sizeof(long long) * 16;
  char        *scratch = nullptr;
  SOGM_HIP_CHECK(hipMallocAsync((void **)&scratch, per * (size_t)n, st));
  sogm::CorridorWorkspace ws{};
  ws.fpc     = (double *)scratch;
  ws.tang    = ws.fpc + (size_t)n * max_points * 3;
  ws.distr   = ws.tang + (size_t)n * max_points * 4;
  ws.seg_dbg = (long long *)(ws.distr + (size_t)n * max_points);
  SogmPlannerParams pp{};
  pp.pc_capacity = max_points;
  sogm::FiriDirect fd{bd, n_bd, pc_xyz, pc_range, a, b, r, iterations, out_hpoly, out_nfaces, out_status,
                      max_faces, 0, epsilon};
  hipLaunchKernelGGL(sogm::k_firi_direct, dim3(n), dim3(64), sogm::firi_direct_lds(max_points), st, sogm::MapView{},
                     pp, ws, fd);
  const hipError_t e = hipGetLastError();
  (void)hipFreeAsync(scratch, st);
  SOGM_HIP_CHECK(e);
  return SOGM_OK;
}

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
  if (hipMemcpyFromSymbol(out2 + 2, HIP_SYMBOL(sogm::g_lbfgs_prof), 40) != hipSuccess) return -1;
  unsigned long long z[5] = {0, 0, 0, 0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(sogm::g_mvie_prof), z, 16);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(sogm::g_lbfgs_prof), z, 40);
  return 0;
}
#endif
