// sogm_qp.hip — batched corridor-constrained min-jerk Bezier QP, OSQP-algorithm ADMM, gfx950.
//
// Reference: traj_opt::BezierOpt::setup / optimize (traj_opt/src/bezier_optimizer.cpp:27-285)
// through IOSQP (traj_opt/include/iosqp.hpp:40-115) into OSQP (external, v0.6 API, not vendored).
// The solver restates the published OSQP algorithm exactly as oracle/qp_oracle.cpp does.
//
// Mapping to CDNA4: one workgroup (256 lanes = 4 waves) per agent; problems are tiny
// (n = 15 M <= 240 variables, m ~ 5e2..5.6e3 rows with <= 6 non-zeros each) and strictly
// latency-bound, so the point is to keep the whole iteration on-chip and off the host:
//   * A is built directly in ELL form (row-parallel) plus a row-sorted CSC index for A^T products;
//   * P is block diagonal (one 15x15 min-jerk block per piece) and lives in LDS;
//   * K = P + sigma I + A^T diag(rho) A is block-banded (half bandwidth 17: continuity rows couple
//     the last 3 control points of a piece with the first 3 of the next) — stored as an n x 18 band
//     in LDS, factored by a banded Cholesky on one wave, re-factored only when rho changes;
//   * per iteration: CSC column sums (A^T w), banded forward/back substitution (wave 0, shuffle
//     reductions), ELL row products (A x), element-wise updates; residual norms are wavefront
//     shuffle reductions combined through LDS.
// No dense contraction anywhere -> no MFMA.  m-sized vectors stay in per-agent HBM scratch
// (L2-resident, ~0.4 MB).
#include <hip/hip_runtime.h>

#include "../../include/sogm_detmath.h"
#include "sogm_planner.hpp"

namespace sogm {
namespace {

#define QP_BW 17                 // half bandwidth of K
#define QP_NMAX (15 * SOGM_MAX_PIECES)
#define QP_ELL 6

__device__ const double OSQP_INFTY  = 1e30;
__device__ const double MIN_SCALING = 1e-04, MAX_SCALING = 1e+04;
__device__ const double RHO_MIN = 1e-06, RHO_TOL = 1e-04, RHO_EQ_OVER_RHO_INEQ = 1e03;

__device__ inline double dabs(double x) { return x < 0 ? -x : x; }
__device__ inline double dmax(double a, double b) { return a > b ? a : b; }
__device__ inline double limit_scaling(double v) {
  v = v < MIN_SCALING ? 1.0 : v;
  v = v > MAX_SCALING ? MAX_SCALING : v;
  return v;
}

__device__ inline double wave_max(double v) {
  for (int d = 32; d >= 1; d >>= 1) v = dmax(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ inline double wave_sum(double v) {
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
// block-wide max over 256 lanes; every lane gets the result.  s_red: 4 doubles.
__device__ inline double block_max(double v, double *s_red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return dmax(dmax(s_red[0], s_red[1]), dmax(s_red[2], s_red[3]));
}

struct Prob {
  int            n, m, M;
  int           *ecol;  // [m][6]
  double        *eval;  // [m][6]
  double        *l, *u, *rho, *E, *z, *zp, *zt, *y, *w, *dy;
  int           *cptr;  // [n+1]
  int           *cidx;  // [nnz]  row * 8 + slot, sorted by row inside a column
};

// ---- K band helpers: Kb[i * 18 + (i - j)], 0 <= i - j <= 17
__device__ inline double &KB(double *Kb, int i, int j) { return Kb[i * (QP_BW + 1) + (i - j)]; }

}  // namespace

__global__ __launch_bounds__(256) void k_qp(SogmPlannerParams pp, SogmQpSettings qs, QpWorkspace ws,
                                            QpConst qc, const double *__restrict__ start_pva,
                                            const double *__restrict__ goal_pv,
                                            const double *__restrict__ polys,
                                            const int32_t *__restrict__ nfaces,
                                            const int32_t *__restrict__ npoly,
                                            double *__restrict__ out_cpts,
                                            int32_t *__restrict__ out_status,
                                            int32_t *__restrict__ out_iters) {
  const int agent = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = npoly[agent];
  if (M <= 0 || M > SOGM_MAX_PIECES) {
    if (tid == 0) {
      out_status[agent] = -100;  // nothing to solve (an earlier stage failed)
      out_iters[agent]  = 0;
    }
    return;
  }
  const int MF = pp.max_faces;
  const int n  = 15 * M;

  extern __shared__ __attribute__((aligned(16))) char qp_smem[];
  double *s_Kb = (double *)qp_smem;            // [n][18]  K band / Cholesky factor
  double *s_P  = s_Kb + (size_t)n * (QP_BW + 1);  // [M][225] scaled cost blocks
  __shared__ double s_ginv[QP_NMAX];
  __shared__ double s_x[QP_NMAX], s_xp[QP_NMAX], s_xt[QP_NMAX], s_D[QP_NMAX], s_Dt[QP_NMAX];
  __shared__ double s_cn[QP_NMAX];  // column norms / scratch
  __shared__ double s_red[4];
  __shared__ double s_sc[8];
  __shared__ int    s_off[SOGM_MAX_PIECES + 1];
  __shared__ int    s_cnt[QP_NMAX + 1];
  __shared__ int    s_cptr[QP_NMAX + 1];
  __shared__ int    s_flag;

  // ---- problem dimensions
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < M; ++i) {
      s_off[i] = acc;
      acc += 5 * nfaces[agent * SOGM_MAX_PIECES + i];
    }
    s_off[M] = acc;
  }
  __syncthreads();
  const int R1 = 3 * (M + 1), R3 = 9 * (M + 1), R4 = R3 + 12 * M,
            R5 = R4 + 9 * M;
  const int m = R5 + s_off[M];
  Prob      pb;
  pb.n = n;
  pb.m = m;
  pb.M = M;
  {
    // Row data (8 m-vectors, ELL rows, CSC index) live in LDS when they fit next to K and P —
    // the common case (m ~ 600) — and in per-agent HBM scratch otherwise (same code, flat pointers).
    const size_t head  = ((size_t)n * (QP_BW + 1) + (size_t)M * 225) * sizeof(double);
    const size_t perow = 8 * sizeof(double) + QP_ELL * (sizeof(double) + 2 * sizeof(int));
    const size_t need  = head + (size_t)m * perow + 64;
    const size_t mc    = ws.m_cap;
    double      *v;
    if (need <= (size_t)ws.dyn_lds_bytes) {
      v         = (double *)(qp_smem + head);
      pb.eval   = v + 8 * (size_t)m;
      pb.ecol   = (int *)(pb.eval + (size_t)m * QP_ELL);
      pb.cidx   = pb.ecol + (size_t)m * QP_ELL;
      pb.l = v;
      pb.u = v + m;
      pb.rho = v + 2 * (size_t)m;
      pb.E = v + 3 * (size_t)m;
      pb.z = v + 4 * (size_t)m;
      pb.y = v + 5 * (size_t)m;
      pb.w = v + 6 * (size_t)m;
      pb.dy = v + 7 * (size_t)m;
    } else {
      pb.ecol = ws.ell_col + (size_t)agent * mc * QP_ELL;
      pb.eval = ws.ell_val + (size_t)agent * mc * QP_ELL;
      v       = ws.mvec + (size_t)agent * mc * 10;
      pb.l = v;
      pb.u = v + mc;
      pb.rho = v + 2 * mc;
      pb.E = v + 3 * mc;
      pb.z = v + 4 * mc;
      pb.y = v + 5 * mc;
      pb.w = v + 6 * mc;
      pb.dy = v + 7 * mc;
      pb.cidx = ws.csc_idx + (size_t)agent * mc * QP_ELL;
    }
    pb.zp = pb.zt = nullptr;
    pb.cptr = s_cptr;
  }
  const double *sp   = start_pva + agent * 9;
  const double *gp   = goal_pv + agent * 6;
  const double  tau  = pp.corridor_tau;  // time_alloc: every piece = corridor_tau (baseline.cpp:411)
  const double  vmax = pp.opt_max_vel, amax = pp.opt_max_acc;

  // ---- 1. assembly of A, l, u in ELL (bezier_optimizer.cpp:113-260), one lane per row
  for (int r = tid; r < m; r += 256) {
    int    col[QP_ELL];
    double val[QP_ELL];
    for (int k = 0; k < QP_ELL; ++k) {
      col[k] = -1;
      val[k] = 0.0;
    }
    double lo = 0.0, hi = 0.0;
    const double p2a[3] = {12, -24, 12};
    if (r < R3) {
      const int kind = r / R1;       // 0 pos, 1 vel, 2 acc
      const int rr   = r - kind * R1;
      const int gidx = rr / 3, d = rr % 3;  // 0 = start, 1..M-1 = knots, M = end
      if (kind == 0) {
        if (gidx == 0) {
          col[0] = d;
          val[0] = 1;
          lo = hi = sp[d];
        } else if (gidx == M) {
          col[0] = M * 15 - 3 + d;
          val[0] = 1;
          lo = hi = gp[d];
        } else {
          col[0] = gidx * 15 + d;
          val[0] = 1;
          col[1] = gidx * 15 - 3 + d;
          val[1] = -1;
        }
      } else if (kind == 1) {
        if (gidx == 0) {
          col[0] = d;
          val[0] = -4;
          col[1] = 3 + d;
          val[1] = 4;
          lo = hi = sp[3 + d] * tau;
        } else if (gidx == M) {
          col[0] = M * 15 - 6 + d;
          val[0] = -4;
          col[1] = M * 15 - 3 + d;
          val[1] = 4;
          lo = hi = gp[3 + d] * tau;
        } else {
          col[0] = gidx * 15 + d;
          val[0] = -4.0 / tau;
          col[1] = gidx * 15 + 3 + d;
          val[1] = 4.0 / tau;
          col[2] = gidx * 15 - 3 + d;
          val[2] = -4.0 / tau;
          col[3] = gidx * 15 - 6 + d;
          val[3] = 4.0 / tau;
        }
      } else {
        if (gidx == 0) {
          for (int k = 0; k < 3; ++k) {
            col[k] = k * 3 + d;
            val[k] = p2a[k];
          }
          lo = hi = sp[6 + d] * tau * tau;
        } else if (gidx == M) {
          for (int k = 0; k < 3; ++k) {
            col[k] = M * 15 - 9 + k * 3 + d;
            val[k] = p2a[k];
          }
          lo = hi = 0.0 * tau * tau;  // final acceleration = 0 (baseline.cpp:423)
        } else {
          const double t2 = tau * tau;
          for (int k = 0; k < 3; ++k) {
            col[k]     = gidx * 15 + k * 3 + d;
            val[k]     = p2a[k] / t2;
            col[3 + k] = gidx * 15 - 9 + k * 3 + d;
            val[3 + k] = -p2a[k] / t2;
          }
        }
      }
    } else if (r < R4) {
      const int rr = r - R3, i = rr / 12, j = (rr % 12) / 3, d = rr % 3;
      col[0] = i * 15 + j * 3 + d;
      val[0] = -4;
      col[1] = i * 15 + j * 3 + 3 + d;
      val[1] = 4;
      hi     = vmax * 1.0 * tau;
      lo     = -vmax * 1.0 * tau;
    } else if (r < R5) {
      const int rr = r - R4, i = rr / 9, j = (rr % 9) / 3, d = rr % 3;
      for (int k = 0; k < 3; ++k) {
        col[k] = i * 15 + j * 3 + k * 3 + d;
        val[k] = p2a[k];
      }
      hi = amax * 1.0 * tau * tau;
      lo = -amax * 1.0 * tau * tau;
    } else {
      const int rr = r - R5;
      int       i  = 0;
      while (i + 1 < M && rr >= s_off[i + 1]) ++i;
      const int     q = rr - s_off[i], face = q / 5, k = q % 5;
      const double *h = polys + (((size_t)agent * SOGM_MAX_PIECES + i) * MF + face) * 4;
      for (int d = 0; d < 3; ++d) {
        col[d] = i * 15 + k * 3 + d;
        val[d] = h[d];
      }
      hi = -h[3];
      lo = -OSQP_INFTY;
    }
    for (int k = 0; k < QP_ELL; ++k) {
      // explicit zeros are not structural non-zeros (sparseView() drops them, :264-265)
      if (col[k] >= 0 && val[k] == 0.0) col[k] = -1;
      pb.ecol[(size_t)r * QP_ELL + k] = col[k];
      pb.eval[(size_t)r * QP_ELL + k] = val[k];
    }
    pb.l[r] = lo;
    pb.u[r] = hi;
    pb.E[r] = 1.0;
    pb.z[r] = 0.0;
    pb.y[r] = 0.0;
  }
  for (int i = tid; i < M * 225; i += 256) s_P[i] = qc.QM[i % 225];
  for (int j = tid; j < n; j += 256) {
    s_D[j] = 1.0;
    s_x[j] = 0.0;
  }
  for (int j = tid; j <= n; j += 256) s_cnt[j] = 0;
  __syncthreads();

  // ---- 2. CSC index of A (row-sorted inside each column)
  for (int r = tid; r < m; r += 256)
    for (int k = 0; k < QP_ELL; ++k) {
      const int c = pb.ecol[(size_t)r * QP_ELL + k];
      if (c >= 0) atomicAdd(&s_cnt[c], 1);
    }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int j = 0; j < n; ++j) {
      pb.cptr[j] = acc;
      acc += s_cnt[j];
      s_cnt[j] = 0;
    }
    pb.cptr[n] = acc;
  }
  __syncthreads();
  for (int r = tid; r < m; r += 256)
    for (int k = 0; k < QP_ELL; ++k) {
      const int c = pb.ecol[(size_t)r * QP_ELL + k];
      if (c >= 0) {
        const int pos            = atomicAdd(&s_cnt[c], 1);
        pb.cidx[pb.cptr[c] + pos] = r * 8 + k;
      }
    }
  __syncthreads();
  for (int j = tid; j < n; j += 256) {  // insertion sort of each column's entries by row
    const int b = pb.cptr[j], e = pb.cptr[j + 1];
    for (int a = b + 1; a < e; ++a) {
      const int v = pb.cidx[a];
      int       q = a - 1;
      while (q >= b && pb.cidx[q] > v) {
        pb.cidx[q + 1] = pb.cidx[q];
        --q;
      }
      pb.cidx[q + 1] = v;
    }
  }
  __syncthreads();

  // ---- 3. Ruiz equilibration with cost scaling (OSQP scale_data)
  double c_scale = 1.0;
  for (int it = 0; it < qs.scaling_iters; ++it) {
    for (int j = tid; j < n; j += 256) {
      double        mx = 0;
      const double *Pb = s_P + (j / 15) * 225;
      const int     jj = j % 15;
      for (int i = 0; i < 15; ++i) mx = dmax(mx, dabs(Pb[i * 15 + jj]));
      for (int a = pb.cptr[j]; a < pb.cptr[j + 1]; ++a) {
        const int e = pb.cidx[a];
        mx          = dmax(mx, dabs(pb.eval[(size_t)(e >> 3) * QP_ELL + (e & 7)]));
      }
      s_Dt[j] = 1.0 / sogm_det::sqrt_rn(limit_scaling(mx));
    }
    __syncthreads();
    for (int r = tid; r < m; r += 256) {
      double mx = 0;
      for (int k = 0; k < QP_ELL; ++k)
        if (pb.ecol[(size_t)r * QP_ELL + k] >= 0) mx = dmax(mx, dabs(pb.eval[(size_t)r * QP_ELL + k]));
      const double et = 1.0 / sogm_det::sqrt_rn(limit_scaling(mx));
      for (int k = 0; k < QP_ELL; ++k) {
        const int c = pb.ecol[(size_t)r * QP_ELL + k];
        if (c >= 0) pb.eval[(size_t)r * QP_ELL + k] *= et * s_Dt[c];
      }
      pb.E[r] *= et;
    }
    for (int i = tid; i < M * 225; i += 256) {
      const int b = i / 225, rr = (i % 225) / 15, cc = i % 15;
      s_P[i] *= s_Dt[b * 15 + rr] * s_Dt[b * 15 + cc];
    }
    for (int j = tid; j < n; j += 256) s_D[j] *= s_Dt[j];
    __syncthreads();
    for (int j = tid; j < n; j += 256) {
      double        mx = 0;
      const double *Pb = s_P + (j / 15) * 225;
      const int     jj = j % 15;
      for (int i = 0; i < 15; ++i) mx = dmax(mx, dabs(Pb[i * 15 + jj]));
      s_cn[j] = mx;
    }
    __syncthreads();
    if (tid == 0) {
      double mean = 0;
      for (int j = 0; j < n; ++j) mean += s_cn[j];
      double c_temp = mean / n;
      double nq     = limit_scaling(0.0);  // q == 0
      c_temp        = dmax(c_temp, nq);
      c_temp        = limit_scaling(c_temp);
      s_sc[0]       = 1.0 / c_temp;
    }
    __syncthreads();
    const double ct = s_sc[0];
    for (int i = tid; i < M * 225; i += 256) s_P[i] *= ct;
    c_scale *= ct;
    __syncthreads();
  }
  for (int r = tid; r < m; r += 256) {
    pb.l[r] *= pb.E[r];
    pb.u[r] *= pb.E[r];
  }
  const double cinv = 1.0 / c_scale;
  __syncthreads();

  // ---- helpers as lambdas over the shared state -------------------------------------------------
  double rho_cur = qs.rho;
  auto set_rho = [&]() {
    for (int r = tid; r < m; r += 256) {
      const double lo = pb.l[r], hi = pb.u[r];
      double       v;
      if (lo < -OSQP_INFTY * MIN_SCALING && hi > OSQP_INFTY * MIN_SCALING)
        v = RHO_MIN;
      else if (hi - lo < RHO_TOL)
        v = RHO_EQ_OVER_RHO_INEQ * rho_cur;
      else
        v = rho_cur;
      pb.rho[r] = v;
    }
    __syncthreads();
  };
  // K band = P + sigma I + A^T diag(rho) A, then banded Cholesky in place (lower)
  auto factor = [&]() -> bool {
    for (int e = tid; e < n * (QP_BW + 1); e += 256) {
      const int i = e / (QP_BW + 1), dlt = e % (QP_BW + 1), j = i - dlt;
      double    s = 0.0;
      if (j >= 0) {
        if (i / 15 == j / 15) s = s_P[(i / 15) * 225 + (i % 15) * 15 + (j % 15)];
        if (i == j) s += qs.sigma;
        for (int a = pb.cptr[i]; a < pb.cptr[i + 1]; ++a) {
          const int    en = pb.cidx[a], r = en >> 3;
          const double vi = pb.eval[(size_t)r * QP_ELL + (en & 7)];
          for (int k = 0; k < QP_ELL; ++k)
            if (pb.ecol[(size_t)r * QP_ELL + k] == j) s += vi * pb.rho[r] * pb.eval[(size_t)r * QP_ELL + k];
        }
      }
      s_Kb[e] = s;
    }
    if (tid == 0) s_flag = 1;
    __syncthreads();
    if (wave == 0) {
      // right-looking banded Cholesky: column j is scaled, then the 17x17 trailing triangle is
      // updated by all 64 lanes (153 pair updates, <= 3 per lane); no reductions on the chain
      for (int j = 0; j < n; ++j) {
        const double djj = KB(s_Kb, j, j);
        if (!(djj > 0)) {
          if (lane == 0) s_flag = 0;
          break;
        }
        const double d = sogm_det::sqrt_rn(djj), inv = 1.0 / d;
        if (lane == 0) {
          KB(s_Kb, j, j) = d;
          s_ginv[j]      = inv;
        }
        if (lane >= 1 && lane <= QP_BW && j + lane < n) KB(s_Kb, j + lane, j) = KB(s_Kb, j + lane, j) * inv;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < (QP_BW * (QP_BW + 1)) / 2; e += 64) {
          // e -> (a, b) with 1 <= b <= a <= 17 (row-major lower triangle)
          int a = 1, rem = e;
          while (rem >= a) {
            rem -= a;
            ++a;
          }
          const int b = rem + 1;
          if (j + a < n) KB(s_Kb, j + a, j + b) -= KB(s_Kb, j + a, j) * KB(s_Kb, j + b, j);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    return s_flag != 0;
  };
  // solve K xt = xt in place (wave 0), banded forward / backward substitution
  auto solveK = [&]() {
    if (wave == 0) {
      for (int j = 0; j < n; ++j) {  // G y = b
        const double xj = s_xt[j] * s_ginv[j];
        if (lane == 0) s_xt[j] = xj;
        if (lane >= 1 && lane <= QP_BW && j + lane < n) s_xt[j + lane] -= KB(s_Kb, j + lane, j) * xj;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
      }
      for (int j = n - 1; j >= 0; --j) {  // G^T x = y
        const double xj = s_xt[j] * s_ginv[j];
        if (lane == 0) s_xt[j] = xj;
        if (lane >= 1 && lane <= QP_BW && j - lane >= 0) s_xt[j - lane] -= KB(s_Kb, j, j - lane) * xj;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
  };
  // residual norms (unscaled), results in s_sc: 0 pr, 1 nAx, 2 nz, 3 dr, 4 nPx, 5 nAty, 6 nq(=0)
  auto residuals = [&]() {
    double pr = 0, nAx = 0, nz = 0;
    for (int r = tid; r < m; r += 256) {
      double s = 0;
      for (int k = 0; k < QP_ELL; ++k) {
        const int c = pb.ecol[(size_t)r * QP_ELL + k];
        if (c >= 0) s += pb.eval[(size_t)r * QP_ELL + k] * s_x[c];
      }
      const double e = pb.E[r];
      pr             = dmax(pr, dabs((s - pb.z[r]) / e));
      nAx            = dmax(nAx, dabs(s / e));
      nz             = dmax(nz, dabs(pb.z[r] / e));
    }
    double dr = 0, nPx = 0, nAty = 0;
    for (int j = tid; j < n; j += 256) {
      double        s  = 0, a = 0;
      const double *Pb = s_P + (j / 15) * 225 + (j % 15) * 15;
      const int     b0 = (j / 15) * 15;
      for (int k = 0; k < 15; ++k) s += Pb[k] * s_x[b0 + k];
      for (int q = pb.cptr[j]; q < pb.cptr[j + 1]; ++q) {
        const int en = pb.cidx[q], r = en >> 3;
        a += pb.eval[(size_t)r * QP_ELL + (en & 7)] * pb.y[r];
      }
      const double dj = s_D[j];
      dr              = dmax(dr, dabs((s + a) / dj));
      nPx             = dmax(nPx, dabs(s / dj));
      nAty            = dmax(nAty, dabs(a / dj));
    }
    pr   = block_max(pr, s_red);
    nAx  = block_max(nAx, s_red);
    nz   = block_max(nz, s_red);
    dr   = block_max(dr, s_red);
    nPx  = block_max(nPx, s_red);
    nAty = block_max(nAty, s_red);
    __syncthreads();
    if (tid == 0) {
      s_sc[0] = pr;
      s_sc[1] = nAx;
      s_sc[2] = nz;
      s_sc[3] = dr * cinv;
      s_sc[4] = nPx;
      s_sc[5] = nAty;
      s_sc[6] = 0.0;
    }
    __syncthreads();
  };

  set_rho();
  bool chol_ok = factor();
  int  status = -2, iter = 0;
  if (!chol_ok) status = -7;

  // ---- 4. ADMM iterations
  const double alpha = qs.alpha;
  if (chol_ok) {
    for (iter = 1; iter <= qs.max_iter; ++iter) {
      for (int j = tid; j < n; j += 256) s_xp[j] = s_x[j];
      for (int r = tid; r < m; r += 256) pb.w[r] = pb.rho[r] * pb.z[r] - pb.y[r];
      __syncthreads();
      for (int j = tid; j < n; j += 256) {
        double s = qs.sigma * s_xp[j];  // q == 0
        for (int q = pb.cptr[j]; q < pb.cptr[j + 1]; ++q) {
          const int en = pb.cidx[q], r = en >> 3;
          s += pb.eval[(size_t)r * QP_ELL + (en & 7)] * pb.w[r];
        }
        s_xt[j] = s;
      }
      __syncthreads();
      solveK();
      for (int j = tid; j < n; j += 256) s_x[j] = alpha * s_xt[j] + (1.0 - alpha) * s_xp[j];
      for (int r = tid; r < m; r += 256) {
        double s = 0;
        for (int k = 0; k < QP_ELL; ++k) {
          const int c = pb.ecol[(size_t)r * QP_ELL + k];
          if (c >= 0) s += pb.eval[(size_t)r * QP_ELL + k] * s_xt[c];
        }
        const double rho = pb.rho[r], yr = pb.y[r];
        const double zr  = alpha * s + (1.0 - alpha) * pb.z[r];
        double       v   = zr + yr / rho;
        const double lo = pb.l[r], hi = pb.u[r];
        v               = v < lo ? lo : (v > hi ? hi : v);
        pb.z[r]         = v;
        const double d  = rho * (zr - v);
        pb.dy[r]        = d;
        pb.y[r]         = yr + d;
      }
      __syncthreads();
      const bool do_adapt = qs.adaptive_rho_interval > 0 && iter % qs.adaptive_rho_interval == 0;
      const bool do_check = qs.check_termination > 0 && iter % qs.check_termination == 0;
      if (do_adapt) {
        residuals();
        const double pr_n = s_sc[0] / (dmax(s_sc[1], s_sc[2]) + 1e-10);
        const double du_n =
            s_sc[3] / (dmax(dmax(cinv * s_sc[4], cinv * s_sc[5]), cinv * s_sc[6]) + 1e-10);
        double rho_new = rho_cur * sogm_det::sqrt_rn(pr_n / (du_n + 1e-10));
        rho_new        = rho_new < RHO_MIN ? RHO_MIN : (rho_new > 1e6 ? 1e6 : rho_new);
        if (rho_new > rho_cur * 5.0 || rho_new < rho_cur / 5.0) {
          rho_cur = rho_new;
          set_rho();
          if (!factor()) {
            status = -7;
            break;
          }
        }
      }
      if (do_check) {
        residuals();
        const double eps_prim = qs.eps_abs + qs.eps_rel * dmax(s_sc[1], s_sc[2]);
        const double eps_dual =
            qs.eps_abs + qs.eps_rel * cinv * dmax(dmax(s_sc[4], s_sc[5]), s_sc[6]);
        const bool p_ok = s_sc[0] < eps_prim, d_ok = s_sc[3] < eps_dual;
        if (p_ok && d_ok) {
          status = 1;
          break;
        }
        // primal infeasibility certificate (eps_prim_inf = 1e-4), as in the oracle
        const double eps_inf = 1e-4;
        double       ndy     = 0;
        for (int r = tid; r < m; r += 256) ndy = dmax(ndy, dabs(pb.E[r] * pb.dy[r]));
        ndy = block_max(ndy, s_red);
        __syncthreads();
        if (!p_ok && ndy > eps_inf) {
          // lhs = sum u max(d,0) + l min(d,0); +inf when an infinite bound is pushed
          double lhs = 0;
          int    bad = 0;
          for (int r = tid; r < m; r += 256) {
            const double d = pb.dy[r] / ndy;
            if (pb.u[r] < OSQP_INFTY * MIN_SCALING)
              lhs += pb.u[r] * (d > 0 ? d : 0);
            else if (d > eps_inf)
              bad = 1;
            if (pb.l[r] > -OSQP_INFTY * MIN_SCALING)
              lhs += pb.l[r] * (d < 0 ? d : 0);
            else if (d < -eps_inf)
              bad = 1;
          }
          lhs = wave_sum(lhs);
          __syncthreads();
          if (lane == 0) s_red[wave] = lhs;
          const int anybad = __syncthreads_or(bad);
          lhs              = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
          __syncthreads();
          if (!anybad && lhs < -eps_inf) {
            double na = 0;
            for (int j = tid; j < n; j += 256) {
              double s = 0;
              for (int q = pb.cptr[j]; q < pb.cptr[j + 1]; ++q) {
                const int en = pb.cidx[q], r = en >> 3;
                s += pb.eval[(size_t)r * QP_ELL + (en & 7)] * (pb.dy[r] / ndy);
              }
              na = dmax(na, dabs(s / s_D[j]));
            }
            na = block_max(na, s_red);
            __syncthreads();
            if (na < eps_inf) {
              status = -3;
              break;
            }
          }
        }
      }
    }
    if (iter > qs.max_iter) {
      iter = qs.max_iter;
      residuals();
      const double eps_prim = qs.eps_abs * 10 + qs.eps_rel * 10 * dmax(s_sc[1], s_sc[2]);
      const double eps_dual =
          qs.eps_abs * 10 + qs.eps_rel * 10 * cinv * dmax(dmax(s_sc[4], s_sc[5]), s_sc[6]);
      status = (s_sc[0] < eps_prim && s_sc[3] < eps_dual) ? 2 : -2;
    }
  }
  __syncthreads();
  double *out = out_cpts + (size_t)agent * SOGM_MAX_PIECES * 15;
  for (int j = tid; j < SOGM_MAX_PIECES * 15; j += 256) out[j] = j < n ? s_D[j] * s_x[j] : 0.0;
  if (tid == 0) {
    out_status[agent] = status;
    out_iters[agent]  = iter;
  }
}

int launch_qp(const SogmPlannerParams &pp, const SogmQpSettings &qs, const QpWorkspace &ws,
              const QpConst &qc, int n_agents, const double *start_pva, const double *goal_pv,
              const double *polys, const int32_t *nfaces, const int32_t *npoly, double *out_cpts,
              int32_t *out_status, int32_t *out_iters, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)k_qp, hipFuncAttributeMaxDynamicSharedMemorySize,
                              ws.dyn_lds_bytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_qp, dim3(n_agents), dim3(256), ws.dyn_lds_bytes, st, pp, qs, ws, qc, start_pva,
                     goal_pv, polys, nfaces, npoly, out_cpts, out_status, out_iters);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace sogm
