// sogm_qp.hip — batched corridor-constrained min-jerk Bezier QP, OSQP-algorithm ADMM, gfx950.
//
// Reference: traj_opt::BezierOpt::setup / optimize (traj_opt/src/bezier_optimizer.cpp:27-285)
// through IOSQP (traj_opt/include/iosqp.hpp:40-115) into OSQP (external, v0.6 API, not vendored).
// The solver restates the published OSQP algorithm exactly as oracle/qp_oracle.cpp does.
//
// Mapping to CDNA4: one workgroup (512 lanes = 8 waves, two per SIMD) per agent.  The problems are tiny
// (n = 15 M <= 240 variables, m ~ 5e2..5.6e3 rows) and strictly latency-bound — an ADMM chain of
// a few hundred dependent iterations — so the design goal is that one iteration never leaves the
// CU and has a short critical path:
//   * rows are split by structure.  "General" rows (continuity, velocity / acceleration boxes:
//     9(M+1) + 21M rows, <= 6 non-zeros) are kept in ELL form plus a row-sorted CSC index for
//     A^T products.  "Safety" rows (5 per corridor face: h . c_k <= -h3 for each control point k)
//     dominate m and have implicit structure — row (piece i, face j, point k) touches exactly the
//     3 coordinates of point k — so they store only 3 coefficients and are addressed directly:
//     no column indices, no CSC, and A^T w for a column is a stride-5 walk over that piece's faces;
//   * P is block diagonal (one 15x15 min-jerk block per piece), K = P + sigma I + A^T diag(rho) A is
//     banded (half bandwidth 17): an n x 18 band in LDS, banded Cholesky on one wave
//     (right-looking, no reductions on the dependency chain), re-factored only when rho changes;
//   * register-resident iteration (M <= 8, <= 256 general and <= 1024 safety rows — every problem of the
//     bench workload): after each factorisation the rows of K^-1 = (G G^T)^-1 are built from the
//     block-bidiagonal factor (LDS-staged 15x15 block products) and kept in registers, 30 entries per
//     lane, four lanes per row; x~ = K^-1 rhs is one register mat-vec with two DPP exchanges.  Waves 0-3
//     own one general row per lane, waves 4-7 up to four safety rows per lane: coefficients, bounds, z
//     and y live in registers, a row update is one batch of independent x~ loads plus FMAs, and only
//     w = rho z - y (what A^T w reads) goes back to LDS.  Column entries of A are register-resident too;
//   * general iteration (anything larger): row data in LDS when it fits, else in a per-agent HBM scratch
//     (same code through flat pointers), banded substitution on wave 0;
//   * an iteration is 3 barriers: [A^T w per column] | [solve] | [per row z~ = A x~, projection, dual
//     update, x]; residual norms (every 25 iterations) are wavefront shuffle reductions combined through
//     LDS; the termination test runs before the rho update on one residual evaluation (osqp.c order).
// DESIGN.md section 3.1 has the measured phase split and the compiler notes (opaque lane ids, separate
// template instances, loop constants kept out of the kernel-argument SGPR tuple).
// No dense contraction anywhere -> no MFMA.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/sogm_detmath.h"
#include "sogm_planner.hpp"

#include <type_traits>

namespace sogm {
namespace {

#define QP_BW 17  // half bandwidth of K
#define QP_NMAX (15 * SOGM_MAX_PIECES)
#define QP_ELL 6
#define QP_NT 512  // lanes per workgroup (8 waves: two per SIMD)
#define QP_K1_DOUBLES (120 * (QP_BW + 1))  // K1 band of a register-resident problem (M <= 8)

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
// wave-wide max / sum of a per-lane value without the LDS crossbar: two quad permutes, row_half_mirror, row_mirror
// (after which every lane of a 16-lane row holds the row's result), then the four rows through readlane.  All 64
// lanes must be active.  Maxima of non-negative values (v_max_f64 ignores a NaN operand like the oracle's std::max).
template <int CTRL>
__device__ inline double dpp_ctrl_t(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo     = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi     = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ inline double read_lane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                          __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ inline double wave_max_dpp(double v) {
  v = __builtin_fmax(v, dpp_ctrl_t<0xB1>(v));
  v = __builtin_fmax(v, dpp_ctrl_t<0x4E>(v));
  v = __builtin_fmax(v, dpp_ctrl_t<0x141>(v));
  v = __builtin_fmax(v, dpp_ctrl_t<0x140>(v));
  return __builtin_fmax(__builtin_fmax(read_lane(v, 0), read_lane(v, 16)),
                        __builtin_fmax(read_lane(v, 32), read_lane(v, 48)));
}
// the same for fp32 values (one DPP move per step instead of two)
template <int CTRL>
__device__ inline float dpp_ctrl_f(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ inline float wave_max_dpp_f(float v) {
  v = __builtin_fmaxf(v, dpp_ctrl_f<0xB1>(v));
  v = __builtin_fmaxf(v, dpp_ctrl_f<0x4E>(v));
  v = __builtin_fmaxf(v, dpp_ctrl_f<0x141>(v));
  v = __builtin_fmaxf(v, dpp_ctrl_f<0x140>(v));
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return __builtin_fmaxf(__builtin_fmaxf(a, b), __builtin_fmaxf(c, d));
}
__device__ inline double wave_sum_dpp(double v) {
  v += dpp_ctrl_t<0xB1>(v);
  v += dpp_ctrl_t<0x4E>(v);
  v += dpp_ctrl_t<0x141>(v);
  v += dpp_ctrl_t<0x140>(v);
  return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}
__device__ inline double block_max(double v, double *s_red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return dmax(dmax(dmax(s_red[0], s_red[1]), dmax(s_red[2], s_red[3])),
              dmax(dmax(s_red[4], s_red[5]), dmax(s_red[6], s_red[7])));
}

// Row storage (LDS or HBM scratch).  The "hot" arrays are the only row data the register-resident ADMM
// iteration touches; they always live in LDS on that path, the "cold" rest may sit in HBM scratch.
struct Rows {
  // general rows [0, G)
  double *gval;  // [G][6]
  double *gl, *gu, *grho, *gE, *gz, *gy, *gdy;
  double *grinv;  // [G] 1 / rho (OSQP rho_inv_vec)
  double *gw;     // [G] rho z - y                                                    (hot)
  int    *gcol;  // [G][6]
  int    *cidx;  // CSC entries over general rows: row * 8 + slot, row-sorted inside a column
  // safety rows [0, S):  s = off_i + 5 * face + k ; columns i*15 + k*3 + {0,1,2}
  double *sval;  // [S][3]                                                             (hot)
  double *su, *sE, *sz, *sy, *sdy;
  double *sw;   // [S] rho z - y (refreshed by the row update, consumed by the next A^T product)   (hot)
  int    *sc0;  // [S] first column of the row's control point
};

__host__ __device__ inline size_t rows_hot_bytes(int G, int S) { return ((size_t)S * 4 + G) * 8; }
__host__ __device__ inline size_t rows_cold_bytes(int G, int S) {
  return (size_t)G * (QP_ELL * 8 + 8 * 8 + QP_ELL * 4 + QP_ELL * 4) + (size_t)S * (5 * 8 + 4) + 64;
}
__host__ __device__ inline size_t rows_bytes(int G, int S) { return rows_hot_bytes(G, S) + rows_cold_bytes(G, S); }
__device__ inline void carve_rows(Rows &R, char *hot, char *cold, int G, int S) {
  double *h = (double *)hot;
  R.sval = h;  h += (size_t)S * 3;
  R.sw = h;    h += S;
  R.gw = h;
  double *d = (double *)cold;
  R.gval = d;  d += (size_t)G * QP_ELL;
  R.gl = d;    d += G;
  R.gu = d;    d += G;
  R.grho = d;  d += G;
  R.gE = d;    d += G;
  R.gz = d;    d += G;
  R.gy = d;    d += G;
  R.gdy = d;   d += G;
  R.grinv = d; d += G;
  R.su = d;    d += S;
  R.sE = d;    d += S;
  R.sz = d;    d += S;
  R.sy = d;    d += S;
  R.sdy = d;   d += S;
  int *q = (int *)d;
  R.gcol = q;  q += (size_t)G * QP_ELL;
  R.sc0 = q;   q += S;
  R.cidx = q;
}

// value of the lane selected by a DPP quad permute (0xB1: lane ^ 1, 0x4E: lane ^ 2)
template <int CTRL>
__device__ inline double dpp_quad_t(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo     = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi     = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
#define dpp_quad(v, ctrl) dpp_quad_t<ctrl>(v)

// Opaque copy of a lane index: addresses derived from it inside a loop are recomputed there (one VALU add)
// instead of being hoisted into dozens of long-lived address registers that spill to scratch.
__device__ inline int launder(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

__device__ inline double &KB(double *Kb, int i, int j) { return Kb[i * (QP_BW + 1) + (i - j)]; }

}  // namespace

size_t qp_k1_scratch_bytes_per_agent() { return (size_t)QP_K1_DOUBLES * sizeof(double); }
size_t qp_scratch_bytes_per_agent(int max_faces) {
  const int G = 9 * (SOGM_MAX_PIECES + 1) + 21 * SOGM_MAX_PIECES;
  const int S = 5 * max_faces * SOGM_MAX_PIECES;
  return (rows_bytes(G, S) + 255) & ~(size_t)255;
}

// One agent's QP, executed by the whole workgroup (QP_NT lanes); called by k_qp (one workgroup per agent) and by
// the dataflow kernel k_qp_flow (a workgroup solves agents one after the other as their corridors become final).
__device__ __forceinline__ void qp_solve_agent(const SogmPlannerParams &pp, const SogmQpSettings &qs,
                                               const QpWorkspace &ws, const QpConst &qc, const double *start_pva,
                                               const double *goal_pv, const double *polys, const int32_t *nfaces,
                                               const int32_t *npoly, double *out_cpts, int32_t *out_status,
                                               int32_t *out_iters, int ablate_arg, int agent) {
  // phase ablation is a profiling aid: compiled in only with -DSOGM_QP_ABLATE_BUILD (tools/qp_ablate.py)
#ifdef SOGM_QP_ABLATE_BUILD
  const int ablate = ablate_arg;
#else
  constexpr int ablate = 0;
  (void)ablate_arg;
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long dbg_t0 = wall_clock64(), dbg_clk0 = clock64();
  long long       dbg_refac = 0, dbg_check = 0, dbg_f1 = 0, dbg_f2 = 0, dbg_f3 = 0, dbg_k1 = 0, dbg_k2 = 0;
  int             dbg_nrefac = 0, dbg_ncheck = 0;
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
  // Dynamic LDS, general path: s_Kb [n][18] K band / Cholesky factor | s_P [M][225] scaled cost blocks | rows.
  // Register-resident path (M <= 8): K is block tridiagonal in 15 x 15 piece blocks (the profile of the continuity
  // rows) and its inverse is built block by block (factor()):
  //   s_P [M][225] | s_Dv [M][225] diagonal blocks of K, then inv(D_i) | s_V [M][225] K(i, i-1), then V_i |
  //   s_Z [M][225] one block row of K^-1 | hot rows | cold rows | K1 [n][18] the band of A^T diag(rho / rho_cur) A
  //   (the cold rows and K1 sit in LDS when everything fits, in the agent's HBM scratch otherwise)
  const bool use_blocks = M <= 8;
  __shared__ double s_ginv[QP_NMAX];
  __shared__ __attribute__((aligned(16))) double s_xt[QP_NMAX];
  __shared__ double s_x[QP_NMAX], s_D[QP_NMAX], s_Dt[QP_NMAX];
  __shared__ __attribute__((aligned(16))) double s_cn[QP_NMAX];
  __shared__ double s_red[8];
  __shared__ double s_T[2][225];  // factor(): running 15 x 15 block of the X row chains
  __shared__ double s_red12[96];
  __shared__ double s_sc[16];
  __shared__ int    s_off[SOGM_MAX_PIECES + 1];  // safety-row offset of each piece
  __shared__ int    s_nf[SOGM_MAX_PIECES];
  __shared__ int    s_cnt[QP_NMAX + 1];
  __shared__ int    s_cptr[QP_NMAX + 1];
  __shared__ int    s_flag;

  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < M; ++i) {
      const int f = nfaces[agent * SOGM_MAX_PIECES + i];
      s_off[i]    = acc;
      s_nf[i]     = f;
      acc += 5 * f;
    }
    s_off[M] = acc;
  }
  __syncthreads();
  const int R1 = 3 * (M + 1), R3 = 9 * (M + 1), R4 = R3 + 12 * M, G = R4 + 9 * M;
  const int S = s_off[M];
  // Register-resident iteration (FAST): K^-1 rows, one general row or up to four safety rows per lane in
  // registers; needs the block form of the factor (M <= 8) and only the hot row arrays in LDS.
  const size_t head1 = ((size_t)n * (QP_BW + 1) + (size_t)M * 225) * sizeof(double);
  const size_t head4 = (size_t)M * 225 * 4 * sizeof(double);
  const size_t k1_bytes = (size_t)n * (QP_BW + 1) * sizeof(double);
  // (the check of the register-resident path stages y through factor()'s block area: it must hold one double per row and 320 partial results)
  const bool   fast  = use_blocks && G <= 256 && S <= 1024 && (size_t)(G + S) + 320 <= (size_t)M * 675 &&
                    head4 + rows_hot_bytes(G, S) <= (size_t)ws.dyn_lds_bytes;
  const size_t head        = fast ? head4 : head1;
  const size_t rows_al     = (rows_bytes(G, S) + 15) & ~(size_t)15;
  const bool   rows_in_lds = head + rows_al + (fast ? k1_bytes : 0) <= (size_t)ws.dyn_lds_bytes;
  double *s_Kb = (double *)qp_smem;
  double *s_P  = fast ? (double *)qp_smem : s_Kb + (size_t)n * (QP_BW + 1);
  double *s_Dv = s_P + (size_t)M * 225, *s_V = s_Dv + (size_t)M * 225, *s_Z = s_V + (size_t)M * 225;  // fast only
  // (read through a generic pointer: once per entry and factorisation)
  const double *k1r = rows_in_lds ? (const double *)(qp_smem + head + rows_al)
                                  : (const double *)(ws.k1_scratch + (size_t)agent * QP_K1_DOUBLES);
  double       *k1w = const_cast<double *>(k1r);
  // The solver body is instantiated twice (forced inline): once with every row pointer derived from
  // the LDS array — so the compiler emits ds_read/ds_write instead of flat accesses — and once for
  // the HBM-scratch fallback.
  auto body = [&](const Rows &R, auto fast_tag) __attribute__((always_inline)) {
  constexpr bool FAST = decltype(fast_tag)::value;  // compile-time: the two iterations never share a loop
  const double *sp   = start_pva + agent * 9;
  const int     gstr = ws.goal_stride == 9 ? 9 : 6;
  const double *gp   = goal_pv + agent * gstr;
  // time allocation t_[i] (bezier_optimizer.cpp:135,164-165,191-192,220,231): replan() gives every piece
  // corridor_tau (baseline.cpp:411); the general entry passes its own vector
  const double *tal  = ws.t_alloc ? ws.t_alloc + (size_t)agent * SOGM_MAX_PIECES : nullptr;
  auto          T    = [&](int i) -> double { return tal ? tal[i] : pp.corridor_tau; };
  const double  vmax = pp.opt_max_vel, amax = pp.opt_max_acc;

  // ---- 1. assembly (bezier_optimizer.cpp:113-260): general rows in ELL, one lane per row
  for (int r = tid; r < G; r += QP_NT) {
    int    col[QP_ELL];
    double val[QP_ELL];
    for (int k = 0; k < QP_ELL; ++k) {
      col[k] = -1;
      val[k] = 0.0;
    }
    double       lo = 0.0, hi = 0.0;
    const double p2a[3] = {12, -24, 12};
    if (r < R3) {
      const int kind = r / R1;  // 0 pos, 1 vel, 2 acc
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
          lo = hi = sp[3 + d] * T(0);
        } else if (gidx == M) {
          col[0] = M * 15 - 6 + d;
          val[0] = -4;
          col[1] = M * 15 - 3 + d;
          val[1] = 4;
          lo = hi = gp[3 + d] * T(M - 1);
        } else {
          col[0] = gidx * 15 + d;
          const double t1 = T(gidx), t1_ = T(gidx - 1);
          val[0] = -4.0 / t1;
          col[1] = gidx * 15 + 3 + d;
          val[1] = 4.0 / t1;
          col[2] = gidx * 15 - 3 + d;
          val[2] = -4.0 / t1_;
          col[3] = gidx * 15 - 6 + d;
          val[3] = 4.0 / t1_;
        }
      } else {
        if (gidx == 0) {
          for (int k = 0; k < 3; ++k) {
            col[k] = k * 3 + d;
            val[k] = p2a[k];
          }
          lo = hi = sp[6 + d] * T(0) * T(0);
        } else if (gidx == M) {
          for (int k = 0; k < 3; ++k) {
            col[k] = M * 15 - 9 + k * 3 + d;
            val[k] = p2a[k];
          }
          // final acceleration: 0 for replan() (baseline.cpp:423), the caller's for the general entry
          lo = hi = (gstr == 9 ? gp[6 + d] : 0.0) * T(M - 1) * T(M - 1);
        } else {
          const double t2 = T(gidx) * T(gidx), t2_ = T(gidx - 1) * T(gidx - 1);  // pow(t_[i], 2) (:191-192)
          for (int k = 0; k < 3; ++k) {
            col[k]     = gidx * 15 + k * 3 + d;
            val[k]     = p2a[k] / t2;
            col[3 + k] = gidx * 15 - 9 + k * 3 + d;
            val[3 + k] = -p2a[k] / t2_;
          }
        }
      }
    } else if (r < R4) {
      const int rr = r - R3, i = rr / 12, j = (rr % 12) / 3, d = rr % 3;
      col[0] = i * 15 + j * 3 + d;
      val[0] = -4;
      col[1] = i * 15 + j * 3 + 3 + d;
      val[1] = 4;
      hi     = vmax * 1.0 * T(i);
      lo     = -vmax * 1.0 * T(i);
    } else {
      const int rr = r - R4, i = rr / 9, j = (rr % 9) / 3, d = rr % 3;
      for (int k = 0; k < 3; ++k) {
        col[k] = i * 15 + j * 3 + k * 3 + d;
        val[k] = p2a[k];
      }
      hi = amax * 1.0 * T(i) * T(i);
      lo = -amax * 1.0 * T(i) * T(i);
    }
    for (int k = 0; k < QP_ELL; ++k) {
      R.gcol[(size_t)r * QP_ELL + k] = col[k];
      R.gval[(size_t)r * QP_ELL + k] = val[k];
    }
    R.gl[r] = lo;
    R.gu[r] = hi;
    R.gE[r] = 1.0;
    R.gz[r] = 0.0;
    R.gy[r] = 0.0;
  }
  // safety rows: one lane per row (explicit zero coefficients simply stay zero)
  for (int s = tid; s < S; s += QP_NT) {
    int i = 0;
    while (i + 1 < M && s >= s_off[i + 1]) ++i;
    const int     q = s - s_off[i], face = q / 5;
    const double *h = polys + (((size_t)agent * SOGM_MAX_PIECES + i) * MF + face) * 4;
    R.sval[(size_t)s * 3 + 0] = h[0];
    R.sval[(size_t)s * 3 + 1] = h[1];
    R.sval[(size_t)s * 3 + 2] = h[2];
    R.su[s] = -h[3];
    R.sE[s] = 1.0;
    R.sz[s] = 0.0;
    R.sy[s] = 0.0;
    R.sw[s] = 0.0;
    R.sc0[s] = i * 15 + (q % 5) * 3;
  }
  for (int i = tid; i < M * 225; i += QP_NT) s_P[i] = qc.QM[i % 225];
  for (int j = tid; j < n; j += QP_NT) {
    s_D[j] = 1.0;
    s_x[j] = 0.0;
  }
  for (int j = tid; j <= n; j += QP_NT) s_cnt[j] = 0;
  __syncthreads();

  // ---- 2. CSC index of the general rows (row-sorted inside each column)
  for (int r = tid; r < G; r += QP_NT)
    for (int k = 0; k < QP_ELL; ++k) {
      const int c = R.gcol[(size_t)r * QP_ELL + k];
      if (c >= 0) atomicAdd(&s_cnt[c], 1);
    }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int j = 0; j < n; ++j) {
      s_cptr[j] = acc;
      acc += s_cnt[j];
      s_cnt[j] = 0;
    }
    s_cptr[n] = acc;
  }
  __syncthreads();
  for (int r = tid; r < G; r += QP_NT)
    for (int k = 0; k < QP_ELL; ++k) {
      const int c = R.gcol[(size_t)r * QP_ELL + k];
      if (c >= 0) {
        const int pos           = atomicAdd(&s_cnt[c], 1);
        R.cidx[s_cptr[c] + pos] = r * 8 + k;
      }
    }
  __syncthreads();
  for (int j = tid; j < n; j += QP_NT) {
    const int b = s_cptr[j], e = s_cptr[j + 1];
    for (int a = b + 1; a < e; ++a) {
      const int v = R.cidx[a];
      int       q = a - 1;
      while (q >= b && R.cidx[q] > v) {
        R.cidx[q + 1] = R.cidx[q];
        --q;
      }
      R.cidx[q + 1] = v;
    }
  }
  __syncthreads();

  // column j = (piece pi, point pk, dim pd): its safety rows are s_off[pi] + 5 * face + pk
#define COL_DECODE(j)                                                   \
  const int pi = (j) / 15, pk = ((j) % 15) / 3, pd = (j) % 3;           \
  const int sbase = s_off[pi] + pk, nface = s_nf[pi];                   \
  (void)pd;

  const long long dbg_s1 = wall_clock64() - dbg_t0;
  // ---- 3. Ruiz equilibration with cost scaling (OSQP scale_data)
  double c_scale = 1.0;
  // Restated for the device without changing one rounding:
  //  * maxima of |.| with the operand modifier and v_max_f64 (finite values); a maximum does not depend on the order of
  //    its candidates, so a column's entries are split over a lane quad and meet through two DPP steps;
  //  * the cost scaling's P <- c P is applied by the NEXT pass's P <- (c P) (d_r d_c) — the same two products per
  //    entry in the same order (a last sweep after the loop) — and the P part of the next pass's column norm is
  //    fl(c * cn_j) without reading P again: x -> fl(c x) is monotone for c > 0, so the largest entry stays the largest;
  //  * the quad that owns column j scales P's column j and takes its norm in one phase.
  // Three barriers per pass instead of five; every product, square root and division is one OSQP's scale_data performs.
  auto p_colmax = [&](int j, int q) __attribute__((always_inline)) -> double {
    const double *Pb = s_P + (j / 15) * 225 + j % 15;
    double        mx = 0;
    for (int i = q; i < 15; i += 4) mx = __builtin_fmax(mx, __builtin_fabs(Pb[i * 15]));
    mx = __builtin_fmax(mx, dpp_quad(mx, 0xB1));
    return __builtin_fmax(mx, dpp_quad(mx, 0x4E));
  };
  for (int jq = launder(tid); jq < 4 * n; jq += QP_NT) {  // (a quad is in or out of a trip as a whole)
    const double mx = p_colmax(jq >> 2, jq & 3);
    if ((jq & 3) == 0) s_cn[jq >> 2] = mx;
  }
  double ct_prev = 1.0;  // the previous pass's c, not yet applied to s_P (x 1.0 is exact)
  __syncthreads();
  for (int it = 0; it < qs.scaling_iters; ++it) {
    for (int jq = launder(tid); jq < 4 * n; jq += QP_NT) {
      const int j = jq >> 2, q = jq & 3;
      double    mx = q == 0 ? s_cn[j] * ct_prev : 0.0;  // max |c P(:, j)|
      for (int a = s_cptr[j] + q; a < s_cptr[j + 1]; a += 4) {
        const int e = R.cidx[a];
        mx          = __builtin_fmax(mx, __builtin_fabs(R.gval[(size_t)(e >> 3) * QP_ELL + (e & 7)]));
      }
      COL_DECODE(j)
      for (int f = q; f < nface; f += 4)
        mx = __builtin_fmax(mx, __builtin_fabs(R.sval[(size_t)(sbase + 5 * f) * 3 + pd]));
      mx = __builtin_fmax(mx, dpp_quad(mx, 0xB1));
      mx = __builtin_fmax(mx, dpp_quad(mx, 0x4E));
      if (q == 0) s_Dt[j] = 1.0 / sogm_det::sqrt_rn(limit_scaling(mx));
    }
    __syncthreads();
    for (int r = tid; r < G; r += QP_NT) {
      double mx = 0;
      for (int k = 0; k < QP_ELL; ++k)
        if (R.gcol[(size_t)r * QP_ELL + k] >= 0) mx = __builtin_fmax(mx, __builtin_fabs(R.gval[(size_t)r * QP_ELL + k]));
      const double et = 1.0 / sogm_det::sqrt_rn(limit_scaling(mx));
      for (int k = 0; k < QP_ELL; ++k) {
        const int c = R.gcol[(size_t)r * QP_ELL + k];
        if (c >= 0) R.gval[(size_t)r * QP_ELL + k] *= et * s_Dt[c];
      }
      R.gE[r] *= et;
    }
    for (int s = tid; s < S; s += QP_NT) {
      double      *v  = R.sval + (size_t)s * 3;
      const double mx = __builtin_fmax(__builtin_fmax(__builtin_fabs(v[0]), __builtin_fabs(v[1])), __builtin_fabs(v[2]));
      const double et = 1.0 / sogm_det::sqrt_rn(limit_scaling(mx));
      const int    c0 = R.sc0[s];  // piece * 15 + point * 3 (assembled once: not searched for in every pass)
      v[0] *= et * s_Dt[c0];
      v[1] *= et * s_Dt[c0 + 1];
      v[2] *= et * s_Dt[c0 + 2];
      R.sE[s] *= et;
    }
    for (int jq = launder(tid); jq < 4 * n; jq += QP_NT) {  // P(:, j) <- (c_prev P(:, j)) (d_r d_j), and its norm
      const int     j = jq >> 2, q = jq & 3, b15 = (j / 15) * 15;
      double       *Pb = s_P + (j / 15) * 225 + j % 15;
      const double  dj = s_Dt[j];
      double        mx = 0;
      for (int i = q; i < 15; i += 4) {
        const double v = (Pb[i * 15] * ct_prev) * (s_Dt[b15 + i] * dj);
        Pb[i * 15]     = v;
        mx             = __builtin_fmax(mx, __builtin_fabs(v));
      }
      mx = __builtin_fmax(mx, dpp_quad(mx, 0xB1));
      mx = __builtin_fmax(mx, dpp_quad(mx, 0x4E));
      if (q == 0) {
        s_cn[j] = mx;
        s_D[j] *= dj;
      }
    }
    __syncthreads();
    if (tid == 0) {
      double mean = 0;
      for (int j = 0; j < n; ++j) mean += s_cn[j];
      double c_temp = mean / n;
      c_temp        = dmax(c_temp, limit_scaling(0.0));  // ||q||inf = 0 -> 1.0
      c_temp        = limit_scaling(c_temp);
      s_sc[0]       = 1.0 / c_temp;
    }
    __syncthreads();
    ct_prev = s_sc[0];
    c_scale *= ct_prev;
  }
  for (int i = tid; i < M * 225; i += QP_NT) s_P[i] *= ct_prev;  // the last pass's c
  for (int r = tid; r < G; r += QP_NT) {
    R.gl[r] *= R.gE[r];
    R.gu[r] *= R.gE[r];
  }
  for (int s = tid; s < S; s += QP_NT) R.su[s] *= R.sE[s];
  const double cinv = 1.0 / c_scale;
  __syncthreads();
  const long long dbg_s2 = wall_clock64() - dbg_t0;

  // ---- helpers -----------------------------------------------------------------------------------
  double rho_cur = qs.rho;  // safety rows are one-sided inequalities: their rho is rho_cur itself
  double rinv_cur = 1.0 / rho_cur;
  // Row rho of K^-1 = X^T X (X = G^-1, block lower triangular), columns 30 q .. 30 q + 29 (zero beyond n),
  // rebuilt after every factorisation: the per-iteration solve is ONE register mat-vec and one barrier.
  // Four adjacent lanes share a row (30 doubles = 60 VGPRs each); the partial dot products meet through two
  // DPP exchanges.
  double    kinv[30] = {};
  auto   set_rho = [&]() {
    // lane ids re-derived from an opaque copy: nothing in here is hoisted out of the ADMM loop
    const int tid = launder((int)threadIdx.x), lane = tid & 63, wave = tid >> 6;
    (void)lane;
    (void)wave;
    for (int r = tid; r < G; r += QP_NT) {
      const double lo = R.gl[r], hi = R.gu[r];
      double       v;
      if (lo < -OSQP_INFTY * MIN_SCALING && hi > OSQP_INFTY * MIN_SCALING)
        v = RHO_MIN;
      else if (hi - lo < RHO_TOL)
        v = RHO_EQ_OVER_RHO_INEQ * rho_cur;
      else
        v = rho_cur;
      R.grho[r]  = v;
      R.grinv[r] = 1.0 / v;
    }
    for (int s = tid; s < S; s += QP_NT) R.sw[s] = rho_cur * R.sz[s] - R.sy[s];
    __syncthreads();
  };
  // Register-resident path: every row's rho is rho_cur times a constant (1000 for an equality row, 1 otherwise: the
  // general rows of this QP all have two finite bounds, the safety rows one), so K = P + sigma I + rho_cur K1 with
  // K1 = A^T diag(rho / rho_cur) A fixed after the scaling: its band is built once and a refactorisation starts from
  // three LDS reads per entry instead of the walk over the column's rows.
  auto build_K1 = [&]() {
    const int tid = launder((int)threadIdx.x);
    for (int e = tid; e < n * (QP_BW + 1); e += QP_NT) {
      const int i = e / (QP_BW + 1), dlt = e % (QP_BW + 1), j = i - dlt;
      double    s = 0.0;
      if (j >= 0) {
        for (int a = s_cptr[i]; a < s_cptr[i + 1]; ++a) {
          const int    en = R.cidx[a], r = en >> 3;
          const double vi = R.gval[(size_t)r * QP_ELL + (en & 7)];
          const double mr = R.gu[r] - R.gl[r] < RHO_TOL ? RHO_EQ_OVER_RHO_INEQ : 1.0;
          for (int k = 0; k < QP_ELL; ++k)
            if (R.gcol[(size_t)r * QP_ELL + k] == j) s += vi * mr * R.gval[(size_t)r * QP_ELL + k];
        }
        if (i / 3 == j / 3) {  // same control point: safety rows couple its 3 coordinates
          COL_DECODE(i)
          const int jd = j % 3;
          for (int f = 0; f < nface; ++f) {
            const double *v = R.sval + (size_t)(sbase + 5 * f) * 3;
            s += v[pd] * v[jd];
          }
        }
      }
      k1w[e] = s;
    }
    __threadfence_block();
    __syncthreads();
  };
  // K band = P + sigma I + A^T diag(rho) A (rows in global row order: general, then safety),
  // then banded Cholesky in place
  auto factor = [&]() __attribute__((always_inline)) -> bool {
    // lane ids re-derived from an opaque copy: nothing in here is hoisted out of the ADMM loop
    const int tid = launder((int)threadIdx.x), lane = tid & 63, wave = tid >> 6;
    (void)lane;
    (void)wave;
    if constexpr (FAST) {
      const long long dbg_fa = wall_clock64();
      // ---- K^-1 of the block tridiagonal K, block by block (no scalar Cholesky chain, no X^T X product) ------------
      // 1. the blocks of K: diagonal D_b = K(b, b) into s_Dv, sub-diagonal O_b = K(b, b-1) into s_V.  An entry is
      //    evaluated with its larger index first, so (i, j) and (j, i) are the same sum: the blocks are symmetric
      //    bit for bit.
      for (int e = tid; e < 2 * M * 225; e += QP_NT) {
        const int which = e >= M * 225 ? 1 : 0, ee = e - which * M * 225;
        const int b = ee / 225, r = (ee % 225) / 15, c = ee % 15;
        int       i = b * 15 + r, j = (b - which) * 15 + c;
        double    s = 0.0;
        if (j >= 0) {
          if (i < j) {
            const int t_ = i;
            i            = j;
            j            = t_;
          }
          if (i / 15 == j / 15) s = s_P[(i / 15) * 225 + (i % 15) * 15 + (j % 15)];
          if (i == j) s += qs.sigma;
          if (i - j <= QP_BW) s = __builtin_fma(rho_cur, k1r[i * (QP_BW + 1) + (i - j)], s);
        }
        (which ? s_V : s_Dv)[ee] = s;
      }
      if (tid == 0) s_flag = 1;
      __syncthreads();
      const long long dbg_fb = wall_clock64();
      dbg_f1 += dbg_fb - dbg_fa;
      // 2. forward sweep: S_0 = D_0, S_i = D_i - O_i inv(S_{i-1}) O_i^T; s_Dv block i <- inv(S_i), s_V block i <-
      //    V_i = O_i inv(S_{i-1}).  The inverse of a 15 x 15 SPD block is fifteen symmetric sweeps (Gauss-Jordan
      //    without pivoting: a_pp <- -1/d, a_rp <- a_rp/d, a_rc <- a_rc - a_rp a_pc/d; all pivots swept: -inv(A)), one
      //    lane per entry, the entry in a register, the block ping-ponged between two LDS copies: one barrier per sweep.
      const int  er = (tid % 225) / 15, ec = tid % 15;  // this lane's entry of a 15 x 15 block (lanes < 225)
      const bool el = tid < 225;
      const int  hi_ = er > ec ? er : ec, lo_ = er > ec ? ec : er;
      for (int i = 0; i < M; ++i) {  // uniform
        double a = 0.0;
        if (i > 0) {
          // T = O_i inv(S_{i-1})
          if (el) {
            const double *Or = s_V + i * 225 + er * 15, *Dc = s_Dv + (i - 1) * 225 + ec;
            double        t  = 0.0;
#pragma unroll
            for (int k = 0; k < 15; ++k) t = __builtin_fma(Or[k], Dc[k * 15], t);
            s_T[0][tid] = t;
          }
          __syncthreads();
          // S_i = D_i - T O_i^T, evaluated for (max, min) on both sides of the diagonal: symmetric bit for bit
          if (el) {
            const double *Tr = s_T[0] + hi_ * 15, *Oc = s_V + i * 225 + lo_ * 15;
            double        t  = s_Dv[i * 225 + hi_ * 15 + lo_];
#pragma unroll
            for (int k = 0; k < 15; ++k) t = __builtin_fma(-Tr[k], Oc[k], t);
            a = t;
          }
          __syncthreads();  // every reader of O_i is done: V_i takes its place
          if (el) s_V[i * 225 + tid] = s_T[0][tid];
        } else if (el) {
          a = s_Dv[hi_ * 15 + lo_];
        }
        if (el) s_T[1][tid] = a;
        __syncthreads();
        int cur = 1;
        // A sweep is bound by the LDS instructions the workgroup issues, and only 225 lanes hold an entry: waves 4-7 skip
        // the body and meet the others at the barrier (a non-positive pivot therefore cannot end the loop early: it is
        // remembered, the sweeps run on — on values nobody reads — and the flag is set behind the loop).
        // (The lane's pivot-column and pivot-row entries are read WITH the pivot, whether or not the lane is on the
        //  pivot's row or column: read inside that branch they were issued behind the division — read, divide, read,
        //  multiply-add, store, barrier as one chain per sweep; er, ec < 15 for every lane.)
        const bool act = tid < 256;  // wave-uniform
        bool       bad = false;
#pragma unroll 1
        for (int p = 0; p < 15; ++p) {
          if (act) {
            const double *A_  = s_T[cur];
            const double  d   = A_[p * 16];
            const double  arp = A_[er * 15 + p], apc = A_[p * 15 + ec];
            bad               = bad || !(d > 0.0);
            if (el) {
              const double inv = 1.0 / d;
              const double gen = __builtin_fma(-(arp * apc), inv, a);
              a                = (er != p && ec != p) ? gen : (er == p && ec == p) ? -inv : a * inv;
              s_T[cur ^ 1][tid] = a;
            }
          }
          cur ^= 1;
          __syncthreads();
        }
        if (bad && tid == 0) s_flag = 0;
        if (el) s_Dv[i * 225 + tid] = -a;  // inv(S_i)
        __syncthreads();
        if (!s_flag) break;  // uniform
      }
      const long long dbg_fc = wall_clock64();
      dbg_f2 += dbg_fc - dbg_fb;
      if (s_flag) {
        // 3. backward: block row i of Z = K^-1 from block row i + 1,
        //      Z(i, j) = -V_{i+1}^T Z(i+1, j)  (j > i),   Z(i, i) = inv(S_i) - V_{i+1}^T Z(i, i+1)^T,
        //    two rows in LDS (s_Z), and every lane (row rho4, columns 30 q4 .. 30 q4 + 29) picks its entries of K^-1
        //    out of the row just finished: directly when the row is in block row i, transposed when its columns are.
        // (the K^-1 layout of solveK: wave w, lane l -> row 16 w + (l & 15), column quarter l >> 4)
        const int rho4 = (tid >> 6) * 16 + (tid & 15), q4 = (tid >> 4) & 3, bi4 = rho4 / 15, rr4 = rho4 % 15;
#pragma unroll
        for (int kk = 0; kk < 30; ++kk) kinv[kk] = 0.0;
        for (int i = M - 1; i >= 0; --i) {  // uniform
          double *Zi = s_Z;  // ONE block row, rewritten in place: block (i, j) only needs block (i + 1, j)
          if (i == M - 1) {
            if (el) Zi[i * 225 + tid] = s_Dv[i * 225 + tid];
          } else {
            const double *Vn = s_V + (i + 1) * 225;
            const int     ne = (M - 1 - i) * 225;
            double        tq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // (M <= 8: at most 7 * 225 entries)
              const int e = tid + q * QP_NT;
              tq[q]       = 0.0;
              if (e < ne) {
                const int     j = i + 1 + e / 225, r = (e % 225) / 15, c = e % 15;
                const double *Zc = Zi + j * 225 + c;
                double        t  = 0.0;
#pragma unroll
                for (int k = 0; k < 15; ++k) t = __builtin_fma(Vn[k * 15 + r], Zc[k * 15], t);
                tq[q] = -t;
              }
            }
            __syncthreads();  // every read of row i + 1 is done (the lanes' K^-1 entries of it as well)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int e = tid + q * QP_NT;
              if (e < ne) Zi[(i + 1) * 225 + e] = tq[q];
            }
            __syncthreads();
            if (el) {  // the diagonal block, (max, min) on both sides
              const double *Zr = Zi + (i + 1) * 225 + lo_ * 15;
              double        t  = s_Dv[i * 225 + hi_ * 15 + lo_];
#pragma unroll
              for (int k = 0; k < 15; ++k) t = __builtin_fma(-Vn[k * 15 + hi_], Zr[k], t);
              Zi[i * 225 + tid] = t;
            }
          }
          __syncthreads();
          if (rho4 < n) {
            const int c0 = 2 * q4, c1 = 2 * q4 + 1;
            if (bi4 == i) {
              if (c0 >= i && c0 < M) {
#pragma unroll
                for (int kk = 0; kk < 15; ++kk) kinv[kk] = Zi[c0 * 225 + rr4 * 15 + kk];
              }
              if (c1 >= i && c1 < M) {
#pragma unroll
                for (int kk = 0; kk < 15; ++kk) kinv[15 + kk] = Zi[c1 * 225 + rr4 * 15 + kk];
              }
            } else if (bi4 > i) {
              if (c0 == i) {
#pragma unroll
                for (int kk = 0; kk < 15; ++kk) kinv[kk] = Zi[bi4 * 225 + kk * 15 + rr4];
              }
              if (c1 == i) {
#pragma unroll
                for (int kk = 0; kk < 15; ++kk) kinv[15 + kk] = Zi[bi4 * 225 + kk * 15 + rr4];
              }
            }
          }
          // (the next stage only reads this row until its first barrier)
        }
        __syncthreads();
      }
      dbg_f3 += wall_clock64() - dbg_fc;
      return s_flag != 0;
    } else {
    for (int e = tid; e < n * (QP_BW + 1); e += QP_NT) {
      const int i = e / (QP_BW + 1), dlt = e % (QP_BW + 1), j = i - dlt;
      double    s = 0.0;
      if (j >= 0) {
        if (i / 15 == j / 15) s = s_P[(i / 15) * 225 + (i % 15) * 15 + (j % 15)];
        if (i == j) s += qs.sigma;
        for (int a = s_cptr[i]; a < s_cptr[i + 1]; ++a) {
          const int    en = R.cidx[a], r = en >> 3;
          const double vi = R.gval[(size_t)r * QP_ELL + (en & 7)];
          for (int k = 0; k < QP_ELL; ++k)
            if (R.gcol[(size_t)r * QP_ELL + k] == j) s += vi * R.grho[r] * R.gval[(size_t)r * QP_ELL + k];
        }
        if (i / 3 == j / 3) {  // same control point: safety rows couple its 3 coordinates
          COL_DECODE(i)
          const int jd = j % 3;
          for (int f = 0; f < nface; ++f) {
            const double *v = R.sval + (size_t)(sbase + 5 * f) * 3;
            s += v[pd] * rho_cur * v[jd];
          }
        }
      }
      s_Kb[e] = s;
    }
    if (tid == 0) s_flag = 1;
    __syncthreads();
    if (wave == 0) {
      // trailing update of column j touches the (a, b) pairs 1 <= b <= a <= QP_BW: a lane's three pairs are the
      // same for every column, so the triangular index decode runs once, not n times
      int pa[3], pb[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int e = lane + 64 * q;
        int       a = 1, rem = e;
        while (rem >= a) {
          rem -= a;
          ++a;
        }
        pa[q] = e < (QP_BW * (QP_BW + 1)) / 2 ? a : 0;
        pb[q] = rem + 1;
      }
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
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int a = pa[q], b = pb[q];
          if (a > 0 && j + a < n) KB(s_Kb, j + a, j + b) -= KB(s_Kb, j + a, j) * KB(s_Kb, j + b, j);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    return s_flag != 0;
    }
  };
  auto solveK = [&]() __attribute__((always_inline)) {
    if constexpr (FAST) {
      // x~ = K^-1 rhs as a register mat-vec WITHOUT reading the right-hand side 64 times per wave: lane (q, i) of wave
      // w — q = lane >> 4 is the 16-lane DPP row, i = lane & 15 — owns K^-1[16 w + i][30 q .. 30 q + 29] and loads
      // just rhs[30 q + i] and rhs[30 q + 16 + i]; entry kk of the quarter reaches the sixteen lanes of the row through
      // the DPP broadcast of v_fmac_f64 (row_newbcast:kk reads lane kk of each row for src0): 30 fused multiply-adds, two
      // LDS reads per lane instead of thirty (the product was bound by those reads: 123 KB per iteration).  The four
      // quarter sums of a row meet through two cross-row exchanges.  (Fused multiply-adds: this solve is not on the
      // bit-exact path — the oracle factors K with a plain banded Cholesky.)  s_xt is zero beyond n, kinv too.  The
      // result goes to s_cn.
      const int    tl = launder(tid), li = tl & 15, lq = (tl >> 4) & 3;
      const double r0 = s_xt[30 * lq + li], r1 = s_xt[30 * lq + 16 + li];
      double       acc = 0.0, acc1 = 0.0;  // two chains: a dependent fp64 FMA costs 8 cycles, an independent one 4
#define QP_BC(A, R, KV, L) "v_fmac_f64_dpp " A ", " R ", " KV " row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n\t"
      asm("s_nop 1\n\t"  // (a VGPR written by the VALU needs two wait states before a DPP read)
          QP_BC("%0", "%2", "%3", 0) QP_BC("%1", "%2", "%4", 1) QP_BC("%0", "%2", "%5", 2) QP_BC("%1", "%2", "%6", 3)
          QP_BC("%0", "%2", "%7", 4) QP_BC("%1", "%2", "%8", 5) QP_BC("%0", "%2", "%9", 6) QP_BC("%1", "%2", "%10", 7)
          : "+v"(acc), "+v"(acc1)
          : "v"(r0), "v"(kinv[0]), "v"(kinv[1]), "v"(kinv[2]), "v"(kinv[3]), "v"(kinv[4]), "v"(kinv[5]), "v"(kinv[6]),
            "v"(kinv[7]));
      asm("s_nop 1\n\t"
          QP_BC("%0", "%2", "%3", 8) QP_BC("%1", "%2", "%4", 9) QP_BC("%0", "%2", "%5", 10) QP_BC("%1", "%2", "%6", 11)
          QP_BC("%0", "%2", "%7", 12) QP_BC("%1", "%2", "%8", 13) QP_BC("%0", "%2", "%9", 14) QP_BC("%1", "%2", "%10", 15)
          : "+v"(acc), "+v"(acc1)
          : "v"(r0), "v"(kinv[8]), "v"(kinv[9]), "v"(kinv[10]), "v"(kinv[11]), "v"(kinv[12]), "v"(kinv[13]), "v"(kinv[14]),
            "v"(kinv[15]));
      asm("s_nop 1\n\t"
          QP_BC("%0", "%2", "%3", 0) QP_BC("%1", "%2", "%4", 1) QP_BC("%0", "%2", "%5", 2) QP_BC("%1", "%2", "%6", 3)
          QP_BC("%0", "%2", "%7", 4) QP_BC("%1", "%2", "%8", 5) QP_BC("%0", "%2", "%9", 6) QP_BC("%1", "%2", "%10", 7)
          : "+v"(acc), "+v"(acc1)
          : "v"(r1), "v"(kinv[16]), "v"(kinv[17]), "v"(kinv[18]), "v"(kinv[19]), "v"(kinv[20]), "v"(kinv[21]),
            "v"(kinv[22]), "v"(kinv[23]));
      asm("s_nop 1\n\t"
          QP_BC("%0", "%2", "%3", 8) QP_BC("%1", "%2", "%4", 9) QP_BC("%0", "%2", "%5", 10) QP_BC("%1", "%2", "%6", 11)
          QP_BC("%0", "%2", "%7", 12) QP_BC("%1", "%2", "%8", 13)
          : "+v"(acc), "+v"(acc1)
          : "v"(r1), "v"(kinv[24]), "v"(kinv[25]), "v"(kinv[26]), "v"(kinv[27]), "v"(kinv[28]), "v"(kinv[29]));
#undef QP_BC
      acc += acc1;
      // the four quarter sums of a row meet: (q0 + q1) + (q2 + q3) in every lane, by gfx950's row / half swaps (a copy of
      // the value is swapped against itself: odd rows <-> even rows, then upper <-> lower half; one add each) instead
      // of two __shfl_xor butterflies through the LDS crossbar (ds_bpermute: ~20 instructions and two LDS round trips on
      // the iteration's critical path).  Same additions in the same order (tools/micro/permlane_rows.hip).
      {
        typedef unsigned u2v __attribute__((ext_vector_type(2)));
        unsigned lo = (unsigned)__double2loint(acc), hi = (unsigned)__double2hiint(acc);
        u2v      a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        u2v      b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        acc        = __hiloint2double((int)b.x, (int)a.x) + __hiloint2double((int)b.y, (int)a.y);
        lo = (unsigned)__double2loint(acc), hi = (unsigned)__double2hiint(acc);
        a   = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        b   = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        acc = __hiloint2double((int)b.x, (int)a.x) + __hiloint2double((int)b.y, (int)a.y);
      }
      const int row = (tl >> 6) * 16 + li;
      if (lq == 0 && row < n) s_cn[row] = acc;
    } else if (wave == 0) {
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
  // (A^T y)_j : general rows first, then safety rows (global row order)
  auto col_sum_y = [&](int j) -> double {
    double a = 0;
    for (int q = s_cptr[j]; q < s_cptr[j + 1]; ++q) {
      const int en = R.cidx[q], r = en >> 3;
      a += R.gval[(size_t)r * QP_ELL + (en & 7)] * R.gy[r];
    }
    COL_DECODE(j)
    for (int f = 0; f < nface; ++f) {
      const int s = sbase + 5 * f;
      a += R.sval[(size_t)s * 3 + pd] * R.sy[s];
    }
    return a;
  };
  // residual norms, results in s_sc: 0 pr, 1 nAx, 2 nz, 3 dr, 4 nPx, 5 nAty, 6 nq(=0) UNSCALED (the termination
  // tests); 8..13 the same six SCALED (no E / D / c: what OSQP's compute_rho_estimate reads), 14 scaled nq(=0)
  auto residuals = [&]() {
    // lane ids re-derived from an opaque copy: nothing in here is hoisted out of the ADMM loop
    const int tid = launder((int)threadIdx.x), lane = tid & 63, wave = tid >> 6;
    (void)lane;
    (void)wave;
    double m[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = 0.0;
    if (!(ablate & 32))
    for (int r = tid; r < G; r += QP_NT) {
      double s = 0;
      for (int k = 0; k < QP_ELL; ++k) {
        const int c = R.gcol[(size_t)r * QP_ELL + k];
        if (c >= 0) s += R.gval[(size_t)r * QP_ELL + k] * s_x[c];
      }
      const double e = R.gE[r], z = R.gz[r];
      const double ei = 1.0 / e;  // OSQP's Einv (scaling.c: vec_ew_recipr): the norms MULTIPLY by it (auxil.c)
      m[0]           = dmax(m[0], dabs(ei * (s - z)));
      m[1]           = dmax(m[1], dabs(ei * s));
      m[2]           = dmax(m[2], dabs(ei * z));
      m[6]           = dmax(m[6], dabs(s - z));
      m[7]           = dmax(m[7], dabs(s));
      m[8]           = dmax(m[8], dabs(z));
    }
    if (!(ablate & 32))
    for (int s = tid; s < S; s += QP_NT) {
      const int     c0 = R.sc0[s];
      const double *v  = R.sval + (size_t)s * 3;
      double        ax = 0;
      ax += v[0] * s_x[c0];
      ax += v[1] * s_x[c0 + 1];
      ax += v[2] * s_x[c0 + 2];
      const double e = R.sE[s], z = R.sz[s];
      const double ei = 1.0 / e;
      m[0]           = dmax(m[0], dabs(ei * (ax - z)));
      m[1]           = dmax(m[1], dabs(ei * ax));
      m[2]           = dmax(m[2], dabs(ei * z));
      m[6]           = dmax(m[6], dabs(ax - z));
      m[7]           = dmax(m[7], dabs(ax));
      m[8]           = dmax(m[8], dabs(z));
    }
    if (!(ablate & 64))
    for (int j = tid; j < n; j += QP_NT) {
      double        s  = 0;
      const double *Pb = s_P + (j / 15) * 225 + (j % 15) * 15;
      const int     b0 = (j / 15) * 15;
      for (int k = 0; k < 15; ++k) s += Pb[k] * s_x[b0 + k];
      const double a  = col_sum_y(j);
      const double dj = 1.0 / s_D[j];  // Dinv
      m[3]            = dmax(m[3], dabs(dj * (s + a)));
      m[4]            = dmax(m[4], dabs(dj * s));
      m[5]            = dmax(m[5], dabs(dj * a));
      m[9]            = dmax(m[9], dabs(s + a));
      m[10]           = dmax(m[10], dabs(s));
      m[11]           = dmax(m[11], dabs(a));
    }
    // the twelve maxima share one wave-reduce / LDS / barrier round
    if (!(ablate & 128))
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = wave_max(m[k]);
    __syncthreads();
    if (lane == 0) {
      double *o = s_red12 + wave * 12;
#pragma unroll
      for (int k = 0; k < 12; ++k) o[k] = m[k];
    }
    __syncthreads();
    if (tid < 12) {
      double v = s_red12[tid];
      for (int w = 1; w < QP_NT / 64; ++w) v = dmax(v, s_red12[12 * w + tid]);
      s_sc[tid < 6 ? tid : tid + 2] = tid == 3 ? v * cinv : v;
    }
    if (tid == 12) s_sc[6] = s_sc[14] = 0.0;
    __syncthreads();
  };

  // ---- register-resident row / column state of the fast iteration ------------------------------------
  // Row roles: lanes 0..255 (waves 0-3) own general row r = tid; lanes 256..511 (waves 4-7) own safety rows
  // s = (tid - 256) + 256 u (u < 4) — two waves per SIMD, one of each kind.  Column role: the lane quad
  // (j = tid >> 2, q = tid & 3) shares column j — its K^-1 row, its x_j and the A^T w product, whose entries
  // (<= 7 general rows, nface safety rows) are dealt round-robin to the four lanes.
  double *const h_sv = (double *)(qp_smem + head);  // == R.sval / R.sw / R.gw when fast, typed as LDS
  double *const h_sw = h_sv + (size_t)S * 3;
  double *const h_gw = h_sw + S;
  // A lane is EITHER a general-row lane or a safety-row lane (wave-uniform), so the two kinds of row state
  // share one set of registers: rs[0..5] = gv | rs[6] rho, [7] 1/rho, [8] l, [9] u, [10] z, [11] y (general)
  //                             rs[6u + 0..2] = normal, [6u + 3] = u, [6u + 4] = z, [6u + 5] = y  (safety slot u)
  double rs[24];
  int    ri[QP_ELL];  // gc[k] | s_c[u]
#define gv(k) rs[k]
#define gc(k) ri[k]
#define g_rho rs[6]
#define g_rinv rs[7]
#define g_lo rs[8]
#define g_hi rs[9]
#define g_z rs[10]
#define g_y rs[11]
#define sv(u, d) rs[6 * (u) + (d)]
#define s_hi(u) rs[6 * (u) + 3]
#define s_zr(u) rs[6 * (u) + 4]
#define s_yr(u) rs[6 * (u) + 5]
#define s_c(u) ri[u]
  double cv[2], xj = 0.0;
  int    cr[2];
#pragma unroll
  for (int k = 0; k < 24; ++k) rs[k] = 0.0;
#pragma unroll
  for (int k = 0; k < QP_ELL; ++k) ri[k] = 0;
  cv[0] = cv[1] = 0.0;
  cr[0] = cr[1] = 0;
  const bool grole = tid < 256;            // wave-uniform: general-row waves / safety-row waves
  const bool grow = tid < G, ccol = tid < 4 * n;
  int        fj_sv = 0, fj_sw = 0, fj_n = 0;  // this lane's faces of column j: f = q + 4 i, i < fj_n
  // Both loaders assign every variable unconditionally (clamped index + select, no branch): the state is then
  // dead across a refactorisation, which needs the whole register file.
  auto fast_load = [&]() __attribute__((always_inline)) {
    const int t  = launder(tid);
    const int tg = grow ? t : 0;
    if (grole) {  // wave-uniform
#pragma unroll
      for (int k = 0; k < QP_ELL; ++k) {
        const int    c = R.gcol[(size_t)tg * QP_ELL + k];
        const double v = R.gval[(size_t)tg * QP_ELL + k];
        gc(k)          = (c < 0 || !grow) ? 0 : c;
        gv(k)          = (c < 0 || !grow) ? 0.0 : v;
      }
      g_lo = R.gl[tg];
      g_hi = R.gu[tg];
      g_z  = R.gz[tg];
      g_y  = R.gy[tg];
#pragma unroll
      for (int k = 12; k < 24; ++k) rs[k] = 0.0;
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int  sr = (t - 256) + 256 * u;
        const bool ok = sr < S;
        const int  sc = ok ? sr : 0;
        const double a0 = R.sval[(size_t)sc * 3 + 0], a1 = R.sval[(size_t)sc * 3 + 1], a2 = R.sval[(size_t)sc * 3 + 2];
        const double hi = R.su[sc], z = R.sz[sc], y = R.sy[sc];
        const int    c0 = R.sc0[sc];
        sv(u, 0) = ok ? a0 : 0.0;
        sv(u, 1) = ok ? a1 : 0.0;
        sv(u, 2) = ok ? a2 : 0.0;
        s_hi(u)  = ok ? hi : 0.0;
        s_zr(u)  = ok ? z : 0.0;
        s_yr(u)  = ok ? y : 0.0;
        s_c(u)   = ok ? c0 : 0;
      }
      ri[4] = ri[5] = 0;
    }
    {
      const int j = ccol ? t >> 2 : 0, q = t & 3;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int    qq = s_cptr[j] + q + 4 * e;
        const bool   ok = ccol && qq < s_cptr[j + 1];
        const int    en = R.cidx[ok ? qq : s_cptr[j]];
        const double v  = R.gval[(size_t)(en >> 3) * QP_ELL + (en & 7)];
        cr[e]           = ok ? en >> 3 : 0;
        cv[e]           = ok ? v : 0.0;
      }
      COL_DECODE(j)
      fj_sv = (sbase + 5 * q) * 3 + pd;
      fj_sw = sbase + 5 * q;
      fj_n  = ccol ? (nface - q + 3) >> 2 : 0;
      xj    = s_x[j];
    }
  };
  // after set_rho(): the general row's rho and its w = rho z - y (set_rho refreshed the safety rows' w)
  auto fast_rho = [&]() __attribute__((always_inline)) {
    const int t  = launder(tid);
    const int tg = grow ? t : 0;
    if (grole) {
      g_rho  = R.grho[tg];
      g_rinv = R.grinv[tg];
      if (grow) h_gw[t] = g_rho * g_z - g_y;
    }
  };
  // The termination check's view of y: the column quads' A^T y reads every row's y, which lives in the row lanes'
  // registers.  It is staged through LDS — factor()'s block area (s_Dv ..), dead between two factorisations — and not
  // through the row storage: for the largest problems (8 pieces) the cold row arrays sit in HBM scratch, and a check
  // paid a store + fence + dependent L2 loads per face for it (the column loop was a chain of L2 round trips).
  double *const c_sy = (double *)qp_smem + (size_t)M * 225;  // [S] safety rows' y (FAST only: == s_Dv)
  double *const c_gy = c_sy + S;                             // [G] general rows' y
  double *const c_red = c_gy + G;                            // [32][10] the check's row-level partial reductions
  auto fast_stage_y = [&]() __attribute__((always_inline)) {
    const int t = launder(tid);
    if (grole) {
      if (grow) c_gy[t] = g_y;
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int sr = (t - 256) + 256 * u;
        if (sr < S) c_sy[sr] = s_yr(u);
      }
    }
    if (ccol && (t & 3) == 0) s_x[t >> 2] = xj;
    __syncthreads();
  };
  // registers -> row storage: before a refactorisation (set_rho() and fast_load() read z and y from there)
  auto fast_spill = [&]() __attribute__((always_inline)) {
    const int t = launder(tid);
    if (grole) {
      if (grow) {
        R.gz[t] = g_z;
        R.gy[t] = g_y;
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int sr = (t - 256) + 256 * u;
        if (sr < S) {
          R.sz[sr] = s_zr(u);
          R.sy[sr] = s_yr(u);
        }
      }
    }
    if (ccol && (t & 3) == 0) s_x[t >> 2] = xj;
    __syncthreads();
  };

  // Primal infeasibility certificate (OSQP auxil.c is_primal_infeasible; called when the primal residual test
  // failed): delta_y is first projected onto the polar of the recession cone of [l, u] — general rows have two
  // finite bounds (untouched), every safety row has l = -OSQP_INFTY, so only its positive part counts (the stored
  // delta_y is overwritten like OSQP's work vector; the next iteration recomputes it) — then, relative to the
  // unscaled ||dy||inf:  u'(dy)+ + l'(dy)- < -eps ||dy||  and  ||Dinv A'dy||inf < eps ||dy||.  Workgroup-uniform.
  // Stage 1 (the projection, ||dy|| -> s_sc[7], the bound product -> s_sc[15]) is part of the register-resident
  // residual pass on the fast path; cert_stage1 is the same over the row storage.
  auto cert_stage1 = [&]() {
    const int tid = launder((int)threadIdx.x), lane = tid & 63, wave = tid >> 6;
    double    ndy = 0, lhs = 0;
    for (int r = tid; r < G; r += QP_NT) {
      const double d = R.gdy[r];
      ndy            = dmax(ndy, dabs(R.gE[r] * d));
      lhs += R.gu[r] * (d > 0 ? d : 0) + R.gl[r] * (d < 0 ? d : 0);
    }
    for (int s = tid; s < S; s += QP_NT) {
      const double d = R.sdy[s] > 0.0 ? R.sdy[s] : 0.0;
      R.sdy[s]       = d;
      ndy            = dmax(ndy, dabs(R.sE[s] * d));
      lhs += R.su[s] * d;  // projected: dy >= 0
    }
    ndy = block_max(ndy, s_red);
    lhs = wave_sum(lhs);
    __syncthreads();
    if (lane == 0) s_red[wave] = lhs;
    __syncthreads();
    if (tid == 0) {
      s_sc[7]  = ndy;
      s_sc[15] = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) + ((s_red[4] + s_red[5]) + (s_red[6] + s_red[7]));
    }
    __syncthreads();
  };
  auto primal_infeasible = [&](double eps_inf) -> bool {
    const int tid = launder((int)threadIdx.x);
    if constexpr (!FAST) cert_stage1();
    const double ndy = s_sc[7], lhs = s_sc[15];
    if (!(ndy > eps_inf)) return false;
    if (!(lhs < -eps_inf * ndy)) return false;
    double na = 0;
    if constexpr (FAST) {
      if (ccol) {  // A^T dy on the column quads, as the dual norms above
        const int     j = tid >> 2;
        double        a = __builtin_fma(cv[1], R.gdy[cr[1]], cv[0] * R.gdy[cr[0]]);
        const double *fvp = h_sv + fj_sv, *fyp = R.sdy + fj_sw;
        const double *fv0 = fvp, *fy0 = fyp;
#pragma unroll 1
        for (int i0 = 0; i0 < fj_n; i0 += 4, fvp += 240, fyp += 80) {  // four faces per round trip (delta_y may sit in HBM scratch)
          double av[4], bv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool ok = i0 + u < fj_n;
            av[u]         = *(ok ? fvp + 60 * u : fv0);
            bv[u]         = *(ok ? fyp + 20 * u : fy0);
            bv[u]         = ok ? bv[u] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) a = __builtin_fma(av[u], bv[u], a);
        }
        a += dpp_quad(a, 0xB1);
        a += dpp_quad(a, 0x4E);
        na = dabs((1.0 / s_D[j]) * a);
      }
    } else {
      for (int j = tid; j < n; j += QP_NT) {
        double a = 0;
        for (int q = s_cptr[j]; q < s_cptr[j + 1]; ++q) {
          const int en = R.cidx[q], r = en >> 3;
          a += R.gval[(size_t)r * QP_ELL + (en & 7)] * R.gdy[r];
        }
        COL_DECODE(j)
        for (int f = 0; f < nface; ++f) {
          const int sr = sbase + 5 * f;
          a += R.sval[(size_t)sr * 3 + pd] * R.sdy[sr];
        }
        na = dmax(na, dabs((1.0 / s_D[j]) * a));
      }
    }
    na = block_max(na, s_red);
    __syncthreads();
    return na < eps_inf * ndy;
  };

  // The residual pass of the register-resident iteration (fast path): the row norms and the certificate's stage 1
  // come straight from the row state in registers (one batch of x loads per lane; |v| / e is monotone in |v|, so
  // max(|Ax|, |z|) / e replaces two of the three divisions per row), the dual norms from one lane per column in the
  // row order of the oracle's dense sums; ten wave reductions on the DPP path, one LDS round for the eight waves.
  // Fills s_sc like residuals() (1 / 4 / 9 / 12 hold the pair maxima, their partners 0) plus s_sc[7], s_sc[15].
  auto fast_residuals = [&]() __attribute__((always_inline)) {
    const int  t = launder(tid), lane = t & 63, wave = t >> 6;
    const bool f32 = qs.residual_fp32 != 0;  // BASELINE configs[4]: residual norms / reductions in fp32
    // |x| as the operand modifier and max as v_max_f64: written as "x < 0 ? -x : x" / "a > b ? a : b" each cost a compare
    // and two selects per use, ~50 a row slot (they differ from these in the sign of a zero and in which NaN survives:
    // neither reaches a threshold test)
    auto fab = [](double x) __attribute__((always_inline)) { return __builtin_fabs(x); };
    auto fmx = [](double a, double b) __attribute__((always_inline)) { return __builtin_fmax(a, b); };
    double    v[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) v[k] = 0.0;
    if (grole) {  // wave-uniform
      double xg[QP_ELL];
#pragma unroll
      for (int k = 0; k < QP_ELL; ++k) xg[k] = s_x[gc(k)];
      const int    tg = grow ? t : 0;
      const double e = R.gE[tg], d = R.gdy[tg];
      double       ax = 0.0;
#pragma unroll
      for (int k = 0; k < QP_ELL; ++k) ax += gv(k) * xg[k];  // absent entries are 0 * x[0]
      if (grow) {
        const double r_ = fab(ax - g_z), n_ = fmx(fab(ax), fab(g_z));
        const double ei = 1.0 / e;  // OSQP's Einv: the norms multiply by the stored reciprocal (one division per row)
        v[0] = f32 ? (double)((float)r_ / (float)e) : ei * r_;
        v[1] = f32 ? (double)((float)n_ / (float)e) : ei * n_;
        v[2] = r_;
        v[3] = n_;
        v[8] = fab(e * d);
        v[9] = g_hi * (d > 0 ? d : 0) + g_lo * (d < 0 ? d : 0);
      }
    } else {
      // every slot's E and delta_y first: behind the store of a projected delta_y the next slot's loads could not be
      // moved up (the compiler cannot tell the arrays apart), and with the cold rows in HBM scratch each slot then
      // cost an L2 round trip of its own
      double se[4], sd[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int sr = (t - 256) + 256 * u, sc = sr < S ? sr : 0;
        se[u] = R.sE[sc];
        sd[u] = R.sdy[sc];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (256 * u >= S) continue;  // workgroup-uniform
        const int    sr = (t - 256) + 256 * u;
        const bool   ok = sr < S;
        const double x0 = s_x[s_c(u)], x1 = s_x[s_c(u) + 1], x2 = s_x[s_c(u) + 2];
        const double e = se[u], dr_ = sd[u];
        double       ax = 0.0;
        ax += sv(u, 0) * x0;
        ax += sv(u, 1) * x1;
        ax += sv(u, 2) * x2;
        if (ok) {
          const double d  = dr_ > 0.0 ? dr_ : 0.0;  // projection onto the polar of the recession cone
          R.sdy[sr]       = d;
          const double r_ = fab(ax - s_zr(u)), n_ = fmx(fab(ax), fab(s_zr(u)));
          const double ei = 1.0 / e;
          v[0] = fmx(v[0], f32 ? (double)((float)r_ / (float)e) : ei * r_);
          v[1] = fmx(v[1], f32 ? (double)((float)n_ / (float)e) : ei * n_);
          v[2] = fmx(v[2], r_);
          v[3] = fmx(v[3], n_);
          v[8] = fmx(v[8], fab(e * d));
          v[9] += s_hi(u) * d;
        }
      }
    }
    if (ccol) {
      // dual norms on the column quads of the iteration (lane q of column j holds every fourth entry of the column —
      // its general-row entries in cv / cr, its faces behind fj_sv / fj_sw — and every fourth term of row j of P): one
      // batch of loads, two DPP exchanges, instead of one lane walking the column (the oracle's row order is not
      // kept: the sums differ from its dense ones in the last bits, the norms are compared with 1e-3 tolerances)
      const int     j = t >> 2, q = t & 3, b0 = (j / 15) * 15;
      const double *Pb = s_P + (j / 15) * 225 + (j % 15) * 15;
      double        pk[4], xk[4], yg[2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = q + 4 * u, kk = k < 15 ? k : 0;
        pk[u]       = Pb[kk];
        xk[u]       = s_x[b0 + kk];
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) yg[e] = c_gy[cr[e]];
      double        s_ = 0.0, a_ = 0.0;
      {  // the faces' terms in face order, eight independent LDS reads per round trip (as the iteration's A^T w)
        const double *fvp = h_sv + fj_sv, *fyp = c_sy + fj_sw;
        const double *fv0 = fvp, *zero = s_xt + 127;  // s_xt is zero beyond n
#pragma unroll 1
        for (int i0 = 0; i0 < fj_n; i0 += 4, fvp += 240, fyp += 80) {
          double a[4], b[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool ok = i0 + u < fj_n;
            a[u]          = *(ok ? fvp + 60 * u : fv0);
            b[u]          = *(ok ? fyp + 20 * u : zero);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 4; ++u) a_ = __builtin_fma(a[u], b[u], a_);  // (an absent face adds v * 0)
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) s_ = __builtin_fma(q + 4 * u < 15 ? pk[u] : 0.0, xk[u], s_);
#pragma unroll
      for (int e = 0; e < 2; ++e) a_ = __builtin_fma(cv[e], yg[e], a_);
      s_ += dpp_quad(s_, 0xB1);
      a_ += dpp_quad(a_, 0xB1);
      s_ += dpp_quad(s_, 0x4E);
      a_ += dpp_quad(a_, 0x4E);
      if (q == 0) {
        const double dj = s_D[j], di = 1.0 / dj;  // Dinv
        const double r_ = fab(s_ + a_), n_ = fmx(fab(s_), fab(a_));
        v[4] = f32 ? (double)((float)r_ / (float)dj) : di * r_;
        v[5] = f32 ? (double)((float)n_ / (float)dj) : di * n_;
        v[6] = r_;
        v[7] = n_;
      }
    }
    if (f32) {  // the nine maxima as fp32 values (rounded up to the next float: a maximum must not shrink)
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        float fv = (float)v[k];
        if ((double)fv < v[k]) fv = __int_as_float(__float_as_int(fv) + 1);  // next float up (v >= 0, finite)
        v[k] = (double)fv;  // (the maximum of floats is the same number whichever width carries it)
      }
    }
    // Ten reductions over 512 lanes: inside each 16-lane row on the DPP path (two quad permutes, row_half_mirror,
    // row_mirror), then ONE LDS round over the 32 row results of the workgroup (c_red, behind the staged y) instead of
    // four readlanes + three operations per value and wave: 40 lanes — value k, waves 2 p and 2 p + 1 — fold eight rows
    // each and meet through two quad permutes.  The bound product keeps its summation tree: rows (r0 + r16) + (r32 + r48)
    // per wave, waves ((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7)).
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      double x = v[k];
      x        = __builtin_fmax(x, dpp_ctrl_t<0xB1>(x));
      x        = __builtin_fmax(x, dpp_ctrl_t<0x4E>(x));
      x        = __builtin_fmax(x, dpp_ctrl_t<0x141>(x));
      v[k]     = __builtin_fmax(x, dpp_ctrl_t<0x140>(x));
    }
    {
      double x = v[9];
      x += dpp_ctrl_t<0xB1>(x);
      x += dpp_ctrl_t<0x4E>(x);
      x += dpp_ctrl_t<0x141>(x);
      v[9] = x + dpp_ctrl_t<0x140>(x);
    }
    if ((lane & 15) == 0) {
      double *o = c_red + (wave * 4 + (lane >> 4)) * 10;
#pragma unroll
      for (int k = 0; k < 10; ++k) o[k] = v[k];
    }
    __syncthreads();
    if (t < 40) {  // (wave 0)
      const int     k = t >> 2, pq = t & 3;
      const double *o = c_red + (size_t)(8 * pq) * 10 + k;  // rows 8 pq .. 8 pq + 7: waves 2 pq, 2 pq + 1
      double        r_;
      if (k == 9) {
        r_ = ((o[0] + o[10]) + (o[20] + o[30])) + ((o[40] + o[50]) + (o[60] + o[70]));
        r_ += dpp_quad(r_, 0xB1);
        r_ += dpp_quad(r_, 0x4E);
      } else {
        r_ = __builtin_fmax(__builtin_fmax(__builtin_fmax(o[0], o[10]), __builtin_fmax(o[20], o[30])),
                            __builtin_fmax(__builtin_fmax(o[40], o[50]), __builtin_fmax(o[60], o[70])));
        r_ = __builtin_fmax(r_, dpp_quad(r_, 0xB1));
        r_ = __builtin_fmax(r_, dpp_quad(r_, 0x4E));
      }
      // slots: pr 0 | max(nAx, nz) 1 | sc_pr 8 | max(sc_nAx, sc_nz) 9 | dr 3 | max(nPx, nAty) 4 | sc_dr 11 |
      // max(sc_nPx, sc_nAty) 12 | ||dy|| 7 | bound product 15
      const int slot = k == 0 ? 0 : k == 1 ? 1 : k == 2 ? 8 : k == 3 ? 9 : k == 4 ? 3 : k == 5 ? 4 : k == 6 ? 11 : k == 7 ? 12 : k == 8 ? 7 : 15;
      if (pq == 0) s_sc[slot] = k == 4 ? r_ * cinv : r_;
    }
    if (t >= 64 && t < 70)  // the partners of the pair maxima, and ||q|| = 0 (a lane of wave 1)
      s_sc[t == 64 ? 2 : t == 65 ? 5 : t == 66 ? 6 : t == 67 ? 10 : t == 68 ? 13 : 14] = 0.0;
    __syncthreads();
  };

  set_rho();
  if constexpr (FAST) build_K1();
  const long long dbg_s3 = wall_clock64() - dbg_t0;
  bool chol_ok = factor();
  const long long dbg_setup = wall_clock64() - dbg_t0;
  int  status = -2, iter = 0;
  if (!chol_ok) status = -7;
  if constexpr (FAST) {
    fast_load();
    fast_rho();
    __syncthreads();
  }

  // ---- 4. ADMM iterations
  int cj_q0 = 0, cj_q1 = 0, cj_sbase = 0, cj_nface = 0, cj_pd = 0;
  if (tid < n) {
    COL_DECODE(tid)
    cj_q0    = s_cptr[tid];
    cj_q1    = s_cptr[tid + 1];
    cj_sbase = sbase;
    cj_nface = nface;
    cj_pd    = pd;
  }
  // loop constants as per-lane values: read from the kernel-argument SGPR tuple they would be re-loaded eight
  // dwords at a time (the tuple spills as a unit) at every use
  double alpha = qs.alpha, oma = 1.0 - qs.alpha, sigma_v = qs.sigma;
  asm volatile("" : "+v"(alpha), "+v"(oma), "+v"(sigma_v));
  const double *xtv   = FAST ? s_cn : s_xt;  // where the solve leaves x~
  for (int j = n + tid; j < 128; j += QP_NT) s_xt[j] = 0.0;  // the register mat-vec reads 120 entries
  __syncthreads();
  // One ADMM iteration (three barriers).  STORE = this is the last iteration before a termination check: the
  // rows also store delta_y (the infeasibility certificate reads it).  The iterations between two checks run in an
  // inner loop of their own whose body is only this lambda, so that nothing of the check / refactorisation code is
  // live in it (register allocation of the hot loop: no SGPR spill traffic).
  auto iteration = [&](auto store_tag, auto ns_tag) __attribute__((always_inline)) {
      constexpr bool STORE = decltype(store_tag)::value;
      constexpr int  NS    = decltype(ns_tag)::value;  // safety-row slots in use: ceil(S / 256), a compile-time count
      (void)NS;
      // (a) rhs_j = sigma x_j - q_j + sum_rows A[r][j] (rho_r z_r - y_r)
      if constexpr (FAST) {
        if (!(ablate & 1) && ccol) {
          double w[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) w[e] = h_gw[cr[e]];
          double        p   = 0.0;
          const double *fvp = h_sv + fj_sv, *fwp = h_sw + fj_sw;
          const double *fv0 = fvp, *zero = s_xt + 127;  // s_xt is zero beyond n
#pragma unroll 1
          for (int i0 = 0; i0 < fj_n; i0 += 4, fvp += 240, fwp += 80) {  // 8 independent LDS reads per round trip
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const bool ok = i0 + u < fj_n;
              a[u]          = *(ok ? fvp + 60 * u : fv0);
              b[u]          = *(ok ? fwp + 20 * u : zero);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) p = __builtin_fma(a[u], b[u], p);
          }
#pragma unroll
          for (int e = 0; e < 2; ++e) p = __builtin_fma(cv[e], w[e], p);
          p += dpp_quad(p, 0xB1);  // lane ^ 1
          p += dpp_quad(p, 0x4E);  // lane ^ 2
          if ((tid & 3) == 0) s_xt[launder(tid) >> 2] = __builtin_fma(sigma_v, xj, p);  // q == 0
        }
      } else
      if (!(ablate & 1))
      if (tid < n) {  // n <= 240 < 256: one column per lane, its constants hoisted out of the loop (cj_*)
        const int j = tid;
        double    s = qs.sigma * s_x[j];  // q == 0
        for (int q = cj_q0; q < cj_q1; ++q) {
          const int en = R.cidx[q], r = en >> 3;
          s += R.gval[(size_t)r * QP_ELL + (en & 7)] * (R.grho[r] * R.gz[r] - R.gy[r]);
        }
        const double *sv = R.sval + (size_t)cj_sbase * 3 + cj_pd;
        const double *sw = R.sw + cj_sbase;
        for (int f = 0; f < cj_nface; ++f) s += sv[(size_t)15 * f] * sw[5 * f];
        s_xt[j] = s;
      }
      __syncthreads();
      // (b) x~ = K^-1 rhs
      if (!(ablate & 2)) solveK();
      // (c) rows: z~ = A x~ ; z = proj(alpha z~ + (1-alpha) z + y/rho) ; y += rho (.. - z)
      if constexpr (FAST) {
        if (!(ablate & 12)) {
          const int    t   = launder(tid);
          const double xtj = xtv[ccol ? t >> 2 : 0];
          if (grole) {  // waves 0-3: one general row per lane
            double xg[QP_ELL];  // every read of x~ issued up front: one LDS round trip
#pragma unroll
            for (int k = 0; k < QP_ELL; ++k) xg[k] = xtv[gc(k)];
            __builtin_amdgcn_sched_barrier(0);
            // fused multiply-adds: like the K^-1 product, the row update is not on a bit-exact path
            double s = gv(0) * xg[0];
#pragma unroll
            for (int k = 1; k < QP_ELL; ++k) s = __builtin_fma(gv(k), xg[k], s);
            const double zr = __builtin_fma(alpha, s, oma * g_z);
            double       v  = __builtin_fma(g_rinv, g_y, zr);  // OSQP update_z: rho_inv_vec[i] * y[i]
            v               = __builtin_fmin(__builtin_fmax(v, g_lo), g_hi);  // (l <= u: the clamp, two instructions)
            g_z             = v;
            const double d  = g_rho * (zr - v);
            g_y             = g_y + d;
            if (grow) {
              h_gw[t] = __builtin_fma(g_rho, v, -g_y);
              if constexpr (STORE) R.gdy[t] = d;
            }
          } else {  // waves 4-7: up to four safety rows per lane
            // (NS, the number of slots that hold rows, is a template argument of the loop between two checks: with the
            //  run-time test "256 u < S" per slot the compiler kept a zero fill, a materialised flag and two branches
            //  per slot in the hot loop)
            double xs[NS][3];
#pragma unroll
            for (int u = 0; u < NS; ++u) {
              xs[u][0] = xtv[s_c(u)];
              xs[u][1] = xtv[s_c(u) + 1];
              xs[u][2] = xtv[s_c(u) + 2];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NS; ++u) {
              double ax = sv(u, 0) * xs[u][0];
              ax              = __builtin_fma(sv(u, 1), xs[u][1], ax);
              ax              = __builtin_fma(sv(u, 2), xs[u][2], ax);
              const double zr = __builtin_fma(alpha, ax, oma * s_zr(u));
              double       v  = __builtin_fma(rinv_cur, s_yr(u), zr);
              v               = __builtin_fmin(v, s_hi(u));  // l = -OSQP_INFTY
              s_zr(u)         = v;
              const double d  = rho_cur * (zr - v);
              const double yn = s_yr(u) + d;
              s_yr(u)         = yn;
              const int sr    = (t - 256) + 256 * u;
              if (u + 1 < NS || sr < S) {  // (only the last slot in use can hold rows beyond S)
                h_sw[sr] = __builtin_fma(rho_cur, v, -yn);
                if constexpr (STORE) R.sdy[sr] = d;
              }
            }
          }
          xj = __builtin_fma(alpha, xtj, oma * xj);
        }
        __syncthreads();
      } else {
      if (!(ablate & 4))
      for (int r = tid; r < G; r += QP_NT) {
        double s = 0;
        for (int k = 0; k < QP_ELL; ++k) {
          const int c = R.gcol[(size_t)r * QP_ELL + k];
          if (c >= 0) s += R.gval[(size_t)r * QP_ELL + k] * xtv[c];
        }
        const double rho = R.grho[r], yr = R.gy[r];
        const double zr  = alpha * s + (1.0 - alpha) * R.gz[r];
        double       v   = zr + R.grinv[r] * yr;  // OSQP update_z: rho_inv_vec[i] * y[i]
        const double lo = R.gl[r], hi = R.gu[r];
        v               = v < lo ? lo : (v > hi ? hi : v);
        R.gz[r]         = v;
        const double d  = rho * (zr - v);
        R.gdy[r]        = d;
        R.gy[r]         = yr + d;
      }
      if (!(ablate & 8))
      for (int s = tid; s < S; s += QP_NT) {
        const int     c0 = R.sc0[s];
        const double *vv = R.sval + (size_t)s * 3;
        double        ax = 0;
        ax += vv[0] * xtv[c0];
        ax += vv[1] * xtv[c0 + 1];
        ax += vv[2] * xtv[c0 + 2];
        const double yr = R.sy[s];
        const double zr = alpha * ax + (1.0 - alpha) * R.sz[s];
        double       v  = zr + rinv_cur * yr;
        const double hi = R.su[s];
        v               = v > hi ? hi : v;  // l = -OSQP_INFTY
        R.sz[s]         = v;
        const double d  = rho_cur * (zr - v);
        R.sdy[s]        = d;
        const double yn = yr + d;
        R.sy[s]         = yn;
        R.sw[s]         = rho_cur * v - yn;
      }
      for (int j = tid; j < n; j += QP_NT) s_x[j] = alpha * xtv[j] + (1.0 - alpha) * s_x[j];
      __syncthreads();
      }
  };
  int adapt_left = qs.adaptive_rho_interval, check_left = qs.check_termination;
  bool finished = !chol_ok;
  if (chol_ok) {
    const int max_iter = qs.max_iter, adapt_iv = qs.adaptive_rho_interval, check_iv = qs.check_termination;
    while (iter < max_iter) {
      // iterations up to the next check / rho update (iter % interval == 0), the last one storing delta_y
      int chunk = max_iter - iter;
      if (adapt_iv > 0 && adapt_left < chunk) chunk = adapt_left;
      if (check_iv > 0 && check_left < chunk) chunk = check_left;
      auto run_chunk = [&](auto ns_tag) __attribute__((always_inline)) {
#pragma unroll 1
        for (int k = 1; k < chunk; ++k) iteration(std::false_type{}, ns_tag);
        iteration(std::true_type{}, ns_tag);
      };
      if constexpr (FAST) {
        switch (__builtin_amdgcn_readfirstlane((S + 255) >> 8)) {  // (FAST: S <= 1024)
          case 0:
          case 1: run_chunk(std::integral_constant<int, 1>{}); break;
          case 2: run_chunk(std::integral_constant<int, 2>{}); break;
          case 3: run_chunk(std::integral_constant<int, 3>{}); break;
          default: run_chunk(std::integral_constant<int, 4>{}); break;
        }
      } else {
        run_chunk(std::integral_constant<int, 4>{});
      }
      iter += chunk;
      adapt_left -= chunk;
      check_left -= chunk;
      const bool do_adapt = adapt_iv > 0 && adapt_left == 0, do_check = check_iv > 0 && check_left == 0;
      if (do_adapt) adapt_left = adapt_iv;
      if (do_check) check_left = check_iv;
      if (!do_adapt && !do_check) continue;  // (only when max_iter ended the chunk)
      const long long dbg_c0 = wall_clock64();
      ++dbg_ncheck;
      if constexpr (FAST) {
        fast_stage_y();
        const long long dbg_c1 = wall_clock64();
        dbg_k1 += dbg_c1 - dbg_c0;
        fast_residuals();
        dbg_k2 += wall_clock64() - dbg_c1;
      } else {
        residuals();
      }
      if (do_check) {
        const double eps_prim = qs.eps_abs + qs.eps_rel * dmax(s_sc[1], s_sc[2]);
        const double eps_dual =
            qs.eps_abs + qs.eps_rel * cinv * dmax(dmax(s_sc[4], s_sc[5]), s_sc[6]);
        const bool p_ok = s_sc[0] < eps_prim, d_ok = s_sc[3] < eps_dual;
        if (p_ok && d_ok) {
          status   = 1;
          finished = true;
          dbg_check += wall_clock64() - dbg_c0;
          break;
        }
        if (ablate & 16) continue;  // profiling aid (ablation build only)
        if ((ablate & 256) && !p_ok) continue;
        if (!p_ok && primal_infeasible(1e-4)) {  // eps_prim_inf default
          status   = -3;
          finished = true;
          dbg_check += wall_clock64() - dbg_c0;
          break;
        }
      }
      dbg_check += wall_clock64() - dbg_c0;
      // adaptive rho after the termination test, on the same residual evaluation (osqp.c: update_info runs once);
      // the estimate reads the SCALED residuals and norms (auxil.c compute_rho_estimate, see the oracle)
      if (do_adapt) {
        const double pr_n = s_sc[8] / (dmax(s_sc[10], s_sc[9]) + 1e-10);
        const double du_n = s_sc[11] / (dmax(dmax(s_sc[14], s_sc[13]), s_sc[12]) + 1e-10);
        double rho_new = rho_cur * sogm_det::sqrt_rn(pr_n / (du_n + 1e-10));
        rho_new        = rho_new < RHO_MIN ? RHO_MIN : (rho_new > 1e6 ? 1e6 : rho_new);
        if (rho_new > rho_cur * 5.0 || rho_new < rho_cur / 5.0) {
          const long long dbg_r0 = wall_clock64();
          ++dbg_nrefac;
          rho_cur  = rho_new;
          rinv_cur = 1.0 / rho_cur;
          if constexpr (FAST) fast_spill();  // z, y: set_rho() and fast_load() read them from the row storage
          set_rho();
          if (!factor()) {
            status   = -7;
            finished = true;
            break;
          }
          if constexpr (FAST) {
            fast_load();
            fast_rho();
            __syncthreads();
          }
          dbg_refac += wall_clock64() - dbg_r0;
        }
      }
    }
    if (!finished) {  // max_iter reached
      if constexpr (FAST) {
        fast_stage_y();
        fast_residuals();
      } else {
        residuals();
      }
      const double eps_prim = qs.eps_abs * 10 + qs.eps_rel * 10 * dmax(s_sc[1], s_sc[2]);
      const double eps_dual =
          qs.eps_abs * 10 + qs.eps_rel * 10 * cinv * dmax(dmax(s_sc[4], s_sc[5]), s_sc[6]);
      // osqp_solve's epilogue: check_termination(approximate = 1), every tolerance x 10, else MAX_ITER_REACHED
      const bool p_ok = s_sc[0] < eps_prim;
      if (p_ok && s_sc[3] < eps_dual)
        status = 2;  // OSQP_SOLVED_INACCURATE
      else if (!p_ok && primal_infeasible(1e-4 * 10))
        status = 3;  // OSQP_PRIMAL_INFEASIBLE_INACCURATE
      else
        status = -2;
    }
  }
  __syncthreads();
  double *out = out_cpts + (size_t)agent * SOGM_MAX_PIECES * 15;
  for (int j = tid; j < SOGM_MAX_PIECES * 15; j += QP_NT) out[j] = j < n ? s_D[j] * s_x[j] : 0.0;
  if (tid == 0) {
    out_status[agent] = status;
    out_iters[agent]  = iter;
    if (ws.dbg) {
      long long *o = ws.dbg + (size_t)agent * 16;
      o[8]  = dbg_f1;  // factor(): block assembly | forward sweep | backward rows + K^-1 registers (all factorisations)
      o[9]  = dbg_f2;
      o[10] = dbg_f3;
      o[11] = clock64() - dbg_clk0;
      o[12] = dbg_k1;  // checks: the spill of the row state | the residual pass (the rest of o[4]: tests, certificate)
      o[13] = dbg_k2;  // shader clocks of the whole solve (o[0]: the same span in 10 ns ticks)
      o[0] = wall_clock64() - dbg_t0;
      // set-up split, three 20-bit fields of 10 ns ticks since the start: assembly + CSC done | Ruiz scaling done | rho + K1 done
      // (o[1] - the third: the first factorisation and the load of the row state)
      o[14] = dbg_s1 | (dbg_s2 << 20) | (dbg_s3 << 40);
      o[1] = dbg_setup;
      o[2] = dbg_refac;
      o[3] = dbg_nrefac;
      o[4] = dbg_check;
      o[5] = dbg_ncheck;
      o[6] = iter;
      o[7] = (FAST ? 1 : 0) | (rows_in_lds ? 2 : 0);  // bit 0: register-resident iteration, bit 1: every row array in LDS
    }
  }
#undef gv
#undef gc
#undef g_rho
#undef g_rinv
#undef g_lo
#undef g_hi
#undef g_z
#undef g_y
#undef sv
#undef s_hi
#undef s_zr
#undef s_yr
#undef s_c
  };  // body
  using std::false_type;
  using std::true_type;
  if (rows_in_lds) {
    Rows R;
    carve_rows(R, qp_smem + head, qp_smem + head + rows_hot_bytes(G, S), G, S);
    if (fast)
      body(R, true_type{});
    else
      body(R, false_type{});
  } else {
    char *scr = ws.scratch + (size_t)agent * ws.scratch_stride;
    Rows  R;
    carve_rows(R, fast ? qp_smem + head : scr, scr + rows_hot_bytes(G, S), G, S);
    if (fast)
      body(R, true_type{});
    else
      body(R, false_type{});
  }
}

__global__ __launch_bounds__(QP_NT) void k_qp(SogmPlannerParams pp, SogmQpSettings qs, QpWorkspace ws,
                                            QpConst qc, const double *__restrict__ start_pva,
                                            const double *__restrict__ goal_pv,
                                            const double *__restrict__ polys,
                                            const int32_t *__restrict__ nfaces,
                                            const int32_t *__restrict__ npoly,
                                            double *__restrict__ out_cpts,
                                            int32_t *__restrict__ out_status,
                                            int32_t *__restrict__ out_iters, int ablate_arg, int agent0) {
  qp_solve_agent(pp, qs, ws, qc, start_pva, goal_pv, polys, nfaces, npoly, out_cpts, out_status, out_iters,
                 ablate_arg, blockIdx.x + agent0);
}

// Dataflow kernel Q (sogm_replan): ONE persistent launch; a workgroup takes tickets and solves the agent whose
// corridors became final ticket-th (k_corridor_flow publishes agents in completion order), so a 4000-iteration QP
// only delays its own agent.  The solved agent is handed to the finishing kernel (deconfliction + record packing).
__global__ __launch_bounds__(QP_NT) void k_qp_flow(SogmPlannerParams pp, SogmQpSettings qs, QpWorkspace ws,
                                                 QpConst qc, FlowCtl fc, const double *start_pva,
                                                 const double *goal_pv, const double *polys,
                                                 const int32_t *nfaces, const int32_t *npoly, double *out_cpts,
                                                 int32_t *out_status, int32_t *out_iters, int ablate_arg,
                                                 int n_agents) {
  __shared__ int s_agent;
  if (threadIdx.x == 0) atomicAdd(&fc.hdr[FLOW_Q_RESIDENT], 1);  // this workgroup holds its CU (pre-stamp gate)
  for (;;) {
    if (threadIdx.x == 0) {
      int       a = -1;
      const int k = atomicAdd(&fc.hdr[FLOW_Q_TICKET], 1);
      if (k < n_agents) {
        const long long t0 = wall_clock64();
        while ((a = __hip_atomic_load(fc.q_ready + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0) {
          flow_pause();
          if (__hip_atomic_load(&fc.hdr[FLOW_ERR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
          if (wall_clock64() - t0 > FLOW_TIMEOUT_TICKS) {
            atomicExch(&fc.hdr[FLOW_ERR], 3);
            break;
          }
        }
      }
      s_agent = a;
    }
    __syncthreads();
    const int agent = __builtin_amdgcn_readfirstlane(s_agent);  // uniform: a scalar branch
    if (agent < 0) break;  // no tickets left, or the tick failed
    __threadfence();       // corridor outputs were published before the ready slot
    if (threadIdx.x == 0) fc.ts[agent * 8 + 4] = wall_clock64();
    qp_solve_agent(pp, qs, ws, qc, start_pva, goal_pv, polys, nfaces, npoly, out_cpts, out_status, out_iters,
                   ablate_arg, agent);
    __syncthreads();
    if (threadIdx.x == 0) fc.ts[agent * 8 + 5] = wall_clock64();
    __threadfence();
    if (threadIdx.x == 0) {
      const int r = atomicAdd(&fc.hdr[FLOW_F_READY_N], 1);
      __hip_atomic_store(fc.f_ready + r, agent, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
}

// Flight kernel Q (sogm_flight_run): as k_qp_flow, over the flight's rings — tickets run over every (agent, tick) of the
// flight, the solved agent goes to the finish queue.
__global__ __launch_bounds__(QP_NT) void k_flight_qp(SogmPlannerParams pp, SogmQpSettings qs, QpWorkspace ws, QpConst qc,
                                                   FlightCtl fl, const double *start_pva, const double *goal_pv,
                                                   const double *polys, const int32_t *nfaces, const int32_t *npoly,
                                                   double *out_cpts, int32_t *out_status, int32_t *out_iters) {
  __shared__ int s_agent;
  const int total = fl.n_agents * fl.n_ticks;
  fl_wg_started(fl, 0);
  for (;;) {
    if (threadIdx.x < 64) {  // the first wave fetches the ticket and waits for its item (wave-uniform helpers)
      int       a = -1;
      const int k = flow_ticket(&fl.hdr[FL_Q_TICKET]);
      if (k < total) a = fl_wait_item(fl.q_ring, fl.ring_mask, k, &fl.hdr[FL_ERR]);
      if (threadIdx.x == 0) s_agent = a;
    }
    __syncthreads();
    const int agent = __builtin_amdgcn_readfirstlane(s_agent);
    if (agent < 0) break;  // no tickets left, or the flight failed
    __threadfence();
    if (threadIdx.x == 0) fl.ts[agent * FL_TS + 4] = wall_clock64();
    qp_solve_agent(pp, qs, ws, qc, start_pva, goal_pv, polys, nfaces, npoly, out_cpts, out_status, out_iters, 0, agent);
    __syncthreads();
    if (threadIdx.x == 0) {
      fl.ts[agent * FL_TS + 5] = wall_clock64();
      wq_push(fl.lw, &fl.hdr[FL_LW_TAIL], ((unsigned)WK_FINISH << 28) | (unsigned)agent, 1);
    }
    __syncthreads();
  }
}

// Dynamic LDS k_qp may ask for: the CU's 160 KiB minus the kernel's static LDS (queried, not assumed).
int qp_dynamic_lds_bytes() {
  hipFuncAttributes a, b;
  if (hipFuncGetAttributes(&a, (const void *)k_qp) != hipSuccess) return 128 * 1024;
  if (hipFuncGetAttributes(&b, (const void *)k_qp_flow) != hipSuccess) return 128 * 1024;
  long stat = (long)(a.sharedSizeBytes > b.sharedSizeBytes ? a.sharedSizeBytes : b.sharedSizeBytes);
  if (hipFuncGetAttributes(&b, (const void *)k_flight_qp) == hipSuccess && (long)b.sharedSizeBytes > stat) stat = (long)b.sharedSizeBytes;
  const long dyn  = 160L * 1024 - stat;
  return (int)(dyn & ~255L);
}

int launch_qp(const SogmPlannerParams &pp, const SogmQpSettings &qs, const QpWorkspace &ws,
              const QpConst &qc, int n_agents, const double *start_pva, const double *goal_pv,
              const double *polys, const int32_t *nfaces, const int32_t *npoly, double *out_cpts,
              int32_t *out_status, int32_t *out_iters, hipStream_t st, int agent0) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)k_qp, hipFuncAttributeMaxDynamicSharedMemorySize,
                              ws.dyn_lds_bytes);
    attr_set = true;
  }
  // ws.ablate (profiling aid only, tuning key qp_ablate): bit0 skip A^T w, bit1 skip the solve, bit2/3 skip row updates
  const int ablate = ws.ablate;
  hipLaunchKernelGGL(k_qp, dim3(n_agents), dim3(QP_NT), ws.dyn_lds_bytes, st, pp, qs, ws, qc, start_pva,
                     goal_pv, polys, nfaces, npoly, out_cpts, out_status, out_iters, ablate, agent0);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_qp_flow(const SogmPlannerParams &pp, const SogmQpSettings &qs, const QpWorkspace &ws,
                   const QpConst &qc, const FlowCtl &fc, int n_agents, int n_workgroups, const double *start_pva,
                   const double *goal_pv, const double *polys, const int32_t *nfaces, const int32_t *npoly,
                   double *out_cpts, int32_t *out_status, int32_t *out_iters, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)k_qp_flow, hipFuncAttributeMaxDynamicSharedMemorySize,
                              ws.dyn_lds_bytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_qp_flow, dim3(n_workgroups), dim3(QP_NT), ws.dyn_lds_bytes, st, pp, qs, ws, qc, fc,
                     start_pva, goal_pv, polys, nfaces, npoly, out_cpts, out_status, out_iters, 0, n_agents);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_flight_qp(const SogmPlannerParams &pp, const SogmQpSettings &qs, const QpWorkspace &ws, const QpConst &qc,
                     const FlightCtl &fl, int n_workgroups, const double *start_pva, const double *goal_pv,
                     const double *polys, const int32_t *nfaces, const int32_t *npoly, double *out_cpts,
                     int32_t *out_status, int32_t *out_iters, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)k_flight_qp, hipFuncAttributeMaxDynamicSharedMemorySize, ws.dyn_lds_bytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_flight_qp, dim3(n_workgroups), dim3(QP_NT), ws.dyn_lds_bytes, st, pp, qs, ws, qc, fl, start_pva,
                     goal_pv, polys, nfaces, npoly, out_cpts, out_status, out_iters);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace sogm
