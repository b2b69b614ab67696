// qp_oracle.cpp — CPU restatement of the corridor-constrained min-jerk Bezier QP and its OSQP
// solve.  TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Assembly follows traj_opt/src/bezier_optimizer.cpp:27-260 (setup, calcCtrlPtsCvtMat,
// calcMinJerkCost, addContinuity/Dynamical/SafetyConstraints); the known answers of
// traj_opt/test/test_bezier_opt.cpp (shapes, boundary conditions / continuity within 1e-3) pin it
// in tests/test_qp_oracle.py.
//
// Solver: the reference calls OSQP (github.com/osqp/osqp, v0.6.x C API, version unpinned,
// NOT vendored under /root/reference) through traj_opt/include/iosqp.hpp:40-115 with
// eps_abs = eps_rel = 1e-3 and otherwise default settings (bezier_optimizer.cpp:262-285).
// This file restates the PUBLISHED OSQP algorithm (Stellato et al., "OSQP: an operator splitting
// solver for quadratic programs", Math. Prog. Comp. 2020) with the v0.6 defaults as recalled:
// Ruiz equilibration (10 passes, with cost scaling), rho = 0.1 (x1e3 on equality rows, 1e-6 on
// free rows), sigma = 1e-6, alpha = 1.6, termination checked every 25 iterations on UNSCALED
// residuals, max_iter 4000, no polish.  Adaptive rho follows SogmQpSettings.adaptive_rho_interval
// (0 = one fixed KKT factor; default 25); its estimate is computed from the SCALED residuals and
// norms, as auxil.c::compute_rho_estimate does (it reads the work vectors update_info left behind).  The primal
// infeasibility certificate projects delta_y onto the polar of the recession cone of [l, u] first
// (auxil.c::is_primal_infeasible); the dual one cannot fire (q = 0) and is not restated; max_iter ends with the
// approximate check (solved / primal infeasible inaccurate, status 2 / 3).  The x-update solves the reduced
// system (P + sigma I + A^T diag(rho) A) x = rhs by Cholesky, which is algebraically the quasi-definite KKT
// solve OSQP performs.
// Solution values: parity unpinned (OSQP absent, reference stops at 1e-3).
#include <cmath>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

const double OSQP_INFTY  = 1e30;
const double MIN_SCALING = 1e-04, MAX_SCALING = 1e+04;
const double RHO_MIN = 1e-06, RHO_TOL = 1e-04, RHO_EQ_OVER_RHO_INEQ = 1e03;

typedef std::vector<double> Vec;

// 15x15 per-piece cost  p2j^T [[I/3, I/6],[I/6, I/3]] p2j   (bezier_optimizer.cpp:64-111)
void minJerkBlock(double QM[15][15]) {
  double p2v[12][15] = {{0}}, v2a[9][12] = {{0}}, a2j[6][9] = {{0}};
  for (int i = 0; i < 4; ++i)
    for (int d = 0; d < 3; ++d) {
      p2v[i * 3 + d][i * 3 + d]       = -4;
      p2v[i * 3 + d][(i + 1) * 3 + d] = 4;
    }
  for (int i = 0; i < 3; ++i)
    for (int d = 0; d < 3; ++d) {
      v2a[i * 3 + d][i * 3 + d]       = -3;
      v2a[i * 3 + d][(i + 1) * 3 + d] = 3;
    }
  for (int i = 0; i < 2; ++i)
    for (int d = 0; d < 3; ++d) {
      a2j[i * 3 + d][i * 3 + d]       = -2;
      a2j[i * 3 + d][(i + 1) * 3 + d] = 2;
    }
  double p2a[9][15] = {{0}}, p2j[6][15] = {{0}};
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 15; ++j)
      for (int k = 0; k < 12; ++k) p2a[i][j] += v2a[i][k] * p2v[k][j];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 15; ++j)
      for (int k = 0; k < 9; ++k) p2j[i][j] += a2j[i][k] * p2a[k][j];
  double P[6][6] = {{0}};
  for (int d = 0; d < 3; ++d) {
    P[d][d]         = 1.0 / 3;
    P[d][3 + d]     = 1.0 / 6;
    P[3 + d][d]     = 1.0 / 6;
    P[3 + d][3 + d] = 1.0 / 3;
  }
  double T[6][15] = {{0}};
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 15; ++j)
      for (int k = 0; k < 6; ++k) T[i][j] += P[i][k] * p2j[k][j];
  for (int i = 0; i < 15; ++i)
    for (int j = 0; j < 15; ++j) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += p2j[k][i] * T[k][j];
      QM[i][j] = s;
    }
}

// Dense assembly; returns the number of rows m.
int assemble(const double start[9], const double goal[9], const double *t, int M,
             const double *polys, const int *nfaces, int max_faces, double vmax, double amax,
             Vec &Q, Vec &A, Vec &l, Vec &u) {
  const int n = 15 * M;
  int       nsafe = 0;
  for (int i = 0; i < M; ++i) nsafe += nfaces[i];
  const int m = 5 * nsafe + 9 * (M + 1) + M * 21;
  Q.assign((size_t)n * n, 0.0);
  A.assign((size_t)m * n, 0.0);
  l.assign(m, 0.0);
  u.assign(m, 0.0);
  double QM[15][15];
  minJerkBlock(QM);
  for (int i = 0; i < M; ++i)
    for (int r = 0; r < 15; ++r)
      for (int c = 0; c < 15; ++c) Q[(size_t)(i * 15 + r) * n + i * 15 + c] = QM[r][c];
  auto at = [&](int r, int c) -> double & { return A[(size_t)r * n + c]; };
  int        idx = 0;
  const double t0 = t[0], tM = t[M - 1];
  // ---- continuity (:136-216)
  for (int d = 0; d < 3; ++d) {
    at(idx + d, d) = 1;
    u[idx + d] = l[idx + d] = start[0 * 3 + d];
  }
  idx += 3;
  for (int i = 1; i < M; ++i) {
    for (int d = 0; d < 3; ++d) {
      at(idx + d, i * 15 + d)     = 1;
      at(idx + d, i * 15 - 3 + d) = -1;
    }
    idx += 3;
  }
  for (int d = 0; d < 3; ++d) {
    at(idx + d, M * 15 - 3 + d) = 1;
    u[idx + d] = l[idx + d] = goal[0 * 3 + d];
  }
  idx += 3;
  for (int d = 0; d < 3; ++d) {
    at(idx + d, d)     = -4;
    at(idx + d, 3 + d) = 4;
    u[idx + d] = l[idx + d] = start[1 * 3 + d] * t0;
  }
  idx += 3;
  for (int i = 1; i < M; ++i) {
    const double t1 = t[i], t1_ = t[i - 1];
    for (int d = 0; d < 3; ++d) {
      at(idx + d, i * 15 + d)     = -4.0 / t1;
      at(idx + d, i * 15 + 3 + d) = 4.0 / t1;
      at(idx + d, i * 15 - 3 + d) = -4.0 / t1_;
      at(idx + d, i * 15 - 6 + d) = 4.0 / t1_;
    }
    idx += 3;
  }
  for (int d = 0; d < 3; ++d) {
    at(idx + d, M * 15 - 6 + d) = -4;
    at(idx + d, M * 15 - 3 + d) = 4;
    u[idx + d] = l[idx + d] = goal[1 * 3 + d] * tM;
  }
  idx += 3;
  const double p2a[3] = {12, -24, 12};  // (v2a * p2v).block<3,9>(0,0) = 12 [I, -2I, I]
  for (int d = 0; d < 3; ++d) {
    for (int k = 0; k < 3; ++k) at(idx + d, k * 3 + d) = p2a[k];
    u[idx + d] = l[idx + d] = start[2 * 3 + d] * t0 * t0;
  }
  idx += 3;
  for (int i = 1; i < M; ++i) {
    const double t2 = std::pow(t[i], 2), t2_ = std::pow(t[i - 1], 2);
    for (int d = 0; d < 3; ++d)
      for (int k = 0; k < 3; ++k) {
        at(idx + d, i * 15 + k * 3 + d)     = p2a[k] / t2;
        at(idx + d, i * 15 - 9 + k * 3 + d) = -p2a[k] / t2_;
      }
    idx += 3;
  }
  for (int d = 0; d < 3; ++d) {
    for (int k = 0; k < 3; ++k) at(idx + d, M * 15 - 9 + k * 3 + d) = p2a[k];
    u[idx + d] = l[idx + d] = goal[2 * 3 + d] * tM * tM;
  }
  idx += 3;
  // ---- dynamical (:218-246)
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < 4; ++j) {
      for (int d = 0; d < 3; ++d) {
        at(idx + d, i * 15 + j * 3 + d)     = -4;
        at(idx + d, i * 15 + j * 3 + 3 + d) = 4;
        u[idx + d]                          = vmax * 1.0 * t[i];
        l[idx + d]                          = -vmax * 1.0 * t[i];
      }
      idx += 3;
    }
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < 3; ++j) {
      for (int d = 0; d < 3; ++d) {
        for (int k = 0; k < 3; ++k) at(idx + d, i * 15 + j * 3 + k * 3 + d) = p2a[k];
        u[idx + d] = amax * 1.0 * t[i] * t[i];
        l[idx + d] = -amax * 1.0 * t[i] * t[i];
      }
      idx += 3;
    }
  // ---- safety (:248-265)
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < nfaces[i]; ++j) {
      const double *h = polys + ((size_t)i * max_faces + j) * 4;
      for (int k = 0; k < 5; ++k) {
        for (int d = 0; d < 3; ++d) at(idx, i * 15 + k * 3 + d) = h[d];
        u[idx] = -h[3];
        l[idx] = -OSQP_INFTY;
        ++idx;
      }
    }
  return m;
}

double ninf(const Vec &v) {
  double m = 0;
  for (double x : v) m = std::max(m, std::fabs(x));
  return m;
}
void limit_scaling(double &v) {
  v = v < MIN_SCALING ? 1.0 : v;
  v = v > MAX_SCALING ? MAX_SCALING : v;
}

// OSQP algorithm on dense data.  P full symmetric n x n, A m x n.
int osqpDense(const double *Pin, const double *qin, const double *Ain, const double *lin,
              const double *uin, int n, int m, const SogmQpSettings *qs, double *xout,
              double *yout, int *iters_out) {
  Vec P(Pin, Pin + (size_t)n * n), A(Ain, Ain + (size_t)m * n), q(qin, qin + n), l(lin, lin + m),
      u(uin, uin + m);
  // ---- Ruiz equilibration with cost scaling
  Vec    D(n, 1.0), E(m, 1.0), Dt(n), Et(m);
  double c = 1.0;
  for (int it = 0; it < qs->scaling_iters; ++it) {
    for (int j = 0; j < n; ++j) {
      double mx = 0;
      for (int i = 0; i < n; ++i) mx = std::max(mx, std::fabs(P[(size_t)i * n + j]));
      for (int i = 0; i < m; ++i) mx = std::max(mx, std::fabs(A[(size_t)i * n + j]));
      Dt[j] = mx;
    }
    for (int i = 0; i < m; ++i) {
      double mx = 0;
      for (int j = 0; j < n; ++j) mx = std::max(mx, std::fabs(A[(size_t)i * n + j]));
      Et[i] = mx;
    }
    for (int j = 0; j < n; ++j) {
      limit_scaling(Dt[j]);
      Dt[j] = 1.0 / std::sqrt(Dt[j]);
    }
    for (int i = 0; i < m; ++i) {
      limit_scaling(Et[i]);
      Et[i] = 1.0 / std::sqrt(Et[i]);
    }
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) P[(size_t)i * n + j] *= Dt[i] * Dt[j];
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < n; ++j) A[(size_t)i * n + j] *= Et[i] * Dt[j];
    for (int j = 0; j < n; ++j) {
      q[j] *= Dt[j];
      D[j] *= Dt[j];
    }
    for (int i = 0; i < m; ++i) E[i] *= Et[i];
    double mean = 0;
    for (int j = 0; j < n; ++j) {
      double mx = 0;
      for (int i = 0; i < n; ++i) mx = std::max(mx, std::fabs(P[(size_t)i * n + j]));
      mean += mx;
    }
    double c_temp = mean / n;
    double nq     = ninf(q);
    limit_scaling(nq);
    c_temp = std::max(c_temp, nq);
    limit_scaling(c_temp);
    c_temp = 1.0 / c_temp;
    for (auto &v : P) v *= c_temp;
    for (auto &v : q) v *= c_temp;
    c *= c_temp;
  }
  for (int i = 0; i < m; ++i) {
    l[i] *= E[i];
    u[i] *= E[i];
  }
  const double cinv = 1.0 / c;
  // ---- rho vector
  Vec rho(m);
  for (int i = 0; i < m; ++i) {
    if (l[i] < -OSQP_INFTY * MIN_SCALING && u[i] > OSQP_INFTY * MIN_SCALING)
      rho[i] = RHO_MIN;
    else if (u[i] - l[i] < RHO_TOL)
      rho[i] = RHO_EQ_OVER_RHO_INEQ * qs->rho;
    else
      rho[i] = qs->rho;
  }
  // ---- K = P + sigma I + A^T diag(rho) A, Cholesky K = G G^T (lower)
  Vec    K((size_t)n * n);
  double rho_cur = qs->rho;
  bool   chol_ok = true;
  auto factor = [&]() {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = P[(size_t)i * n + j];
      if (i == j) s += qs->sigma;
      for (int r = 0; r < m; ++r) s += A[(size_t)r * n + i] * rho[r] * A[(size_t)r * n + j];
      K[(size_t)i * n + j] = s;
    }
  for (int j = 0; j < n; ++j) {
    double d = K[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= K[(size_t)j * n + k] * K[(size_t)j * n + k];
    if (!(d > 0)) {
      chol_ok = false;  // OSQP_NON_CVX-like failure
      return;
    }
    d                    = std::sqrt(d);
    K[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = K[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) s -= K[(size_t)i * n + k] * K[(size_t)j * n + k];
      K[(size_t)i * n + j] = s / d;
    }
  }
  };
  factor();
  if (!chol_ok) return -7;
  auto solveK = [&](Vec &b) {
    for (int i = 0; i < n; ++i) {
      double s = b[i];
      for (int k = 0; k < i; ++k) s -= K[(size_t)i * n + k] * b[k];
      b[i] = s / K[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = b[i];
      for (int k = i + 1; k < n; ++k) s -= K[(size_t)k * n + i] * b[k];
      b[i] = s / K[(size_t)i * n + i];
    }
  };
  Vec x(n, 0.0), z(m, 0.0), y(m, 0.0), xp(n), zp(m), xt(n), zt(m), dx(n), dy(m);
  Vec Ax(m), Px(n), Aty(n);
  const double alpha = qs->alpha;
  int          status = -2, iter = 0;
  // scaled residual norms of the last evaluation: what compute_rho_estimate reads (the work vectors z_prev / x_prev
  // still hold Ax - z and Px + q + A'y in SCALED form after update_info; OSQP 1.0 names them scaled_prim_res /
  // scaled_dual_res), with the scaled norms of z, Ax, q, A'y, Px
  double sc_pr = 0, sc_dr = 0, sc_nAx = 0, sc_nz = 0, sc_nPx = 0, sc_nAty = 0, sc_nq = 0;
  // OSQP unscales its residual norms by MULTIPLYING with the stored reciprocals of the scaling (scaling.c: Dinv, Einv =
  // vec_ew_recipr(D), (E) once; auxil.c compute_pri_res / compute_dua_res / is_primal_infeasible: vec_scaled_norm_inf(Einv, .),
  // (Dinv, .)).  Until round 6 this restatement divided by E and D: the same number up to the last bit, not the same bits.
  Vec Einv(m), Dinv(n);
  for (int i = 0; i < m; ++i) Einv[i] = 1.0 / E[i];
  for (int j = 0; j < n; ++j) Dinv[j] = 1.0 / D[j];
  auto residuals = [&](double eps_abs, double eps_rel, bool &prim_ok, bool &dual_ok) {
    double pr = 0, nAx = 0, nz = 0;
    sc_pr = sc_nAx = sc_nz = 0;
    for (int i = 0; i < m; ++i) {
      double s = 0;
      for (int j = 0; j < n; ++j) s += A[(size_t)i * n + j] * x[j];
      Ax[i] = s;
      pr    = std::max(pr, std::fabs(Einv[i] * (s - z[i])));
      nAx   = std::max(nAx, std::fabs(Einv[i] * s));
      nz    = std::max(nz, std::fabs(Einv[i] * z[i]));
      sc_pr  = std::max(sc_pr, std::fabs(s - z[i]));
      sc_nAx = std::max(sc_nAx, std::fabs(s));
      sc_nz  = std::max(sc_nz, std::fabs(z[i]));
    }
    double dr = 0, nPx = 0, nAty = 0, nq = 0;
    sc_dr = sc_nPx = sc_nAty = sc_nq = 0;
    for (int j = 0; j < n; ++j) {
      double s = 0, a = 0;
      for (int k = 0; k < n; ++k) s += P[(size_t)j * n + k] * x[k];
      for (int i = 0; i < m; ++i) a += A[(size_t)i * n + j] * y[i];
      Px[j]  = s;
      Aty[j] = a;
      dr     = std::max(dr, std::fabs(Dinv[j] * (s + q[j] + a)));
      nPx    = std::max(nPx, std::fabs(Dinv[j] * s));
      nAty   = std::max(nAty, std::fabs(Dinv[j] * a));
      nq     = std::max(nq, std::fabs(Dinv[j] * q[j]));
      sc_dr   = std::max(sc_dr, std::fabs(s + q[j] + a));
      sc_nPx  = std::max(sc_nPx, std::fabs(s));
      sc_nAty = std::max(sc_nAty, std::fabs(a));
      sc_nq   = std::max(sc_nq, std::fabs(q[j]));
    }
    dr *= cinv;
    const double eps_prim = eps_abs + eps_rel * std::max(nAx, nz);
    const double eps_dual = eps_abs + eps_rel * cinv * std::max(std::max(nPx, nAty), nq);
    prim_ok               = pr < eps_prim;
    dual_ok               = dr < eps_dual;
  };
  // primal infeasibility certificate (OSQP v0.6 auxil.c is_primal_infeasible; check_termination calls it only
  // when the primal residual test failed).  delta_y is first PROJECTED onto the polar of the recession cone of
  // [l, u]: a component whose upper bound is infinite can only count when negative, one whose lower bound is
  // infinite only when positive (every corridor face is such a row, l = -OSQP_INFTY), a free row not at all.
  // The norm, u'(dy)+ + l'(dy)- and A'dy all use the projected vector; ||dy|| and A'dy are unscaled (E dy,
  // Dinv A'dy); both tests are relative to ||dy||.  (dy is overwritten, as work->delta_y is; the next iteration
  // recomputes it.)
  auto primal_infeasible = [&](double eps_inf) -> bool {
    for (int i = 0; i < m; ++i) {
      if (u[i] > OSQP_INFTY * MIN_SCALING) {
        if (l[i] < -OSQP_INFTY * MIN_SCALING) dy[i] = 0.0;
        else dy[i] = std::min(dy[i], 0.0);
      } else if (l[i] < -OSQP_INFTY * MIN_SCALING) {
        dy[i] = std::max(dy[i], 0.0);
      }
    }
    double ndy = 0;
    for (int i = 0; i < m; ++i) ndy = std::max(ndy, std::fabs(E[i] * dy[i]));
    if (!(ndy > eps_inf)) return false;
    double lhs = 0;
    for (int i = 0; i < m; ++i) lhs += u[i] * std::max(dy[i], 0.0) + l[i] * std::min(dy[i], 0.0);
    if (!(lhs < -eps_inf * ndy)) return false;
    double na = 0;
    for (int j = 0; j < n; ++j) {
      double s = 0;
      for (int i = 0; i < m; ++i) s += A[(size_t)i * n + j] * dy[i];
      na = std::max(na, std::fabs(Dinv[j] * s));
    }
    return na < eps_inf * ndy;
  };
  for (iter = 1; iter <= qs->max_iter; ++iter) {
    xp = x;
    zp = z;
    // x-tilde
    for (int j = 0; j < n; ++j) {
      double s = qs->sigma * xp[j] - q[j];
      for (int i = 0; i < m; ++i) s += A[(size_t)i * n + j] * (rho[i] * zp[i] - y[i]);
      xt[j] = s;
    }
    solveK(xt);
    for (int i = 0; i < m; ++i) {
      double s = 0;
      for (int j = 0; j < n; ++j) s += A[(size_t)i * n + j] * xt[j];
      zt[i] = s;
    }
    for (int j = 0; j < n; ++j) {
      x[j]  = alpha * xt[j] + (1.0 - alpha) * xp[j];
      dx[j] = x[j] - xp[j];
    }
    for (int i = 0; i < m; ++i) {
      const double zr = alpha * zt[i] + (1.0 - alpha) * zp[i];
      double       v  = zr + (1.0 / rho[i]) * y[i];  // OSQP update_z: rho_inv_vec[i] * y[i] (auxil.c)
      v               = v < l[i] ? l[i] : (v > u[i] ? u[i] : v);
      z[i]            = v;
      dy[i]           = rho[i] * (zr - z[i]);
      y[i] += dy[i];
    }
    const bool do_check = qs->check_termination > 0 && iter % qs->check_termination == 0;
    const bool do_adapt = qs->adaptive_rho_interval > 0 && iter % qs->adaptive_rho_interval == 0;
    bool       p_ok = false, d_ok = false;
    if (do_check || do_adapt) residuals(qs->eps_abs, qs->eps_rel, p_ok, d_ok);
    if (do_check) {
      if (p_ok && d_ok) {
        status = 1;
        break;
      }
      if (!p_ok && primal_infeasible(1e-4)) {  // eps_prim_inf default
        status = -3;
        break;
      }
    }
    // adaptive rho (OSQP adapt_rho / compute_rho_estimate) after the termination test, on the same residual
    // evaluation (osqp.c: update_info runs once per iteration).  The estimate uses the SCALED residuals and norms:
    // compute_rho_estimate reads vec_norm_inf(work->z_prev) / (work->x_prev) — the scaled Ax - z and Px + q + A'y that
    // compute_pri_res / compute_dua_res left there — and the plain norms of z, Ax, q, A'y, Px.
    if (do_adapt) {
      const double pr_n = sc_pr / (std::max(sc_nz, sc_nAx) + 1e-10);
      const double du_n = sc_dr / (std::max(std::max(sc_nq, sc_nAty), sc_nPx) + 1e-10);
      double       rho_new = rho_cur * std::sqrt(pr_n / (du_n + 1e-10));
      rho_new              = std::min(std::max(rho_new, RHO_MIN), 1e6);
      if (rho_new > rho_cur * 5.0 || rho_new < rho_cur / 5.0) {
        rho_cur = rho_new;
        for (int i = 0; i < m; ++i) {
          if (l[i] < -OSQP_INFTY * MIN_SCALING && u[i] > OSQP_INFTY * MIN_SCALING)
            rho[i] = RHO_MIN;
          else if (u[i] - l[i] < RHO_TOL)
            rho[i] = RHO_EQ_OVER_RHO_INEQ * rho_cur;
          else
            rho[i] = rho_cur;
        }
        factor();
        if (!chol_ok) return -7;
      }
    }
  }
  if (iter > qs->max_iter) {
    iter = qs->max_iter;
    bool p_ok, d_ok;
    // osqp_solve's epilogue: check_termination(work, approximate = 1) — every tolerance x 10 — then MAX_ITER_REACHED
    residuals(qs->eps_abs * 10, qs->eps_rel * 10, p_ok, d_ok);
    if (p_ok && d_ok) status = 2;                                // OSQP_SOLVED_INACCURATE
    else if (!p_ok && primal_infeasible(1e-4 * 10)) status = 3;  // OSQP_PRIMAL_INFEASIBLE_INACCURATE
    else status = -2;                                            // OSQP_MAX_ITER_REACHED
  }
  for (int j = 0; j < n; ++j) xout[j] = D[j] * x[j];
  if (yout)
    for (int i = 0; i < m; ++i) yout[i] = cinv * E[i] * y[i];
  if (iters_out) *iters_out = iter;
  return status;
}

}  // namespace

extern "C" {

int orc_qp_assemble(const double start[9], const double goal[9], const double *t_alloc, int M,
                    const double *polys, const int *nfaces, int max_faces, double vmax,
                    double amax, double *Qo, double *Ao, double *lo, double *uo, int m_cap) {
  Vec       Q, A, l, u;
  const int m = assemble(start, goal, t_alloc, M, polys, nfaces, max_faces, vmax, amax, Q, A, l, u);
  const int n = 15 * M;
  if (Qo) std::memcpy(Qo, Q.data(), sizeof(double) * (size_t)n * n);
  if (m <= m_cap) {
    if (Ao) std::memcpy(Ao, A.data(), sizeof(double) * (size_t)m * n);
    if (lo) std::memcpy(lo, l.data(), sizeof(double) * m);
    if (uo) std::memcpy(uo, u.data(), sizeof(double) * m);
  }
  return m;
}

int orc_qp_solve(const double start[9], const double goal[9], const double *t_alloc, int M,
                 const double *polys, const int *nfaces, int max_faces, double vmax, double amax,
                 const SogmQpSettings *qs, double *x_out, int *iters_out) {
  Vec       Q, A, l, u;
  const int m = assemble(start, goal, t_alloc, M, polys, nfaces, max_faces, vmax, amax, Q, A, l, u);
  const int n = 15 * M;
  Vec       q(n, 0.0);
  return osqpDense(Q.data(), q.data(), A.data(), l.data(), u.data(), n, m, qs, x_out, nullptr,
                   iters_out);
}

int orc_osqp_dense(const double *P, const double *q, const double *A, const double *l,
                   const double *u, int n, int m, const SogmQpSettings *qs, double *x, double *y,
                   int *iters_out) {
  return osqpDense(P, q, A, l, u, n, m, qs, x, y, iters_out);
}

}  // extern "C"
