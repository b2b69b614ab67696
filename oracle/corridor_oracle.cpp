// corridor_oracle.cpp — CPU restatement of safe-corridor generation.  TEST INFRASTRUCTURE ONLY.
//
// Follows:
//   plan_manager/include/sfc_gen/firi.hpp:44-365      chol3d, smoothedL1, costMVIE,
//                                                      maxVolInsEllipsoid, firi
//   plan_manager/include/sfc_gen/lbfgs.hpp:276-600     Lewis-Overton line search, L-BFGS loop
//   plan_manager/src/baseline_fake.cpp:122-221,300-412 (fake planner) and
//   plan_manager/src/baseline.cpp:127-228,296-403      corridor stage of replan()
// LP calls go to lp_oracle.cpp (Seidel).  The 3x3 SVD (Eigen::JacobiSVD in the reference,
// firi.hpp:214-217) is a cyclic Jacobi eigen-solve of L L^T written here; singular vectors are
// unique only up to sign/order for distinct singular values, which leaves the ellipsoid
// (R diag(r)) and therefore the polytope unchanged.
// Parity unpinned: no reference test covers FIRI / L-BFGS / the corridor stage.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "../include/sogm_detmath.h"
#include "oracle.h"

extern int orc_g_use_libm;

double orc_linprog3(const double *c, const double *A, const double *b, int m, double *x);
double orc_linprog4(const double *c, const double *A, const double *b, int m, double *x);

namespace {

typedef double M3[3][3];

// firi.hpp:44-55
void chol3d(const M3 A, M3 L) {
  L[0][0] = std::sqrt(A[0][0]);
  L[0][1] = 0.0;
  L[0][2] = 0.0;
  L[1][0] = 0.5 * (A[0][1] + A[1][0]) / L[0][0];
  L[1][1] = std::sqrt(A[1][1] - L[1][0] * L[1][0]);
  L[1][2] = 0.0;
  L[2][0] = 0.5 * (A[0][2] + A[2][0]) / L[0][0];
  L[2][1] = (0.5 * (A[1][2] + A[2][1]) - L[2][0] * L[1][0]) / L[1][1];
  L[2][2] = std::sqrt(A[2][2] - L[2][0] * L[2][0] - L[2][1] * L[2][1]);
}

// firi.hpp:57-72
bool smoothedL1(double mu, double x, double &f, double &df) {
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
  int                 M;
  double              smoothEps, penaltyWt;
  std::vector<double> A;  // M x 3 row-major
};

// firi.hpp:74-140
double costMVIE(const MvieData &D, const double x[9], double g[9]) {
  const double *p = x, *rtd = x + 3, *cde = x + 6;
  double       *gdp = g, *gdrtd = g + 3, *gdcde = g + 6;
  M3 L;
  L[0][0] = rtd[0] * rtd[0] + DBL_EPSILON;
  L[0][1] = 0.0;
  L[0][2] = 0.0;
  L[1][0] = cde[0];
  L[1][1] = rtd[1] * rtd[1] + DBL_EPSILON;
  L[1][2] = 0.0;
  L[2][0] = cde[2];
  L[2][1] = cde[1];
  L[2][2] = rtd[2] * rtd[2] + DBL_EPSILON;
  // firi.hpp:93-122: one running sum per quantity, faces in index order
  double cost = 0;
  for (int j = 0; j < 3; ++j) gdp[j] = gdrtd[j] = gdcde[j] = 0.0;
  for (int i = 0; i < D.M; ++i) {
    const double *a = &D.A[(size_t)i * 3];
    double        AL[3];
    for (int j = 0; j < 3; ++j) AL[j] = (a[0] * L[0][j] + a[1] * L[1][j]) + a[2] * L[2][j];
    const double normAL = std::sqrt((AL[0] * AL[0] + AL[1] * AL[1]) + AL[2] * AL[2]);
    const double adj[3] = {AL[0] / normAL, AL[1] / normAL, AL[2] / normAL};
    const double Ap     = (a[0] * p[0] + a[1] * p[1]) + a[2] * p[2];
    const double viola  = (normAL + Ap) - 1.0;
    double       c, dc;
    if (smoothedL1(D.smoothEps, viola, c, dc)) {
      cost += c;
      const double vec[3] = {dc * a[0], dc * a[1], dc * a[2]};
      for (int j = 0; j < 3; ++j) gdp[j] += vec[j];
      for (int j = 0; j < 3; ++j) gdrtd[j] += adj[j] * vec[j];
      gdcde[0] += adj[0] * vec[1];
      gdcde[1] += adj[1] * vec[2];
      gdcde[2] += adj[0] * vec[2];
    }
  }
  cost *= D.penaltyWt;
  for (int j = 0; j < 3; ++j) {
    gdp[j] *= D.penaltyWt;
    gdrtd[j] *= D.penaltyWt;
    gdcde[j] *= D.penaltyWt;
  }
  // log via include/sogm_detmath.h (<= 2 ulp from libm) so the HIP side can match bit for bit
  if (orc_g_use_libm)
    cost -= std::log(L[0][0]) + std::log(L[1][1]) + std::log(L[2][2]);
  else
    cost -= sogm_det::log(L[0][0]) + sogm_det::log(L[1][1]) + sogm_det::log(L[2][2]);
  gdrtd[0] -= 1.0 / L[0][0];
  gdrtd[1] -= 1.0 / L[1][1];
  gdrtd[2] -= 1.0 / L[2][2];
  gdrtd[0] *= 2.0 * rtd[0];
  gdrtd[1] *= 2.0 * rtd[1];
  gdrtd[2] *= 2.0 * rtd[2];
  return cost;
}

double dotn(const double *a, const double *b, int n) {
  double s = 0;
  for (int i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

// lbfgs.hpp line_search_lewisoverton (weak Wolfe, bisection / doubling)
int lineSearchLO(const MvieData &D, double x[9], double &f, double g[9], double &stp,
                 const double s[9], const double xp[9], const double gp[9], double stpmin,
                 double stpmax) {
  const double f_dec = 1.0e-4, s_curv = 0.9, machine_prec = 1.0e-16;
  const int    max_linesearch = 64;
  int          count = 0;
  bool         brackt = false, touched = false;
  double       mu = 0.0, nu = stpmax;
  if (!(stp > 0.0)) return -1;
  const double dginit = dotn(gp, s, 9);
  if (0.0 < dginit) return -2;
  const double finit = f, dgtest = f_dec * dginit, dstest = s_curv * dginit;
  while (true) {
    for (int i = 0; i < 9; ++i) x[i] = xp[i] + stp * s[i];
    f = costMVIE(D, x, g);
    ++count;
    if (std::isinf(f) || std::isnan(f)) return -3;
    if (f > finit + stp * dgtest) {
      nu     = stp;
      brackt = true;
    } else {
      if (dotn(g, s, 9) < dstest)
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

// lbfgs_parameter_t::max_iterations (lbfgs.hpp:68-74; firi.hpp leaves it at its default 0 = unlimited).  A TEST knob
// (orc_lbfgs_set_max_iterations): with the iteration count capped, two readings of the optimiser can be compared after
// 1, 2, 5 ... iterations, before rounding differences in the cost function have been amplified by the line searches.
int g_lbfgs_max_iterations = 0;

// lbfgs.hpp lbfgs_optimize with the parameters of firi.hpp:191-199
int lbfgsMVIE(const MvieData &D, double x[9], double &fout) {
  const int    n = 9, m = 18, past = 3;
  const double g_epsilon = 0.0, delta = 1.0e-7, min_step = 1.0e-32, max_step = 1.0e+20,
               cautious = 1.0e-6;
  double xp[9], g[9], gp[9], d[9], pf[3];
  double lm_alpha[18], lm_s[18][9], lm_y[18][9], lm_ys[18];
  std::memset(lm_alpha, 0, sizeof(lm_alpha));
  std::memset(lm_s, 0, sizeof(lm_s));
  std::memset(lm_y, 0, sizeof(lm_y));
  std::memset(lm_ys, 0, sizeof(lm_ys));
  double fx = costMVIE(D, x, g);
  pf[0]     = fx;
  for (int i = 0; i < n; ++i) d[i] = -g[i];
  auto ninf = [](const double *v, int k) {
    double mx = 0;
    for (int i = 0; i < k; ++i) mx = std::max(mx, std::fabs(v[i]));
    return mx;
  };
  int ret;
  if (ninf(g, n) / std::max(1.0, ninf(x, n)) < g_epsilon) {
    ret = 0;
  } else {
    double step = 1.0 / std::sqrt(dotn(d, d, n));
    int    k = 1, end = 0, bound = 0;
    while (true) {
      for (int i = 0; i < n; ++i) {
        xp[i] = x[i];
        gp[i] = g[i];
      }
      const int ls = lineSearchLO(D, x, fx, g, step, d, xp, gp, min_step, max_step);
      if (ls < 0) {
        for (int i = 0; i < n; ++i) {
          x[i] = xp[i];
          g[i] = gp[i];
        }
        ret = ls;
        break;
      }
      if (ninf(g, n) / std::max(1.0, ninf(x, n)) < g_epsilon) {
        ret = 0;
        break;
      }
      if (past <= k) {
        const double rate = std::fabs(pf[k % past] - fx) / std::max(1.0, std::fabs(fx));
        if (rate < delta) {
          ret = 1;  // LBFGS_STOP
          break;
        }
      }
      pf[k % past] = fx;
      if (g_lbfgs_max_iterations != 0 && g_lbfgs_max_iterations <= k) {
        ret = -8;  // LBFGSERR_MAXIMUMITERATION (lbfgs.hpp:597-602)
        break;
      }
      ++k;
      for (int i = 0; i < n; ++i) {
        lm_s[end][i] = x[i] - xp[i];
        lm_y[end][i] = g[i] - gp[i];
      }
      const double ys = dotn(lm_y[end], lm_s[end], n);
      const double yy = dotn(lm_y[end], lm_y[end], n);
      lm_ys[end]      = ys;
      for (int i = 0; i < n; ++i) d[i] = -g[i];
      const double cau = dotn(lm_s[end], lm_s[end], n) * std::sqrt(dotn(gp, gp, n)) * cautious;
      if (ys > cau) {
        ++bound;
        bound = m < bound ? m : bound;
        end   = (end + 1) % m;
        int j = end;
        for (int i = 0; i < bound; ++i) {
          j           = (j + m - 1) % m;
          lm_alpha[j] = dotn(lm_s[j], d, n) / lm_ys[j];
          for (int q = 0; q < n; ++q) d[q] += (-lm_alpha[j]) * lm_y[j][q];
        }
        for (int q = 0; q < n; ++q) d[q] *= ys / yy;
        for (int i = 0; i < bound; ++i) {
          const double beta = dotn(lm_y[j], d, n) / lm_ys[j];
          for (int q = 0; q < n; ++q) d[q] += (lm_alpha[j] - beta) * lm_s[j][q];
          j = (j + 1) % m;
        }
      }
      step = 1.0;
    }
  }
  fout = fx;
  return ret;
}

// eigen-decomposition of a symmetric 3x3 by cyclic Jacobi; V columns = eigenvectors
void jacobiEig3(M3 S, M3 V, double w[3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 32; ++sweep) {
    const double off = std::fabs(S[0][1]) + std::fabs(S[0][2]) + std::fabs(S[1][2]);
    const double dia = std::fabs(S[0][0]) + std::fabs(S[1][1]) + std::fabs(S[2][2]);
    if (off <= 1e-300 || off <= 1e-17 * dia) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (S[p][q] == 0.0) continue;
        const double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {  // S <- S J
          const double skp = S[k][p], skq = S[k][q];
          S[k][p]          = c * skp - s * skq;
          S[k][q]          = s * skp + c * skq;
        }
        for (int k = 0; k < 3; ++k) {  // S <- J^T S
          const double spk = S[p][k], sqk = S[q][k];
          S[p][k]          = c * spk - s * sqk;
          S[q][k]          = s * spk + c * sqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p]          = c * vkp - s * vkq;
          V[k][q]          = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < 3; ++i) w[i] = S[i][i];
}

// firi.hpp:146-236
bool maxVolInsEllipsoid(const double *hPoly, int M, M3 R, double p[3], double r[3]) {
  std::vector<double> Alp((size_t)M * 4), blp(M), hNorm(M);
  for (int i = 0; i < M; ++i) {
    const double *h = hPoly + (size_t)i * 4;
    hNorm[i]        = std::sqrt((h[0] * h[0] + h[1] * h[1]) + h[2] * h[2]);
    for (int j = 0; j < 3; ++j) Alp[(size_t)i * 4 + j] = h[j] / hNorm[i];
    Alp[(size_t)i * 4 + 3] = 1.0;
    blp[i]                 = -h[3] / hNorm[i];
  }
  const double clp[4] = {0, 0, 0, -1.0};
  double       xlp[4];
  const double maxdepth = -orc_linprog4(clp, Alp.data(), blp.data(), M, xlp);
  if (!(maxdepth > 0.0) || std::isinf(maxdepth)) return false;
  const double interior[3] = {xlp[0], xlp[1], xlp[2]};

  MvieData D;
  D.M = M;
  D.A.resize((size_t)M * 3);
  for (int i = 0; i < M; ++i) {
    const double *a   = &Alp[(size_t)i * 4];
    const double  den = blp[i] - ((a[0] * interior[0] + a[1] * interior[1]) + a[2] * interior[2]);
    for (int j = 0; j < 3; ++j) D.A[(size_t)i * 3 + j] = a[j] / den;
  }
  double x[9];
  M3     Q, L;
  // Q = R diag(r^2) R^T
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      Q[i][j] = (R[i][0] * (r[0] * r[0]) * R[j][0] + R[i][1] * (r[1] * r[1]) * R[j][1]) +
                R[i][2] * (r[2] * r[2]) * R[j][2];
  chol3d(Q, L);
  for (int j = 0; j < 3; ++j) x[j] = p[j] - interior[j];
  x[3]        = std::sqrt(L[0][0]);
  x[4]        = std::sqrt(L[1][1]);
  x[5]        = std::sqrt(L[2][2]);
  x[6]        = L[1][0];
  x[7]        = L[2][1];
  x[8]        = L[2][0];
  D.smoothEps = 1.0e-2;
  D.penaltyWt = 1.0e+3;
  double    minCost;
  const int ret = lbfgsMVIE(D, x, minCost);

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
  // SVD of L: U from the eigenvectors of L L^T, singular values sqrt(eigenvalues), descending
  M3 S, V;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      S[i][j] = (L[i][0] * L[j][0] + L[i][1] * L[j][1]) + L[i][2] * L[j][2];
  double w[3];
  jacobiEig3(S, V, w);
  int ord[3] = {0, 1, 2};
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2 - a; ++b)
      if (w[ord[b]] < w[ord[b + 1]]) {
        const int t = ord[b];
        ord[b]      = ord[b + 1];
        ord[b + 1]  = t;
      }
  M3     U;
  double Sg[3];
  for (int c = 0; c < 3; ++c) {
    Sg[c] = std::sqrt(w[ord[c]] > 0 ? w[ord[c]] : 0.0);
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
  return ret >= 0;
}

inline double dot3(const double *a, const double *b) {
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

// firi.hpp:238-365.  Returns the number of faces, or -1 when a seed lies outside bd (hPoly is then
// left untouched == empty in the caller).
int firi(const double *bd, int M, const double *pc, int N, const double a[3], const double b[3],
         int iterations, double *hPoly, int max_faces, double r[3]) {
  const double epsilon = 1.0e-6;
  for (int i = 0; i < M; ++i) {
    const double *h = bd + (size_t)i * 4;
    if (dot3(h, a) + h[3] > 0.0 || dot3(h, b) + h[3] > 0.0) return -1;
  }
  M3     R = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  double p[3] = {0.5 * (a[0] + b[0]), 0.5 * (a[1] + b[1]), 0.5 * (a[2] + b[2])};
  std::vector<double> forwardH((size_t)(M + N) * 4), forwardB((size_t)M * 3), forwardD(M),
      forwardPC((size_t)N * 3), distDs(M), tangents((size_t)N * 4), distRs(N);
  std::vector<unsigned char> bdFlags(M), pcFlags(N);
  std::vector<double>        poly;
  int                        nH = 0;
  for (int loop = 0; loop < iterations; ++loop) {
    M3 forward, backward;
    for (int k = 0; k < 3; ++k)
      for (int j = 0; j < 3; ++j) {
        forward[k][j]  = (1.0 / r[k]) * R[j][k];  // diag(1/r) R^T
        backward[k][j] = R[k][j] * r[j];          // R diag(r)
      }
    for (int i = 0; i < M; ++i) {
      const double *h = bd + (size_t)i * 4;
      for (int j = 0; j < 3; ++j)
        forwardB[(size_t)i * 3 + j] =
            (h[0] * backward[0][j] + h[1] * backward[1][j]) + h[2] * backward[2][j];
      forwardD[i] = h[3] + dot3(h, p);
    }
    for (int i = 0; i < N; ++i) {
      const double d[3] = {pc[(size_t)i * 3] - p[0], pc[(size_t)i * 3 + 1] - p[1],
                           pc[(size_t)i * 3 + 2] - p[2]};
      for (int k = 0; k < 3; ++k) forwardPC[(size_t)i * 3 + k] = dot3(forward[k], d);
    }
    double fwd_a[3], fwd_b[3];
    {
      const double da[3] = {a[0] - p[0], a[1] - p[1], a[2] - p[2]};
      const double db[3] = {b[0] - p[0], b[1] - p[1], b[2] - p[2]};
      for (int k = 0; k < 3; ++k) {
        fwd_a[k] = dot3(forward[k], da);
        fwd_b[k] = dot3(forward[k], db);
      }
    }
    for (int i = 0; i < M; ++i) {
      const double *fb = &forwardB[(size_t)i * 3];
      distDs[i]        = std::fabs(forwardD[i]) / std::sqrt(dot3(fb, fb));
    }
    for (int i = 0; i < N; ++i) {
      const double *q = &forwardPC[(size_t)i * 3];
      double       *t = &tangents[(size_t)i * 4];
      distRs[i]       = std::sqrt(dot3(q, q));
      t[3]            = -distRs[i];
      for (int k = 0; k < 3; ++k) t[k] = q[k] / distRs[i];
      if (dot3(t, fwd_a) + t[3] > epsilon) {
        const double delta[3] = {q[0] - fwd_a[0], q[1] - fwd_a[1], q[2] - fwd_a[2]};
        const double s        = dot3(delta, fwd_a) / dot3(delta, delta);
        for (int k = 0; k < 3; ++k) t[k] = fwd_a[k] - s * delta[k];
        distRs[i] = std::sqrt(dot3(t, t));
        t[3]      = -distRs[i];
        for (int k = 0; k < 3; ++k) t[k] /= distRs[i];
      }
      if (dot3(t, fwd_b) + t[3] > epsilon) {
        const double delta[3] = {q[0] - fwd_b[0], q[1] - fwd_b[1], q[2] - fwd_b[2]};
        const double s        = dot3(delta, fwd_b) / dot3(delta, delta);
        for (int k = 0; k < 3; ++k) t[k] = fwd_b[k] - s * delta[k];
        distRs[i] = std::sqrt(dot3(t, t));
        t[3]      = -distRs[i];
        for (int k = 0; k < 3; ++k) t[k] /= distRs[i];
      }
      if (dot3(t, fwd_a) + t[3] > epsilon) {
        const double u[3] = {fwd_a[0] - q[0], fwd_a[1] - q[1], fwd_a[2] - q[2]};
        const double v[3] = {fwd_b[0] - q[0], fwd_b[1] - q[1], fwd_b[2] - q[2]};
        double       n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2],
                             u[0] * v[1] - u[1] * v[0]};
        const double nn   = std::sqrt(dot3(n, n));
        // Eigen normalized(): divides only when the norm is positive
        if (nn > 0)
          for (int k = 0; k < 3; ++k) n[k] /= nn;
        for (int k = 0; k < 3; ++k) t[k] = n[k];
        t[3]           = -dot3(t, fwd_a);
        const double s = t[3] > 0.0 ? -1.0 : 1.0;
        for (int k = 0; k < 4; ++k) t[k] *= s;
      }
    }
    std::fill(bdFlags.begin(), bdFlags.end(), 1);
    std::fill(pcFlags.begin(), pcFlags.end(), 1);
    nH             = 0;
    bool   completed = false;
    int    bdMinId = 0, pcMinId = 0;
    double minSqrD = INFINITY, minSqrR = INFINITY;
    for (int j = 0; j < M; ++j)
      if (distDs[j] < minSqrD) {
        minSqrD = distDs[j];
        bdMinId = j;
      }
    for (int j = 0; j < N; ++j)
      if (distRs[j] < minSqrR) {
        minSqrR = distRs[j];
        pcMinId = j;
      }
    for (int i = 0; !completed && i < (M + N); ++i) {
      double *fh = &forwardH[(size_t)nH * 4];
      if (minSqrD < minSqrR) {
        for (int k = 0; k < 3; ++k) fh[k] = forwardB[(size_t)bdMinId * 3 + k];
        fh[3]            = forwardD[bdMinId];
        bdFlags[bdMinId] = 0;
      } else {
        for (int k = 0; k < 4; ++k) fh[k] = tangents[(size_t)pcMinId * 4 + k];
        pcFlags[pcMinId] = 0;
      }
      completed = true;
      minSqrD   = INFINITY;
      for (int j = 0; j < M; ++j)
        if (bdFlags[j]) {
          completed = false;
          if (minSqrD > distDs[j]) {
            bdMinId = j;
            minSqrD = distDs[j];
          }
        }
      minSqrR = INFINITY;
      for (int j = 0; j < N; ++j)
        if (pcFlags[j]) {
          if (dot3(fh, &forwardPC[(size_t)j * 3]) + fh[3] > -epsilon) {
            pcFlags[j] = 0;
          } else {
            completed = false;
            if (minSqrR > distRs[j]) {
              pcMinId = j;
              minSqrR = distRs[j];
            }
          }
        }
      ++nH;
    }
    if (nH > 128) return -2;  // capacity of the HIP side's plane buffer (FIRI_MAX_H); treated as failure
    poly.assign((size_t)nH * 4, 0.0);
    for (int i = 0; i < nH; ++i) {
      const double *fh = &forwardH[(size_t)i * 4];
      double       *h  = &poly[(size_t)i * 4];
      for (int j = 0; j < 3; ++j)
        h[j] = (fh[0] * forward[0][j] + fh[1] * forward[1][j]) + fh[2] * forward[2][j];
      h[3] = fh[3] - dot3(h, p);
    }
    if (loop == iterations - 1) break;
    maxVolInsEllipsoid(poly.data(), nH, R, p, r);
  }
  const int kept = nH < max_faces ? nH : max_faces;
  for (int i = 0; i < kept * 4; ++i) hPoly[i] = poly[i];
  return nH;
}

// checkCorridorValidity (baseline.cpp:191-204)
bool corridorValid(const double *poly, int m) {
  std::vector<double> A((size_t)m * 3), b(m);
  for (int i = 0; i < m; ++i) {
    for (int j = 0; j < 3; ++j) A[(size_t)i * 3 + j] = poly[(size_t)i * 4 + j];
    b[i] = -poly[(size_t)i * 4 + 3];
  }
  const double c[3] = {0, 0, 0};
  double       x[3];
  return !std::isinf(orc_linprog3(c, A.data(), b.data(), m, x));
}

// checkGoalReachability (baseline.cpp:143-182)
bool goalReachable(const double *poly, int m, const double start[3], double goal[3]) {
  if (m <= 0) return true;
  double mx = -INFINITY;
  for (int i = 0; i < m; ++i) {
    const double *h = poly + (size_t)i * 4;
    mx              = std::max(mx, dot3(h, goal) + h[3] * 1.0);
  }
  if (mx <= 0) return true;
  std::vector<double> A((size_t)m * 3), b(m);
  for (int i = 0; i < m; ++i) {
    for (int j = 0; j < 3; ++j) A[(size_t)i * 3 + j] = poly[(size_t)i * 4 + j];
    b[i] = -poly[(size_t)i * 4 + 3];
  }
  double c[3] = {-goal[0] + start[0], -goal[1] + start[1], -goal[2] + start[2]};
  double gmax[3], gmin[3];
  orc_linprog3(c, A.data(), b.data(), m, gmax);
  for (int j = 0; j < 3; ++j) c[j] = goal[j] - start[j];
  orc_linprog3(c, A.data(), b.data(), m, gmin);
  for (int j = 0; j < 3; ++j) goal[j] = 0.5 * (gmax[j] + gmin[j]);
  return false;
}

}  // namespace

extern "C" {

int orc_firi(const double *bd, int n_bd, const double *pc, int n_pc, const double a[3],
             const double b[3], int iterations, double *hpoly, int max_faces, double r[3]) {
  return firi(bd, n_bd, pc, n_pc, a, b, iterations, hpoly, max_faces, r);
}

void orc_lbfgs_set_max_iterations(int k) { g_lbfgs_max_iterations = k > 0 ? k : 0; }

int orc_mvie(const double *hpoly, int m, double Rio[9], double p[3], double r[3]) {
  M3 R;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = Rio[i * 3 + j];
  const bool ok = maxVolInsEllipsoid(hpoly, m, R, p, r);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rio[i * 3 + j] = R[i][j];
  return ok ? 1 : 0;
}

// Corridor stage.  out_polys: [SOGM_MAX_PIECES][max_faces][4]; out_nfaces [SOGM_MAX_PIECES];
// returns the number of polytopes kept (0 = replan() returns false at this stage).
int orc_corridor_generate(const SogmSpec *s, const SogmPlannerParams *pp, const float *grid,
                          const float pose[3], double stamp, const double start_pva[9],
                          double t_start, const double *route, int route_len, double *out_polys,
                          int *out_nfaces, double out_goal[6]) {
  const int MF = pp->max_faces;
  for (int i = 0; i < SOGM_MAX_PIECES; ++i) out_nfaces[i] = 0;
  for (int i = 0; i < 6; ++i) out_goal[i] = 0;
  if (!pp->fake_planner && route_len < 2) return 0;  // baseline.cpp:304-307
  if (route_len < 1) return 0;
  const double *start_pos = start_pva;
  std::vector<double> wpts((size_t)route_len * 3);
  for (int i = 0; i < route_len; ++i) {
    for (int k = 0; k < 3; ++k) wpts[(size_t)i * 3 + k] = route[(size_t)i * 6 + k];
    if (wpts[(size_t)i * 3 + 2] < 0) wpts[(size_t)i * 3 + 2] = 0.1;
  }
  double lower[3] = {-4 + start_pos[0], -4 + start_pos[1], -1 + start_pos[2]};
  double higher[3] = {4 + start_pos[0], 4 + start_pos[1], 1 + start_pos[2]};
  if (lower[2] < 0) lower[2] = 0;
  if (higher[2] > 4) higher[2] = 4;
  // getInitCorridor (baseline.cpp:127-141)
  double bd[24] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 0};
  std::vector<std::vector<double>> hPolys;
  std::vector<double>              pc((size_t)pp->pc_capacity * 3);
  for (int i = 0; i < route_len - 1 && (int)hPolys.size() < SOGM_MAX_PIECES; ++i) {
    const double *w0 = &wpts[(size_t)i * 3], *w1 = &wpts[(size_t)(i + 1) * 3];
    double        lhc[3], llc[3];
    for (int k = 0; k < 3; ++k) {
      lhc[k] = std::min(std::max(w0[k], w1[k]) + pp->init_range, higher[k]);
      llc[k] = std::max(std::min(w0[k], w1[k]) - pp->init_range, lower[k]);
    }
    for (int k = 0; k < 3; ++k) {
      bd[k * 4 + 3]       = -lhc[k];
      bd[(k + 3) * 4 + 3] = llc[k];
    }
    const double t1 = t_start + i * pp->corridor_tau;
    const double t2 = t_start + (i + 1) * pp->corridor_tau;
    int n = orc_obstacle_points(s, grid, pose, stamp, t1, t2, llc, lhc, pc.data(), pp->pc_capacity);
    // capacity limits (not in the reference, which grows its vectors): a corridor whose point list
    // or face list would be truncated is unsafe, so it is treated as invalid (loop breaks)
    if (n > pp->pc_capacity) break;
    std::vector<double> hp((size_t)MF * 4, 0.0);
    double              r[3] = {1, 1, 1};
    int nf = firi(bd, 6, pc.data(), n, w0, w1, pp->firi_iterations, hp.data(), MF, r);
    if (nf == -2 || nf > MF) break;  // capacity exceeded -> invalid
    if (nf < 0) nf = 0;  // seed outside bd: hPoly stays empty (firi's return value is ignored)
    // ShrinkCorridor(hPoly, path)
    const double path[3] = {w1[0] - w0[0], w1[1] - w0[1], w1[2] - w0[2]};
    for (int f = 0; f < nf; ++f) {
      double      *h    = &hp[(size_t)f * 4];
      const double nrm  = std::sqrt(dot3(h, h));
      if (pp->fake_planner) {
        // baseline_fake.cpp:211-223
        const double pn = std::sqrt(dot3(path, path));
        if (dot3(h, path) / nrm / pn > 0.8) continue;
        if (std::fabs(h[2]) / nrm > 0.8) continue;
      }
      h[3] += nrm * pp->shrink_size;
    }
    if (!corridorValid(hp.data(), nf)) break;
    hp.resize((size_t)nf * 4);
    hPolys.push_back(hp);
  }
  if (hPolys.empty()) return 0;  // the reference would index hPolys[0] (size_t underflow)
  // adjacent intersection (baseline.cpp:366-377 / baseline_fake.cpp:375-386)
  for (size_t i = 0; i + 1 < hPolys.size(); ++i) {
    std::vector<double> both(hPolys[i]);
    both.insert(both.end(), hPolys[i + 1].begin(), hPolys[i + 1].end());
    if (!corridorValid(both.data(), (int)both.size() / 4)) {
      if (i < 2) return 0;
      hPolys.erase(hPolys.begin() + (pp->fake_planner ? i + 1 : i), hPolys.end());
      break;
    }
  }
  if (pp->fake_planner ? hPolys.empty() : hPolys.size() <= 1) return 0;
  // goal (baseline_fake.cpp:400-414)
  double gpos[3], gvel[3];
  int    gi = (int)hPolys.size() - 1;
  for (int k = 0; k < 3; ++k) {
    gpos[k] = route[(size_t)gi * 6 + k];
    gvel[k] = route[(size_t)gi * 6 + 3 + k];
  }
  auto scan = [&]() {
    for (int it = (int)hPolys.size() - 1; it != 0; --it) {
      if (goalReachable(hPolys[it].data(), (int)hPolys[it].size() / 4, start_pos, gpos)) {
        hPolys.erase(hPolys.begin() + it + 1, hPolys.end());
        const int idx = (int)hPolys.size() - 1;
        for (int k = 0; k < 3; ++k) {
          gpos[k] = route[(size_t)idx * 6 + k];
          gvel[k] = route[(size_t)idx * 6 + 3 + k];
        }
        break;
      }
    }
  };
  if (pp->fake_planner) {
    scan();
  } else {
    // baseline.cpp:391-403
    if (!goalReachable(hPolys.back().data(), (int)hPolys.back().size() / 4, start_pos, gpos)) scan();
  }
  for (size_t i = 0; i < hPolys.size(); ++i) {
    out_nfaces[i] = (int)hPolys[i].size() / 4;
    std::memcpy(out_polys + i * (size_t)MF * 4, hPolys[i].data(), hPolys[i].size() * 8);
  }
  for (int k = 0; k < 3; ++k) {
    out_goal[k]     = gpos[k];
    out_goal[3 + k] = gvel[k];
  }
  return (int)hPolys.size();
}

}  // extern "C"
