// lp_oracle.cpp — low-dimensional LP  min c^T x  s.t.  A x <= b   (d = 3 or 4).
// TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Role in the reference: sdlp::linprog<d> (traj_utils/include/traj_utils/sdlp.hpp:709-787), a
// projective-space implementation of Seidel's randomised incremental LP, called for corridor
// validity / intersection / goal projection (plan_manager/src/baseline.cpp:143-204) and for the
// deepest interior point of the MVIE (plan_manager/include/sfc_gen/firi.hpp:146-164).
//
// This is a restatement of the PUBLISHED algorithm (R. Seidel, "Small-dimensional linear
// programming and convex hulls made easy", 1991), not of sdlp's source: incremental insertion,
// on violation recurse on the violated hyperplane with one variable eliminated, 1-D base case.
// Differences a caller can observe, all outside this path's use (every polytope here is bounded by
// the corridor's bounding box, firi.hpp:313-349 always keeps the 6 box planes):
//   * a true bounding box |x_j| <= 1e4 makes every sub-problem bounded; a solution on that box is
//     reported as unbounded (-inf), as sdlp does;
//   * the insertion order is a fixed pseudo-random permutation (sdlp uses a process-global
//     mt19937_64, so its order depends on call history — not reproducible in a batched setting);
//     for a non-degenerate LP the optimum is independent of the order;
//   * with c = 0 any feasible point may be returned (callers only test isinf()).
// Parity unpinned (no reference test covers sdlp); tests/test_lp_oracle.py checks optimal values
// against scipy.optimize.linprog (HiGHS).
#include <cmath>
#include <vector>

#include "oracle.h"

namespace {

const double LP_BOX  = 1.0e4;   // true bounding box half-size
const double LP_BIG  = 1.0e7;   // implicit start box of the sub-levels
const double LP_TOL  = 1.0e-10; // violation tolerance on unit-normalised rows
const double LP_TINY = 1.0e-12; // a projected row with a smaller inf-norm is treated as 0

// rows: a[i*D .. i*D+D), b[i];  returns false if infeasible
template <int D>
struct Seidel {
  static bool solve(const double *a, const double *b, int m, const double *c, double *x,
                    std::vector<double> *scratch /* [D-1 levels] */) {
    for (int j = 0; j < D; ++j) x[j] = c[j] > 0 ? -LP_BIG : (c[j] < 0 ? LP_BIG : 0.0);
    std::vector<double> &pa = scratch[0];
    pa.resize((size_t)m * (D - 1) + m);
    double *sa = pa.data();
    double *sb = pa.data() + (size_t)m * (D - 1);
    for (int i = 0; i < m; ++i) {
      const double *ai = a + (size_t)i * D;
      double        v  = 0;
      for (int j = 0; j < D; ++j) v += ai[j] * x[j];
      if (v <= b[i] + LP_TOL) continue;
      // eliminate the variable with the largest coefficient
      int    k  = 0;
      double mx = std::fabs(ai[0]);
      for (int j = 1; j < D; ++j)
        if (std::fabs(ai[j]) > mx) {
          mx = std::fabs(ai[j]);
          k  = j;
        }
      if (mx < LP_TINY) return false;  // 0 * x <= b with b < 0
      const double inv = 1.0 / ai[k];
      for (int r = 0; r < i; ++r) {
        const double *ar = a + (size_t)r * D;
        const double  f  = ar[k] * inv;
        int           q  = 0;
        for (int j = 0; j < D; ++j)
          if (j != k) sa[(size_t)r * (D - 1) + q++] = ar[j] - f * ai[j];
        sb[r] = b[r] - f * b[i];
      }
      double cc[D > 1 ? D - 1 : 1];
      {
        const double f = c[k] * inv;
        int          q = 0;
        for (int j = 0; j < D; ++j)
          if (j != k) cc[q++] = c[j] - f * ai[j];
      }
      double xs[D > 1 ? D - 1 : 1];
      if (!Seidel<D - 1>::solve(sa, sb, i, cc, xs, scratch + 1)) return false;
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
  static bool solve(const double *a, const double *b, int m, const double *c, double *x,
                    std::vector<double> *) {
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

// fixed pseudo-random permutation (LCG Fisher-Yates), identical on the HIP side
void fixed_permutation(int n, int *p) {
  for (int i = 0; i < n; ++i) p[i] = i;
  unsigned long long s = 0x9E3779B97F4A7C15ULL;
  for (int i = n - 1; i > 0; --i) {
    s                = s * 6364136223846793005ULL + 1442695040888963407ULL;
    const int j      = (int)((s >> 33) % (unsigned long long)(i + 1));
    const int t      = p[i];
    p[i]             = p[j];
    p[j]             = t;
  }
}

template <int D>
double linprog(const double *c, const double *A, const double *b, int m, double *x) {
  // sdlp.hpp:720-724: no constraints
  for (int j = 0; j < D; ++j) x[j] = 0.0;
  if (m <= 0) {
    double mx = 0;
    for (int j = 0; j < D; ++j) mx = std::max(mx, std::fabs(c[j]));
    return mx > 0.0 ? -INFINITY : 0.0;
  }
  const int           M = m + 2 * D;
  std::vector<double> a((size_t)M * D, 0.0), bb(M);
  for (int j = 0; j < D; ++j) {  // true box first
    a[(size_t)(2 * j) * D + j]     = 1.0;
    bb[2 * j]                      = LP_BOX;
    a[(size_t)(2 * j + 1) * D + j] = -1.0;
    bb[2 * j + 1]                  = LP_BOX;
  }
  std::vector<int> perm(m);
  fixed_permutation(m, perm.data());
  for (int i = 0; i < m; ++i) {
    const double *src = A + (size_t)perm[i] * D;
    double        nn  = 0;
    for (int j = 0; j < D; ++j) nn += src[j] * src[j];
    nn = std::sqrt(nn);
    double *dst = &a[(size_t)(2 * D + i) * D];
    if (nn > 0) {
      for (int j = 0; j < D; ++j) dst[j] = src[j] / nn;
      bb[2 * D + i] = b[perm[i]] / nn;
    } else {
      for (int j = 0; j < D; ++j) dst[j] = 0;
      bb[2 * D + i] = b[perm[i]];
    }
  }
  std::vector<double> scratch[D];
  double              xs[D];
  if (!Seidel<D>::solve(a.data(), bb.data(), M, c, xs, scratch)) return INFINITY;
  for (int j = 0; j < D; ++j) x[j] = xs[j];
  for (int j = 0; j < D; ++j)
    if (std::fabs(xs[j]) > 0.99 * LP_BOX) return -INFINITY;
  double v = 0;
  for (int j = 0; j < D; ++j) v += c[j] * xs[j];
  return v;
}

}  // namespace

double orc_linprog3(const double *c, const double *A, const double *b, int m, double *x) {
  return linprog<3>(c, A, b, m, x);
}
double orc_linprog4(const double *c, const double *A, const double *b, int m, double *x) {
  return linprog<4>(c, A, b, m, x);
}

extern "C" double orc_linprog(int d, const double *c, const double *A, const double *b, int m,
                              double *x) {
  if (d == 3) return linprog<3>(c, A, b, m, x);
  if (d == 4) return linprog<4>(c, A, b, m, x);
  return NAN;
}
