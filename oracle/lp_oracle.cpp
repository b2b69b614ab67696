// lp_oracle.cpp — low-dimensional LP  min c^T x  s.t.  A x <= b   (d = 3 or 4).
// TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Restates sdlp::linprog<d> (traj_utils/include/traj_utils/sdlp.hpp:709-787): Hohmeyer's projective
// formulation of Seidel's randomised incremental LP — the problem is lifted to homogeneous
// coordinates (d+1 numbers per half-space, plane 0 = "x_d >= 0"), the objective is the ratio
// n.x / d.x, and a violated plane recurses into the (d-1)-dimensional problem on that plane with the
// coordinate of its largest coefficient eliminated.  Followed block by block, same operation order:
//   unit / lp_no_con            sdlp.hpp:76-129
//   move_to_front               sdlp.hpp:132-150   (doubly linked list shared by all recursion levels)
//   lp_min_lin_rat              sdlp.hpp:152-258
//   wedge / lp_base_case        sdlp.hpp:260-446   (the 1-D problem on the projective line)
//   findimax, vector_up, vector_down, plane_down   sdlp.hpp:449-524
//   linfracprog<d>              sdlp.hpp:526-684
//   rand_permutation            sdlp.hpp:686-706
//   linprog<d>                  sdlp.hpp:709-787
// Callers in the reference: corridor validity / intersection / goal projection
// (plan_manager/src/baseline_fake.cpp:143-199, baseline.cpp:143-204) and the deepest interior point of
// the MVIE (plan_manager/include/sfc_gen/firi.hpp:150-165).
//
// The insertion order.  sdlp shuffles the planes with a function-local static std::mt19937_64
// (default seed) through std::uniform_int_distribution<int> (sdlp.hpp:686-705): deterministic per
// process but dependent on every earlier call.  Two modes here:
//   mode 0 (default; what the batched HIP path does — the one documented deviation): a fixed LCG
//          Fisher-Yates permutation that depends on the row count only;
//   mode 1: the reference's generator, std::mt19937_64 + std::uniform_int_distribution<int> of THIS
//          image's libstdc++ (GCC 11: Lemire's multiply-shift rejection), advanced call by call;
//          orc_lp_rng_reset() puts it back to the state of a fresh process, so a single-agent call
//          sequence is replayable;
//   mode 2: the same std::mt19937_64 stream mapped to [0, n) as libstdc++ <= 10 does (scaling =
//          (2^64-1) / n, reject >= n*scaling, divide) — the reference's tested platform is Ubuntu
//          20.04 (README.md:9), i.e. GCC 9.4, whose uniform_int_distribution is that algorithm.
// orc_linprog_perm() takes the permutation as an explicit input.
//
// Not reproducible at source level (Eigen internals, parity unpinned): halves.colwise().normalize()
// (sdlp.hpp:740) sums the d+1 squares in an order that depends on Eigen's vectorisation and the
// column's address alignment; the sequential order is used here.  c.dot(x) likewise (only its sign /
// finiteness is consumed by the callers).
#include <cmath>
#include <random>
#include <vector>

#include "oracle.h"

namespace {

const double kEps = 1.0e-12;  // sdlp.hpp:35

enum { MINIMUM = 0, INFEASIBLE, UNBOUNDED, AMBIGUOUS };  // sdlp.hpp:37-47

inline double dot2(const double a[2], const double b[2]) { return a[0] * b[0] + a[1] * b[1]; }
inline double cross2(const double a[2], const double b[2]) { return a[0] * b[1] - a[1] * b[0]; }

// sdlp.hpp:61-73
inline bool unit2(const double a[2], double b[2]) {
  const double mag = std::sqrt(a[0] * a[0] + a[1] * a[1]);
  if (mag < 2.0 * kEps) return true;
  b[0] = a[0] / mag;
  b[1] = a[1] / mag;
  return false;
}

// sdlp.hpp:76-94: normalise a (d+1)-vector; true = it was (numerically) zero
bool unit(int d, double *a) {
  double mag = 0.0;
  for (int i = 0; i <= d; i++) mag += a[i] * a[i];
  if (mag < (d + 1) * kEps * kEps) return true;
  mag = 1.0 / std::sqrt(mag);
  for (int i = 0; i <= d; i++) a[i] *= mag;
  return false;
}

// sdlp.hpp:97-129: optimum of the objective without constraints
int lp_no_con(int d, const double *n_vec, const double *d_vec, double *opt) {
  double n_dot_d = 0.0, d_dot_d = 0.0;
  for (int i = 0; i <= d; i++) {
    n_dot_d += n_vec[i] * d_vec[i];
    d_dot_d += d_vec[i] * d_vec[i];
  }
  if (d_dot_d < kEps * kEps) {
    n_dot_d = 0.0;
    d_dot_d = 1.0;
  }
  for (int i = 0; i <= d; i++) opt[i] = -n_vec[i] + d_vec[i] * n_dot_d / d_dot_d;
  if (unit(d, opt)) {
    opt[d] = 1.0;
    return AMBIGUOUS;
  }
  return MINIMUM;
}

// sdlp.hpp:132-150: returns the plane index that is in i's place afterwards
int move_to_front(int i, int *next, int *prev) {
  if (i == 0 || i == next[0]) return i;
  const int previ = prev[i];
  next[prev[i]]   = next[i];
  prev[next[i]]   = prev[i];
  next[i]         = next[0];
  prev[i]         = 0;
  prev[next[i]]   = i;
  next[0]         = i;
  return previ;
}

// sdlp.hpp:152-258
void lp_min_lin_rat(bool degen, const double cw_vec[2], const double ccw_vec[2], const double n_vec[2],
                    const double d_vec[2], double opt[2]) {
  const double d_cw = dot2(cw_vec, d_vec), d_ccw = dot2(ccw_vec, d_vec);
  const double n_cw = dot2(cw_vec, n_vec), n_ccw = dot2(ccw_vec, n_vec);
  bool take_cw;
  if (degen) {
    take_cw = n_cw / d_cw < n_ccw / d_ccw;
  } else if (std::fabs(d_cw) > 2.0 * kEps && std::fabs(d_ccw) > 2.0 * kEps) {
    if (d_cw * d_ccw > 0.0) {
      take_cw = n_cw / d_cw < n_ccw / d_ccw;
    } else {  // the valid region contains a pole
      if (d_cw > 0.0) {
        opt[0] = -d_vec[1];
        opt[1] = d_vec[0];
      } else {
        opt[0] = d_vec[1];
        opt[1] = -d_vec[0];
      }
      return;
    }
  } else if (std::fabs(d_cw) > 2.0 * kEps) {
    take_cw = n_ccw * d_cw > 0.0;  // CCW bound near a pole
  } else if (std::fabs(d_ccw) > 2.0 * kEps) {
    take_cw = !(n_cw * d_ccw > 2.0 * kEps);  // CW bound near a pole
  } else {
    take_cw = cross2(d_vec, n_vec) > 0.0;  // both near poles
  }
  const double *src = take_cw ? cw_vec : ccw_vec;
  opt[0]            = src[0];
  opt[1]            = src[1];
}

// sdlp.hpp:260-375: feasible wedge [cw, ccw] of the projective line; halves has stride 2
int wedge(const double *halves, int m, int *next, int *prev, double cw_vec[2], double ccw_vec[2],
          bool *degen) {
  int i;
  *degen = false;
  for (i = 0; i != m; i = next[i]) {
    if (!unit2(halves + 2 * i, ccw_vec)) {
      cw_vec[0]  = ccw_vec[1];
      cw_vec[1]  = -ccw_vec[0];
      ccw_vec[0] = -cw_vec[0];
      ccw_vec[1] = -cw_vec[1];
      break;
    }
  }
  if (i == m) return UNBOUNDED;
  i = 0;
  while (i != m) {
    const double *h         = halves + 2 * i;
    bool          offensive = false;
    const double  d_cw = dot2(cw_vec, h), d_ccw = dot2(ccw_vec, h);
    if (d_ccw >= 2.0 * kEps) {
      if (d_cw <= -2.0 * kEps) {
        cw_vec[0] = h[1];
        cw_vec[1] = -h[0];
        unit2(cw_vec, cw_vec);
        offensive = true;
      }
    } else if (d_cw >= 2.0 * kEps) {
      if (d_ccw <= -2.0 * kEps) {
        ccw_vec[0] = -h[1];
        ccw_vec[1] = h[0];
        unit2(ccw_vec, ccw_vec);
        offensive = true;
      }
    } else if (d_ccw <= -2.0 * kEps && d_cw <= -2.0 * kEps) {
      return INFEASIBLE;
    } else if (d_cw <= -2.0 * kEps || d_ccw <= -2.0 * kEps || cross2(cw_vec, h) < 0.0) {
      if (d_cw <= -2.0 * kEps)
        unit2(ccw_vec, cw_vec);
      else if (d_ccw <= -2.0 * kEps)
        unit2(cw_vec, ccw_vec);
      *degen    = true;
      offensive = true;
    }
    if (offensive) i = move_to_front(i, next, prev);
    i = next[i];
    if (*degen) break;
  }
  if (*degen) {
    while (i != m) {
      const double *h    = halves + 2 * i;
      const double  d_cw = dot2(cw_vec, h), d_ccw = dot2(ccw_vec, h);
      if (d_cw < -2.0 * kEps) {
        if (d_ccw < -2.0 * kEps) return INFEASIBLE;
        cw_vec[0] = ccw_vec[0];
        cw_vec[1] = ccw_vec[1];
      } else if (d_ccw < -2.0 * kEps) {
        ccw_vec[0] = cw_vec[0];
        ccw_vec[1] = cw_vec[1];
      }
      i = next[i];
    }
  }
  return MINIMUM;
}

// sdlp.hpp:378-446
int lp_base_case(const double *halves, int m, const double n_vec[2], const double d_vec[2],
                 double opt[2], int *next, int *prev) {
  double cw_vec[2], ccw_vec[2];
  bool   degen;
  int    status = wedge(halves, m, next, prev, cw_vec, ccw_vec, &degen);
  if (status == INFEASIBLE) return status;
  if (status == UNBOUNDED) return lp_no_con(1, n_vec, d_vec, opt);
  if (std::fabs(cross2(n_vec, d_vec)) < 2.0 * kEps * kEps) {
    if (dot2(n_vec, n_vec) < 2.0 * kEps * kEps || dot2(d_vec, d_vec) > 2.0 * kEps * kEps) {
      opt[0] = cw_vec[0];
      opt[1] = cw_vec[1];
      status = AMBIGUOUS;
    } else {
      if (!degen && cross2(cw_vec, n_vec) <= 0.0 && cross2(n_vec, ccw_vec) <= 0.0) {
        opt[0] = -n_vec[0];
        opt[1] = -n_vec[1];
      } else if (dot2(n_vec, cw_vec) > dot2(n_vec, ccw_vec)) {
        opt[0] = ccw_vec[0];
        opt[1] = ccw_vec[1];
      } else {
        opt[0] = cw_vec[0];
        opt[1] = cw_vec[1];
      }
      status = MINIMUM;
    }
  } else {
    lp_min_lin_rat(degen, cw_vec, ccw_vec, n_vec, d_vec, opt);
    status = MINIMUM;
  }
  return status;
}

// sdlp.hpp:449-464
int findimax(int d, const double *pln) {
  int    imax = 0;
  double rmax = std::fabs(pln[0]);
  for (int i = 1; i <= d; i++) {
    const double ab = std::fabs(pln[i]);
    if (ab > rmax) {
      imax = i;
      rmax = ab;
    }
  }
  return imax;
}

// sdlp.hpp:466-483
void vector_up(int d, const double *equation, int ivar, const double *low_vector, double *vector) {
  vector[ivar] = 0.0;
  for (int i = 0; i <= d; i++) {
    if (i != ivar) {
      const int j = i < ivar ? i : i - 1;
      vector[i]   = low_vector[j];
      vector[ivar] -= equation[i] * low_vector[j];
    }
  }
  vector[ivar] /= equation[ivar];
}

// sdlp.hpp:485-507
void vector_down(int d, const double *elim_eqn, int ivar, const double *old_vec, double *new_vec) {
  double ve = 0.0, ee = 0.0;
  for (int i = 0; i <= d; i++) {
    ve += old_vec[i] * elim_eqn[i];
    ee += elim_eqn[i] * elim_eqn[i];
  }
  const double fac = ve / ee;
  for (int i = 0; i <= d; i++)
    if (i != ivar) new_vec[i < ivar ? i : i - 1] = old_vec[i] - elim_eqn[i] * fac;
}

// sdlp.hpp:509-524
void plane_down(int d, const double *elim_eqn, int ivar, const double *old_plane, double *new_plane) {
  const double crit = old_plane[ivar] / elim_eqn[ivar];
  for (int i = 0; i <= d; i++)
    if (i != ivar) new_plane[i < ivar ? i : i - 1] = old_plane[i] - elim_eqn[i] * crit;
}

// sdlp.hpp:526-684.  halves: max_size x (d+1); the list 0, next[0], ... ends at marker m.
// levels[d-2] is the sub-problem's plane array (max_size x d), shared storage like sdlp's `work`.
int linfracprog(int d, const double *halves, int max_size, int m, const double *n_vec,
                const double *d_vec, double *opt, std::vector<double> *levels, int *next, int *prev) {
  if (d == 1) {  // sdlp.hpp:664-684
    if (m > 0) return lp_base_case(halves, m, n_vec, d_vec, opt, next, prev);
    return lp_no_con(1, n_vec, d_vec, opt);
  }
  double val = 0.0;
  for (int j = 0; j <= d; j++) val += d_vec[j] * d_vec[j];
  const bool d_vec_zero = val < (d + 1) * kEps * kEps;

  int status = lp_no_con(d, n_vec, d_vec, opt);
  if (m <= 0) return status;

  double  new_opt[4], new_n_vec[4], new_d_vec[4];
  std::vector<double> &store = levels[d - 2];
  if ((int)store.size() < max_size * d) store.resize((size_t)max_size * d);
  double *new_halves = store.data();

  for (int i = 0; i != m; i = next[i]) {
    const double *plane_i = halves + (size_t)i * (d + 1);
    val                   = 0.0;
    for (int j = 0; j <= d; j++) val += opt[j] * plane_i[j];
    if (val < -(d + 1) * kEps) {
      const int imax = findimax(d, plane_i);
      if (i != 0) {
        const double fac = 1.0 / plane_i[imax];
        for (int j = 0; j != i; j = next[j]) {
          const double *old_plane = halves + (size_t)j * (d + 1);
          const double  crit      = old_plane[imax] * fac;
          double       *new_plane = new_halves + (size_t)j * d;
          for (int k = 0; k <= d; k++)
            if (k != imax) new_plane[k < imax ? k : k - 1] = old_plane[k] - plane_i[k] * crit;
        }
      }
      if (d_vec_zero) {
        vector_down(d, plane_i, imax, n_vec, new_n_vec);
        for (int j = 0; j < d; j++) new_d_vec[j] = 0.0;
      } else {
        plane_down(d, plane_i, imax, n_vec, new_n_vec);
        plane_down(d, plane_i, imax, d_vec, new_d_vec);
      }
      status = linfracprog(d - 1, new_halves, max_size, i, new_n_vec, new_d_vec, new_opt, levels,
                           next, prev);
      if (status == INFEASIBLE) return status;
      vector_up(d, plane_i, imax, new_opt, opt);
      double mag = 0.0;
      for (int j = 0; j <= d; j++) mag += opt[j] * opt[j];
      mag = 1.0 / std::sqrt(mag);
      for (int j = 0; j <= d; j++) opt[j] *= mag;
      i = move_to_front(i, next, prev);
    }
  }
  return status;
}

// mode-0 insertion order: LCG Fisher-Yates, a function of n only (identical in sogm_corridor.hip)
void fixed_permutation(int n, int *p) {
  for (int i = 0; i < n; ++i) p[i] = i;
  unsigned long long s = 0x9E3779B97F4A7C15ULL;
  for (int i = n - 1; i > 0; --i) {
    s           = s * 6364136223846793005ULL + 1442695040888963407ULL;
    const int j = (int)((s >> 33) % (unsigned long long)(i + 1));
    const int t = p[i];
    p[i]        = p[j];
    p[j]        = t;
  }
}

int g_lp_mode = 0;

// sdlp.hpp:686-706 with the generator state made resettable
std::mt19937_64 &sdlp_gen() {
  static std::mt19937_64 gen;
  return gen;
}
void sdlp_rand_permutation(int n, int *p) {
  typedef std::uniform_int_distribution<int> rand_int;
  typedef rand_int::param_type               rand_range;
  static rand_int rdi(0, 1);
  for (int i = 0; i < n; i++) p[i] = i;
  for (int i = 0; i < n; i++) {
    int draw;
    if (g_lp_mode == 2) {  // libstdc++ <= 10, bits/uniform_int_dist.h: downscaling branch
      const unsigned long long uerange = (unsigned long long)(n - i - 1) + 1ULL;
      const unsigned long long scaling = 0xFFFFFFFFFFFFFFFFULL / uerange;
      const unsigned long long past    = uerange * scaling;
      unsigned long long       ret;
      do ret = sdlp_gen()(); while (ret >= past);
      draw = (int)(ret / scaling);
    } else {
      rdi.param(rand_range(0, n - i - 1));
      draw = rdi(sdlp_gen());
    }
    const int j = draw + i;
    const int k = p[j];
    p[j]        = p[i];
    p[i]        = k;
  }
}

// sdlp.hpp:709-787 with the permutation as an input
double linprog_perm(int d, const double *c, const double *A, const double *b, int rows,
                    const int *perm, double *x) {
  const int m = rows + 1;
  for (int j = 0; j < d; ++j) x[j] = 0.0;
  if (m <= 1) {
    double mx = 0;
    for (int j = 0; j < d; ++j) mx = std::max(mx, std::fabs(c[j]));
    return mx > 0.0 ? -INFINITY : 0.0;
  }
  std::vector<int>    next(m), prev(m + 1);
  std::vector<double> halves((size_t)m * (d + 1), 0.0);
  double              n_vec[5], d_vec[5], opt[5];
  halves[d] = 1.0;  // plane 0 = (0, ..., 0, 1)
  for (int i = 1; i < m; ++i) {
    double *h = &halves[(size_t)i * (d + 1)];
    for (int j = 0; j < d; ++j) h[j] = -A[(size_t)(i - 1) * d + j];
    h[d] = b[i - 1];
  }
  for (int i = 0; i < m; ++i) {  // halves.colwise().normalize()  (:740)
    double *h  = &halves[(size_t)i * (d + 1)];
    double  nn = 0.0;
    for (int j = 0; j <= d; ++j) nn += h[j] * h[j];
    nn = std::sqrt(nn);
    if (nn > 0.0)  // Eigen::normalize leaves a zero vector alone
      for (int j = 0; j <= d; ++j) h[j] /= nn;
  }
  for (int j = 0; j < d; ++j) {
    n_vec[j] = c[j];
    d_vec[j] = 0.0;
  }
  n_vec[d] = 0.0;
  d_vec[d] = 1.0;

  prev[0]           = 0;
  next[0]           = perm[0] + 1;
  prev[perm[0] + 1] = 0;
  for (int i = 0; i < m - 2; i++) {
    next[perm[i] + 1]     = perm[i + 1] + 1;
    prev[perm[i + 1] + 1] = perm[i] + 1;
  }
  next[perm[m - 2] + 1] = m;

  std::vector<double> levels[3];
  const int status = linfracprog(d, halves.data(), m, m, n_vec, d_vec, opt, levels, next.data(),
                                 prev.data());
  double minimum = INFINITY;
  if (status != INFEASIBLE) {
    if (opt[d] != 0.0 && status != UNBOUNDED) {
      for (int j = 0; j < d; ++j) x[j] = opt[j] / opt[d];
      minimum = 0.0;
      for (int j = 0; j < d; ++j) minimum += c[j] * x[j];
    }
    if (opt[d] == 0.0 || status == UNBOUNDED) {
      for (int j = 0; j < d; ++j) x[j] = opt[j];
      minimum = -INFINITY;
    }
  }
  return minimum;
}

double linprog(int d, const double *c, const double *A, const double *b, int rows, double *x) {
  std::vector<int> perm(rows > 0 ? rows : 1);
  if (rows > 0) {
    if (g_lp_mode != 0)
      sdlp_rand_permutation(rows, perm.data());
    else
      fixed_permutation(rows, perm.data());
  }
  return linprog_perm(d, c, A, b, rows, perm.data(), x);
}

}  // namespace

double orc_linprog3(const double *c, const double *A, const double *b, int m, double *x) {
  return linprog(3, c, A, b, m, x);
}
double orc_linprog4(const double *c, const double *A, const double *b, int m, double *x) {
  return linprog(4, c, A, b, m, x);
}

extern "C" double orc_linprog(int d, const double *c, const double *A, const double *b, int m,
                              double *x) {
  if (d == 3 || d == 4) return linprog(d, c, A, b, m, x);
  return NAN;
}
extern "C" double orc_linprog_perm(int d, const double *c, const double *A, const double *b, int m,
                                   const int *perm, double *x) {
  if (d == 3 || d == 4) return linprog_perm(d, c, A, b, m, perm, x);
  return NAN;
}
extern "C" void orc_lp_set_mode(int mode) { g_lp_mode = (mode == 1 || mode == 2) ? mode : 0; }
extern "C" void orc_lp_rng_reset(void) { sdlp_gen() = std::mt19937_64(); }
extern "C" void orc_lp_rand_permutation(int n, int *p) { sdlp_rand_permutation(n, p); }
extern "C" void orc_lp_fixed_permutation(int n, int *p) { fixed_permutation(n, p); }
