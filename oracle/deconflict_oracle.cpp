// deconflict_oracle.cpp — CPU restatement of ParticleATC::isSafeAfterOpt (row f3).  TEST INFRASTRUCTURE ONLY.
//
// traj_coordinator/src/particles.cpp:223-283: the new trajectory's control points (set A) must be
// linearly separable from the not-yet-passed control points of every other agent's active trajectory
// (set B).  The reference decides that with separator::Separator::solveModel
// (utils/separator/src/separator_glpk.cpp:75-190): the feasibility LP  n.a + d >= 1,  n.b + d <= -1
// with a zero objective, solved by GLPK (external, absent here) — true iff GLP_OPT / GLP_FEAS.
// Here the same LP goes through the sdlp restatement already used for the corridor checks (lp_oracle.cpp),
// with the variables boxed at +-1e4 (8 extra rows after the point rows): sdlp works in projective space
// and reports a feasible point AT INFINITY (two crossing segments: the plane through both is a weakly
// separating direction) as "-inf", which GLPK's affine model calls infeasible.  With the box the answer is
// finite (feasible) or +inf (infeasible).
// Parity: UNPINNED (no reference test beyond utils/separator/src/test_separator.cpp, which prints).
#include <cmath>
#include <vector>

#include "oracle.h"

extern "C" int orc_separable(const double *A, int nA, const double *B, int nB) {
  // Disjoint bounding boxes are separated by an axis-aligned plane: the LP is feasible (n, d scale
  // freely in the reference's GLPK model), no need to solve it.  Most pairs of a swarm end here.
  for (int k = 0; k < 3; ++k) {
    double loA = INFINITY, hiA = -INFINITY, loB = INFINITY, hiB = -INFINITY;
    for (int i = 0; i < nA; ++i) {
      loA = std::fmin(loA, A[i * 3 + k]);
      hiA = std::fmax(hiA, A[i * 3 + k]);
    }
    for (int i = 0; i < nB; ++i) {
      loB = std::fmin(loB, B[i * 3 + k]);
      hiB = std::fmax(hiB, B[i * 3 + k]);
    }
    if (hiA < loB || hiB < loA) return 1;
  }
  // A few more candidate normals (face / body diagonals and the line between the box centres): if the
  // projections of the two sets on one of them are disjoint the sets are separable — exact, and it spares
  // the LP for neighbours that fly side by side.  The same directions, in the same order, are tried by
  // the HIP kernel.
  {
    double cA[3], cB[3];
    for (int k = 0; k < 3; ++k) {
      double loA = INFINITY, hiA = -INFINITY, loB = INFINITY, hiB = -INFINITY;
      for (int i = 0; i < nA; ++i) {
        loA = std::fmin(loA, A[i * 3 + k]);
        hiA = std::fmax(hiA, A[i * 3 + k]);
      }
      for (int i = 0; i < nB; ++i) {
        loB = std::fmin(loB, B[i * 3 + k]);
        hiB = std::fmax(hiB, B[i * 3 + k]);
      }
      cA[k] = 0.5 * (loA + hiA);
      cB[k] = 0.5 * (loB + hiB);
    }
    double dirs[11][3] = {{1, 1, 0}, {1, -1, 0}, {1, 0, 1}, {1, 0, -1}, {0, 1, 1}, {0, 1, -1},
                          {1, 1, 1}, {1, 1, -1}, {1, -1, 1}, {1, -1, -1},
                          {cB[0] - cA[0], cB[1] - cA[1], cB[2] - cA[2]}};
    for (int q = 0; q < 11; ++q) {
      const double *n = dirs[q];
      double loA = INFINITY, hiA = -INFINITY, loB = INFINITY, hiB = -INFINITY;
      for (int i = 0; i < nA; ++i) {
        const double v = (n[0] * A[i * 3] + n[1] * A[i * 3 + 1]) + n[2] * A[i * 3 + 2];
        loA = std::fmin(loA, v);
        hiA = std::fmax(hiA, v);
      }
      for (int i = 0; i < nB; ++i) {
        const double v = (n[0] * B[i * 3] + n[1] * B[i * 3 + 1]) + n[2] * B[i * 3 + 2];
        loB = std::fmin(loB, v);
        hiB = std::fmax(hiB, v);
      }
      if (hiA < loB || hiB < loA) return 1;
    }
  }
  std::vector<double> rows((size_t)(nA + nB + 8) * 4, 0.0), rhs(nA + nB + 8);
  for (int i = 0; i < nA; ++i) {  // -(n.a + d) <= -1
    for (int k = 0; k < 3; ++k) rows[i * 4 + k] = -A[i * 3 + k];
    rows[i * 4 + 3] = -1.0;
    rhs[i]          = -1.0;
  }
  for (int i = 0; i < nB; ++i) {  // n.b + d <= -1
    for (int k = 0; k < 3; ++k) rows[(nA + i) * 4 + k] = B[i * 3 + k];
    rows[(nA + i) * 4 + 3] = 1.0;
    rhs[nA + i]            = -1.0;
  }
  for (int k = 0; k < 4; ++k) {  // x_k <= 1e4, -x_k <= 1e4
    rows[(size_t)(nA + nB + 2 * k) * 4 + k]     = 1.0;
    rows[(size_t)(nA + nB + 2 * k + 1) * 4 + k] = -1.0;
    rhs[nA + nB + 2 * k] = rhs[nA + nB + 2 * k + 1] = 1.0e4;
  }
  const double c[4] = {0, 0, 0, 0};
  double       x[4];
  const double v = orc_linprog(4, c, rows.data(), rhs.data(), nA + nB + 8, x);
  return !std::isinf(v);
}

// cpts: the new trajectory's 5*M control points; t_now: "ros::Time::now()" of the check.
extern "C" int orc_safe_after_opt(const double *cpts, int M, const SogmTrajRecord *rec, int n_rec,
                                  int ego_id, double t_now, int max_rows) {
  for (int i = 0; i < n_rec; ++i) {
    const SogmTrajRecord &r = rec[i];
    if (r.n_pieces <= 0 || r.drone_id == ego_id) continue;
    double time_end = r.time_start;
    for (int k = 0; k < r.n_pieces; ++k) time_end += r.duration[k];
    if (!(r.time_start < t_now && t_now < time_end)) continue;
    double t     = t_now - r.time_start;  // Bezier::locatePiece (bernstein.hpp:164-172)
    int    piece = r.n_pieces - 1;
    for (int k = 0; k < r.n_pieces; ++k) {
      t -= r.duration[k];
      if (t < 0) {
        piece = k;
        break;
      }
    }
    const int nB = (r.n_pieces - piece) * 5;  // cpts.bottomRows(rows - piece_idx * order)
    if (5 * M + nB > max_rows) return 0;      // capacity of the batched LP (deviation)
    if (!orc_separable(cpts, 5 * M, r.cpts + piece * 15, nB)) return 0;
  }
  return 1;
}
