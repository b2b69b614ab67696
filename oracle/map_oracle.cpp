// map_oracle.cpp — CPU restatement of the SOGM map path.  TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Follows, line by line:
//   plan_env/include/plan_env/map.h:153-215              index math
//   plan_env/src/fake_particle_risk_voxel.cpp:20-346     fkpcp SOGM update + query
//   plan_env/src/risk_base.cpp:15-337                    RiskBase query / obstacle points / overlay
//   plan_env/src/map.cpp:480-518                         MapBase::getObstaclePoints (fake map)
//   traj_coordinator/src/particles.cpp:62-87,302-422     body particles, waypoints
//   traj_utils/src/bernstein.cpp, bernstein.hpp          Bezier evaluation
// Parity: pinned only by the Bernstein KATs (tests/golden/bernstein_kat.json); everything else is
// "parity unpinned" — the reference cannot be built in this image (Eigen/ROS/PCL absent).
//
// Deliberate deviations (documented in DESIGN.md):
//   * [V][T] semantics are the intended ones; the reference's PREDICTION_TIMES 6-vs-9 aliasing
//     (fake_particle_risk_voxel.h:28-29 vs map.h:52) is undefined behaviour and is not emulated.
//   * the inclusive upper slice bound of getObstaclePoints (j <= idx_end with idx_end clamped to T,
//     risk_base.cpp:305-306,328) reads one slice past the array; slices >= T are skipped here.
#include <cmath>
#include <cstring>
#include <vector>

#include "oracle.h"
#include "../include/sogm_detmath.h"

namespace {

struct Grid {
  int   L, W, H, T;
  float res, rx, ry, rz;
  explicit Grid(const SogmSpec *s) {
    L   = s->L;
    W   = s->W;
    H   = s->H;
    T   = s->T;
    res = s->resolution;
    // risk_base.cpp:25-28 / fake_particle_risk_voxel.cpp:26-29: int/int division, then * float
    rx = (float)(L / 2) * res;
    ry = (float)(W / 2) * res;
    rz = (float)(H / 2) * res;
  }
  // map.h:153-157 (strict inequalities)
  bool inRangeF(float x, float y, float z) const {
    return x > -rx && x < rx && y > -ry && y < ry && z > -rz && z < rz;
  }
  // map.h:159-162
  bool inRangeI(int x, int y, int z) const {
    return x >= 0 && x < L && y >= 0 && y < W && z >= 0 && z < H;
  }
  // map.h:169-174: float division, truncation toward zero
  int indexF(float x, float y, float z) const {
    int ix = (int)((x + rx) / res);
    int iy = (int)((y + ry) / res);
    int iz = (int)((z + rz) / res);
    return iz * L * W + iy * L + ix;
  }
  // A write through indexF is dropped when the index is >= V: reference UB (a coordinate one ulp below +range rounds
  // up in "x + r" and in the fp32 division, the index component equals the axis size; for z the reference writes
  // outside risk_maps_).  x / y overflows that stay inside the array wrap into the next row / layer as the
  // reference's do.  The HIP path drops the same marks (csrc/sogm_map.hip, k_stamp_bits).
  bool writableF(float x, float y, float z) const { return inRangeF(x, y, z) && indexF(x, y, z) < L * W * H; }
  int indexI(int x, int y, int z) const { return z * L * W + y * L + x; }
  // map.h:186-194: voxel CORNER + pose
  void position(int index, const float pose[3], float out[3]) const {
    int x  = index % L;
    int y  = (index / L) % W;
    int z  = index / (L * W);
    out[0] = ((float)x * res - rx) + pose[0];
    out[1] = ((float)y * res - ry) + pose[1];
    out[2] = ((float)z * res - rz) + pose[2];
  }
};

const double kBern[5][5] = {{1, -4, 6, -4, 1},
                            {0, 4, -12, 12, -4},
                            {0, 0, 6, -12, 6},
                            {0, 0, 0, 4, -4},
                            {0, 0, 0, 0, 1}};

// BernsteinPiece::getPos/getVel/getAcc (bernstein.cpp:25-59): C^T * A * S evaluated left to right
void pieceEval(const double *c /*5x3*/, double t0, double tf, double t, int der, double out[3]) {
  const double dur = tf - t0;  // t_ = tf_ - t0_ (bernstein.hpp:43)
  const double s   = (t - t0) / dur;
  double       S[5];
  if (der == 0) {
    S[0] = 1;
    for (int i = 1; i <= 4; ++i) S[i] = std::pow(s, i);
  } else if (der == 1) {
    S[0] = 0;
    S[1] = 1;
    for (int i = 2; i <= 4; ++i) S[i] = i * std::pow(s, i - 1);
  } else {
    S[0] = 0;
    S[1] = 0;
    S[2] = 2;
    for (int i = 3; i <= 4; ++i) S[i] = i * (i - 1) * std::pow(s, i - 2);
  }
  for (int d = 0; d < 3; ++d) {
    double B[5];
    for (int j = 0; j < 5; ++j) {
      double acc = 0.0;
      for (int i = 0; i < 5; ++i) acc += c[i * 3 + d] * kBern[i][j];
      B[j] = acc;
    }
    double acc = 0.0;
    for (int j = 0; j < 5; ++j) acc += B[j] * S[j];
    if (der == 1) acc = acc / dur;
    if (der == 2) acc = acc / std::pow(dur, 2);
    out[d] = acc;
  }
}

// Bezier::locatePiece (bernstein.hpp:164-172)
int locatePiece(const double *dur, int M, double t) {
  for (int i = 0; i < M; ++i) {
    t -= dur[i];
    if (t < 0) return i;
  }
  return M - 1;
}

void bezierEval(const double *dur, const double *cpts, int M, double t, int der, double out[3]) {
  int    i  = locatePiece(dur, M, t);
  double t0 = 0;  // Bezier::calcPieces (bernstein.cpp:176-188) accumulates piece start times
  for (int k = 0; k < i; ++k) t0 += dur[k];
  pieceEval(cpts + i * 15, t0, t0 + dur[i], t, der, out);
}

}  // namespace

extern "C" {

void orc_bernstein_coeff(double A[25]) {
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) A[i * 5 + j] = kBern[i][j];
}

void orc_piece_eval(const double *cpts5x3, double t0, double tf, double t, int derivative,
                    double out[3]) {
  pieceEval(cpts5x3, t0, tf, t, derivative, out);
}

void orc_bezier_eval(const double *durations, const double *cpts, int M, double t, int derivative,
                     double out[3]) {
  bezierEval(durations, cpts, M, t, derivative, out);
}

// BernsteinPiece::calcDerivativeCtrlPts (bernstein.cpp:128-137)
void orc_derivative_ctrl_pts(const double *in, int n_in, double *out) {
  int n = n_in - 1;
  for (int i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) out[i * 3 + d] = n * (in[(i + 1) * 3 + d] - in[i * 3 + d]);
}

// getMaxVelRate / getMaxAccRate (bernstein.cpp:155-218)
double orc_bezier_max_rate(const double *durations, const double *cpts, int M, int derivative) {
  double best = 0;
  double t0   = 0;
  for (int p = 0; p < M; ++p) {
    double tf  = t0 + durations[p];
    double dur = tf - t0;
    t0         = tf;
    double v[12], a[9];
    orc_derivative_ctrl_pts(cpts + p * 15, 5, v);
    const double *pts = v;
    int           n   = 4;
    if (derivative == 2) {
      orc_derivative_ctrl_pts(v, 4, a);
      pts = a;
      n   = 3;
    }
    double mx = 0;
    for (int i = 0; i < n; ++i) {
      double r = std::sqrt(pts[i * 3] * pts[i * 3] + pts[i * 3 + 1] * pts[i * 3 + 1] +
                           pts[i * 3 + 2] * pts[i * 3 + 2]);
      if (r > mx) mx = r;
    }
    double rate = derivative == 1 ? mx / dur : mx / std::pow(dur, 2);
    if (rate > best) best = rate;
  }
  return best;
}

int orc_is_in_range_f(const SogmSpec *s, const float p[3]) {
  return Grid(s).inRangeF(p[0], p[1], p[2]) ? 1 : 0;
}
int orc_voxel_index_f(const SogmSpec *s, const float p[3]) {
  return Grid(s).indexF(p[0], p[1], p[2]);
}
void orc_voxel_position(const SogmSpec *s, const float pose[3], int index, float out[3]) {
  Grid(s).position(index, pose, out);
}
// fake_particle_risk_voxel.cpp:37 / risk_base.cpp:31: float / float truncated to int
int orc_inf_step(const SogmSpec *s) { return (int)(s->clearance / s->resolution); }
void orc_ranges(const SogmSpec *s, float out[3]) {
  Grid g(s);
  out[0] = g.rx;
  out[1] = g.ry;
  out[2] = g.rz;
}

// ParticleATC::initEgoParticles (particles.cpp:62-75): fp64 loop counters, STEP 0.15
int orc_ego_particles(double sx, double sy, double sz, double *out, int cap) {
  const double STEP = 0.15;
  int          n    = 0;
  for (double x = -sx / 2; x <= sx / 2; x += STEP)
    for (double y = -sy / 2; y <= sy / 2; y += STEP)
      for (double z = -sz / 2; z <= sz / 2; z += STEP) {
        if (n < cap) {
          out[n * 3 + 0] = x;
          out[n * 3 + 1] = y;
          out[n * 3 + 2] = z;
        }
        ++n;
      }
  return n;
}

// FakeParticleRiskVoxel::updateMap (fake_particle_risk_voxel.cpp:80-170), overlay excluded.
void orc_update_gt(const SogmSpec *s, const float *cloud, int n_points, const SogmCylinder *cyl,
                   int n_cyl, const float pose[3], float *grid) {
  Grid      g(s);
  const int V = g.L * g.W * g.H, T = g.T;
  // :107-108 zero the temporary grid
  std::memset(grid, 0, sizeof(float) * (size_t)V * T);

  // :88-104 PCL PassThrough on x, y, z: keep lo <= v <= hi (limits are fp32)
  const float lox = pose[0] - g.rx, hix = pose[0] + g.rx;
  const float loy = pose[1] - g.ry, hiy = pose[1] + g.ry;
  const float loz = pose[2] - g.rz, hiz = pose[2] + g.rz;
  for (int i = 0; i < n_points; ++i) {
    const float px = cloud[i * 3], py = cloud[i * 3 + 1], pz = cloud[i * 3 + 2];
    if (!(px >= lox && px <= hix)) continue;
    if (!(py >= loy && py <= hiy)) continue;
    if (!(pz >= loz && pz <= hiz)) continue;
    // :111-116
    const float x = px - pose[0], y = py - pose[1], z = pz - pose[2];
    if (g.writableF(x, y, z)) grid[(size_t)g.indexF(x, y, z) * T + 0] = 1.0F;
  }
  // :121-125 collect occupied voxels of slice 0 in index order
  std::vector<int> obs;
  for (int i = 0; i < V; ++i)
    if (grid[(size_t)i * T] > s->risk_threshold) obs.push_back(i);

  // :127-161
  for (int i : obs) {
    float pt[3];
    g.position(i, pose, pt);
    float vel[3] = {0, 0, 0};
    for (int c = 0; c < n_cyl; ++c) {
      const SogmCylinder &cy = cyl[c];
      if (cy.type == 3) {
        // :130-136  Vector3f(cyl.x, cyl.y, pt.z()); (pt - pt_cyl).norm() in fp32
        const float dx = pt[0] - (float)cy.x, dy = pt[1] - (float)cy.y, dz = pt[2] - pt[2];
        const float dist = std::sqrt(dx * dx + dy * dy + dz * dz);
        // dist (float) <= cyl.w (double) + clearance_ (float): evaluated in double
        if ((double)dist <= cy.w + (double)s->clearance) {
          vel[0] = (float)cy.vx;
          vel[1] = (float)cy.vy;
          vel[2] = 0.0F;
          break;
        }
      } else if (cy.type == 2) {
        // :137-149 ring obstacle: plane through the ring centre and the points centre + q*(0,1,0),
        // centre + q*(1,0,0).  Eigen's operation sequence in fp32:
        //   q * v  = QuaternionBase::_transformVector: uv = q.vec x v; uv += uv; v + w*uv + q.vec x uv
        //   Hyperplane::Through(p0, p1, p2): v0 = p2 - p0, v1 = p1 - p0, normal = v0 x v1 / |v0 x v1|,
        //                                    offset = -p0.normal   (the SVD fallback for collinear
        //                                    points cannot trigger: the two rotated axes are orthonormal)
        //   projection(p) = p - signedDistance(p) * normal,  signedDistance = normal.p + offset
        const float qw = (float)cy.qw, qv[3] = {(float)cy.qx, (float)cy.qy, (float)cy.qz};
        auto cross = [](const float a[3], const float b[3], float o[3]) {
          o[0] = a[1] * b[2] - a[2] * b[1];
          o[1] = a[2] * b[0] - a[0] * b[2];
          o[2] = a[0] * b[1] - a[1] * b[0];
        };
        auto rotate = [&](const float v[3], float o[3]) {
          float uv[3], w2[3];
          cross(qv, v, uv);
          for (int k = 0; k < 3; ++k) uv[k] += uv[k];
          cross(qv, uv, w2);
          for (int k = 0; k < 3; ++k) o[k] = (v[k] + qw * uv[k]) + w2[k];
        };
        const float c0[3] = {(float)cy.x, (float)cy.y, (float)cy.z};
        const float ey[3] = {0, 1, 0}, ex[3] = {1, 0, 0};
        float       ry[3], rx[3], p1[3], p2[3], v0[3], v1[3], n[3];
        rotate(ey, ry);
        rotate(ex, rx);
        for (int k = 0; k < 3; ++k) {
          p1[k] = c0[k] + ry[k];
          p2[k] = c0[k] + rx[k];
          v0[k] = p2[k] - c0[k];
          v1[k] = p1[k] - c0[k];
        }
        cross(v0, v1, n);
        const float nn = std::sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
        for (int k = 0; k < 3; ++k) n[k] /= nn;
        const float off = -((c0[0] * n[0] + c0[1] * n[1]) + c0[2] * n[2]);
        const float sd  = ((n[0] * pt[0] + n[1] * pt[1]) + n[2] * pt[2]) + off;  // signedDistance
        const float b[3]          = {pt[0] - sd * n[0], pt[1] - sd * n[1], pt[2] - sd * n[2]};
        const float dist_to_plane = std::fabs(sd);
        const float e0 = c0[0] - b[0], e1 = c0[1] - b[1], e2 = c0[2] - b[2];
        const float dist = std::sqrt((e0 * e0 + e1 * e1) + e2 * e2);
        // abs(cyl.w / 2 - dist) < 2 * resolution_  (double arithmetic on the left)
        if (std::fabs(cy.w / 2 - (double)dist) < (double)(2 * g.res) &&
            dist_to_plane < 2 * g.res) {
          vel[0] = (float)cy.vx;
          vel[1] = (float)cy.vy;
          vel[2] = 0.0F;
          break;
        }
      }
    }
    // :155-160  pt + vel * time_resolution_ * k - pose_   (fp32, left to right)
    for (int k = 1; k < T; ++k) {
      const float fx = (pt[0] + (vel[0] * s->time_resolution) * (float)k) - pose[0];
      const float fy = (pt[1] + (vel[1] * s->time_resolution) * (float)k) - pose[1];
      const float fz = (pt[2] + (vel[2] * s->time_resolution) * (float)k) - pose[2];
      if (g.writableF(fx, fy, fz)) grid[(size_t)g.indexF(fx, fy, fz) * T + k] = 1.0F;
    }
  }
}

// addOtherAgents (risk_base.cpp:136-168) == fake_particle_risk_voxel.cpp:178-218
// RiskVoxel::addOtherAgents (risk_voxel.cpp:258-288): getWaypoints directly (particles.cpp:316-344) —
// body particles while the trajectory runs, its single last point on the slice where it has ended
// (after which the chain is broken) — and addObstaclesToRiskMap (:311-318) SETS the cell to 1.0.
static void projectNeighboursRiskVoxel(const SogmSpec *s, const SogmTrajRecord *rec, int n_rec,
                                       int ego_id, const double *body, int n_body,
                                       const float pose[3], double stamp, float *grid) {
  Grid              g(s);
  const int         T = g.T;
  std::vector<char> valid(n_rec, 1);
  for (int t_idx = 0; t_idx < T; ++t_idx) {
    const double        t = stamp + (double)(s->time_resolution * (float)t_idx);  // :276
    std::vector<double> pts;
    for (int i = 0; i < n_rec; ++i) {
      const SogmTrajRecord &r = rec[i];
      if (r.drone_id == ego_id || !valid[i]) continue;
      if (r.n_pieces <= 0) {  // find_if fails -> false (particles.cpp:322)
        valid[i] = 0;
        continue;
      }
      double time_end = r.time_start;
      for (int k = 0; k < r.n_pieces; ++k) time_end += r.duration[k];
      bool ok = false;
      if (r.time_start < t && time_end > t) {
        double p[3];
        bezierEval(r.duration, r.cpts, r.n_pieces, t - r.time_start, 0, p);
        for (int e = 0; e < n_body; ++e)
          for (int k = 0; k < 3; ++k) pts.push_back(p[k] + body[e * 3 + k]);
        ok = true;
      } else if (r.time_start > t) {
        ok = true;
      } else if (time_end < t) {
        double dur = 0.0, p[3];  // Bezier::getDuration
        for (int k = 0; k < r.n_pieces; ++k) dur += r.duration[k];
        bezierEval(r.duration, r.cpts, r.n_pieces, dur, 0, p);
        for (int k = 0; k < 3; ++k) pts.push_back(p[k]);
        ok = false;
      }
      valid[i] = ok ? 1 : 0;
    }
    for (size_t e = 0; e < pts.size() / 3; ++e) {
      const float fx = (float)(pts[e * 3 + 0] - (double)pose[0]);
      const float fy = (float)(pts[e * 3 + 1] - (double)pose[1]);
      const float fz = (float)(pts[e * 3 + 2] - (double)pose[2]);
      if (!g.writableF(fx, fy, fz)) continue;
      grid[(size_t)g.indexF(fx, fy, fz) * T + t_idx] = 1.0F;
    }
  }
}

// ParticleATC's resample branch (particles.cpp:365-409), off by default; see include/sogm_abi.h sogm_set_resample for
// the injected table that stands in for std::default_random_engine(time(NULL))
static float        g_rs_rate = 0.0f;
static int          g_rs_n    = 0;
static const float *g_rs_z    = nullptr;
void orc_set_resample(float rate, int n, const float *table) {
  g_rs_rate = rate;
  g_rs_n    = n;
  g_rs_z    = table;
}

void orc_project_neighbours(const SogmSpec *s, const SogmTrajRecord *rec, int n_rec, int ego_id,
                            const double *body, int n_body, const float pose[3], double stamp,
                            float *grid) {
  if (s->map_kind == SOGM_MAP_RISKVOXEL) {
    projectNeighboursRiskVoxel(s, rec, n_rec, ego_id, body, n_body, pose, stamp, grid);
    return;
  }
  Grid              g(s);
  const int         T = g.T;
  std::vector<char> valid(n_rec, 1);  // is_swarm_traj_valid
  for (int t_idx = 0; t_idx < T; ++t_idx) {
    // risk_base.cpp:150: double + (float * int -> float)
    const double t = stamp + (double)(s->time_resolution * (float)t_idx);
    for (int i = 0; i < n_rec; ++i) {
      const SogmTrajRecord &r = rec[i];
      if (r.n_pieces <= 0) continue;  // no trajectory stored for this drone (particles.cpp:322)
      if (r.drone_id == ego_id || !valid[i]) continue;
      // getWaypoints (particles.cpp:316-344)
      double time_end = r.time_start;  // trajectoryCallback: t_end += duration[i] (:160-164)
      for (int k = 0; k < r.n_pieces; ++k) time_end += r.duration[k];
      bool                got = false;
      std::vector<double> pts;
      if (r.time_start < t && time_end > t) {
        double p[3];
        bezierEval(r.duration, r.cpts, r.n_pieces, t - r.time_start, 0, p);
        for (int e = 0; e < n_body; ++e) {  // loadParticles (:302-307)
          pts.push_back(p[0] + body[e * 3 + 0]);
          pts.push_back(p[1] + body[e * 3 + 1]);
          pts.push_back(p[2] + body[e * 3 + 2]);
        }
        got = true;
      } else if (r.time_start > t) {
        got = true;  // not started: no points, "true"
      } else {
        got = false;  // ended (the pushed last point is discarded by the caller, :354-356)
      }
      // getParticlesWithRisk (:353-356): false when !got or no waypoints
      const bool ok = got && !pts.empty();
      valid[i]      = ok ? 1 : 0;
      if (!ok) continue;
      // :365  float pos_stddev = replan_risk_rate_ * static_cast<float>(t0 - ptr->time_start)
      const float sd = g_rs_rate * (float)(t - r.time_start);
      if (g_rs_n > 0 && g_rs_z && !(sd < 1e-3F)) {  // :382-409 resample particles
        const int n = g_rs_n;
        for (size_t e = 0; e < pts.size() / 3; ++e) {
          const float       *z = g_rs_z + 3 * e * n;
          std::vector<float> risk_buf, noise;
          for (int i = 0; i < n; ++i) {
            const float nx = z[3 * i] * sd, ny = z[3 * i + 1] * sd, nz = z[3 * i + 2] * sd;  // N(0, sd) = z * sd + 0
            noise.push_back(nx);
            noise.push_back(ny);
            noise.push_back(nz);
            // :399  std::exp(-0.5f * (nx * nx + ny * ny + nz * nz) / (pos_stddev * pos_stddev))
            risk_buf.push_back(sogm_det::expf_neg((-0.5F * ((nx * nx + ny * ny) + nz * nz)) / (sd * sd)));
          }
          float sum = 0.0F;  // :405 std::accumulate(..., 0.0f)
          for (float v : risk_buf) sum += v;
          for (int i = 0; i < n; ++i) {
            const float rk = (risk_buf[i] * (float)n) / sum;  // :406 r * num_resample_ / sum
            const float fx = (float)((pts[e * 3 + 0] + (double)noise[3 * i + 0]) - (double)pose[0]);
            const float fy = (float)((pts[e * 3 + 1] + (double)noise[3 * i + 1]) - (double)pose[1]);
            const float fz = (float)((pts[e * 3 + 2] + (double)noise[3 * i + 2]) - (double)pose[2]);
            if (!g.writableF(fx, fy, fz)) continue;
            // (the reference indexes one risks vector with all agents' particles, UB: each particle adds its own weight)
            grid[(size_t)g.indexF(fx, fy, fz) * T + t_idx] += rk;
          }
        }
        continue;
      }
      // replan_risk_rate == 0  ->  risk 1.0 per particle (:375-381)
      for (size_t e = 0; e < pts.size() / 3; ++e) {
        // risk_base.cpp:162-164: pt - pose_.cast<double>(), then cast<float> (:203)
        const float fx = (float)(pts[e * 3 + 0] - (double)pose[0]);
        const float fy = (float)(pts[e * 3 + 1] - (double)pose[1]);
        const float fz = (float)(pts[e * 3 + 2] - (double)pose[2]);
        if (!g.writableF(fx, fy, fz)) continue;
        grid[(size_t)g.indexF(fx, fy, fz) * T + t_idx] += 1.0F;
      }
    }
  }
}

int orc_query_clear_idx(const SogmSpec *s, const float *grid, const float pose[3],
                        const double pos[3], int t) {
  Grid      g(s);
  const int T    = g.T;
  const int step = orc_inf_step(s);
  if (s->map_kind == SOGM_MAP_FAKE) {
    // fake_particle_risk_voxel.cpp:309-331
    if (pos[2] < (double)s->ground_height || pos[2] > (double)s->ceiling_height) return -1;
  } else {
    // risk_base.cpp:228-231
    if (pos[2] < (double)s->ground_height) return 1;
    if (pos[2] > (double)s->ceiling_height) return 1;
  }
  const float fx = (float)pos[0] - pose[0], fy = (float)pos[1] - pose[1],
              fz = (float)pos[2] - pose[2];
  // getVoxelRelIndex (map.h:201-206)
  const int ix = (int)((fx + g.rx) / g.res), iy = (int)((fy + g.ry) / g.res),
            iz = (int)((fz + g.rz) / g.res);
  if (!g.inRangeI(ix, iy, iz)) return -1;
  float sum = 0.0F;
  // kernel order x, y, z nested (risk_base.cpp:33-39); the fake map's z loop degenerates to z = 0
  // (fake_particle_risk_voxel.cpp:41: `for (int z = -0.3; z <= 0.3; z++)`)
  const int zs = s->map_kind == SOGM_MAP_FAKE ? 0 : step;
  for (int x = -step; x <= step; ++x)
    for (int y = -step; y <= step; ++y)
      for (int z = -zs; z <= zs; ++z) {
        const int px = ix + x, py = iy + y, pz = iz + z;
        if (!g.inRangeI(px, py, pz)) continue;
        sum += grid[(size_t)g.indexI(px, py, pz) * T + t];
        if (s->map_kind == SOGM_MAP_FAKE) {
          if (sum > s->risk_threshold) return 1;
        } else {
          // risk_base.cpp:249: float - int*float
          // RiskVoxel (risk_voxel.cpp:426): fixed threshold, no decay
          const float dec = s->map_kind == SOGM_MAP_RISKVOXEL ? 0.0F : s->risk_thres_reg_decay;
          if (sum > s->risk_threshold_region - (float)t * dec) return 1;
        }
      }
  return 0;
}

int orc_query_clear_time(const SogmSpec *s, const float *grid, const float pose[3],
                         const double pos[3], double dt) {
  // fake_particle_risk_voxel.cpp:341-346 / risk_base.cpp:256-260: double / float -> double
  int tf = (int)std::floor(dt / (double)s->time_resolution);
  tf     = tf > (s->T - 1) ? s->T - 1 : tf;
  // NOTE: a negative dt gives a negative slice in the reference (out-of-bounds read); clamp to 0.
  if (tf < 0) tf = 0;
  return orc_query_clear_idx(s, grid, pose, pos, tf);
}

int orc_obstacle_points(const SogmSpec *s, const float *grid, const float pose[3], double stamp,
                        double t_start, double t_end, const double lc[3], const double hc[3],
                        double *out, int cap) {
  Grid      g(s);
  const int T = g.T;
  // map.cpp:485-492 / risk_base.cpp:300-308
  const double tr        = (double)s->time_resolution;
  int          idx_start = (int)std::floor((t_start - stamp) / tr);
  int          idx_end   = (int)std::ceil((t_end - stamp) / tr);
  idx_start              = idx_start < 0 ? 0 : idx_start;
  idx_start              = idx_start > T ? T : idx_start;
  idx_end                = idx_end > T ? T : idx_end;
  idx_end                = idx_end < 0 ? 0 : idx_end;
  // map.cpp:495-500: (double - float + float) / float, truncation
  int lx = (int)((lc[0] - pose[0] + g.rx) / g.res);
  int ly = (int)((lc[1] - pose[1] + g.ry) / g.res);
  int lz = (int)((lc[2] - pose[2] + g.rz) / g.res);
  int hx = (int)((hc[0] - pose[0] + g.rx) / g.res);
  int hy = (int)((hc[1] - pose[1] + g.ry) / g.res);
  int hz = (int)((hc[2] - pose[2] + g.rz) / g.res);
  hx     = hx < g.L - 1 ? hx : g.L - 1;
  hy     = hy < g.W - 1 ? hy : g.W - 1;
  hz     = hz < g.H - 1 ? hz : g.H - 1;
  lx     = lx > 0 ? lx : 0;
  ly     = ly > 0 ? ly : 0;
  lz     = lz > 0 ? lz : 0;
  int n  = 0;
  for (int z = lz; z <= hz; ++z)
    for (int y = ly; y <= hy; ++y)
      for (int x = lx; x <= hx; ++x) {
        const int i = x + y * g.L + z * g.L * g.W;
        for (int j = idx_start; j <= idx_end; ++j) {
          if (j >= T) continue;  // deviation: the reference reads slice T (one past the end)
          // RiskVoxel inherits MapBase::getObstaclePoints (map.cpp:480-518): fixed threshold
          const float thr = (s->map_kind == SOGM_MAP_FAKE || s->map_kind == SOGM_MAP_RISKVOXEL)
                                ? s->risk_threshold
                                : s->risk_threshold - s->risk_thres_vox_decay * (float)j;
          if (grid[(size_t)i * T + j] > thr) {
            if (n < cap) {
              float p[3];
              g.position(i, pose, p);
              out[n * 3 + 0] = (double)p[0];
              out[n * 3 + 1] = (double)p[1];
              out[n * 3 + 2] = (double)p[2];
            }
            ++n;
          }
        }
      }
  return n;
}

}  // extern "C"

// BaselinePlanner::isTrajSafe (plan_manager/src/baseline.cpp:45-68); traj_start_time_ = r->time_start
extern "C" int orc_traj_safe(const SogmSpec *s, const float *grid, const float pose[3], double map_stamp,
                             const SogmTrajRecord *r, double t_now, double T) {
  if (r->n_pieces <= 0) return 1;
  double t0 = t_now - r->time_start;
  if (t0 < 0) t0 = 0;
  if (t0 > T) return 1;
  double dur = 0;
  for (int k = 0; k < r->n_pieces; ++k) dur += r->duration[k];
  T = T > dur ? dur : T;
  for (double t = t0; t < T; t += 0.1) {
    double p[3];
    bezierEval(r->duration, r->cpts, r->n_pieces, t, 0, p);
    const double dt = t + r->time_start - map_stamp;
    if (orc_query_clear_time(s, grid, pose, p, dt) == 1) return 0;
  }
  return 1;
}
