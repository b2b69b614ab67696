// dsp_oracle.cpp — CPU restatement of the particle-filter SOGM (row a6).  TEST INFRASTRUCTURE ONLY.
//
// Follows dsp_map::DSPMap in plan_env/include/plan_env/dsp_dynamic.h line by line, as configured by
// RiskVoxel::init (plan_env/src/risk_voxel.cpp:42-50): sequential voxel sweep, first-empty-slot
// placement, 20-deep pyramid lists, the 20000-entry Gaussian PDF table, table-driven randoms.
//
// Parity status: UNPINNED.  The reference ships no test or fixture for this class, seeds its random
// tables from time(0) (dsp_dynamic.h:629,1231) and cannot be compiled here (Eigen/PCL/munkres
// absent).  What this file pins is the arithmetic of the written algorithm; three inputs that the
// reference derives from absent third-party code are taken as data instead:
//   * the Gaussian / rand() tables are caller-supplied (same tables go to the HIP path);
//   * the per-point velocity labels {vx, vy, vz, intensity} and the point order of
//     `input_cloud_with_velocity` — produced in the reference by PCL Euclidean clustering + Munkres
//     matching (velocityEstimationThread, dsp_dynamic.h:1487-1678) — are inputs;
//   * rotateVectorByQuaternion (:1391-1411) uses Eigen's quaternion product; the scalar formula of
//     Eigen's generic (non-SIMD) product is restated below.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

struct Dsp {
  SogmSpec      spec;
  SogmDspParams P;
  int           V, S, SP, nph, npv, NP, T, OD;  // slots/voxel, slots/pyramid, pyramids h, v, total
  float         res, hx, hy, hz;
  std::vector<float> store;   // [V][S][9]   dsp_dynamic.h:88
  std::vector<float> objnum;  // [V][4+T]    :92
  std::vector<int>   pyr;     // [NP][SP][3] :96
  std::vector<int>   nbr;     // [NP][10]    :99
  std::vector<float> pc;      // [NP][OM][5] :540
  std::vector<int>   nobs;    // [NP]
  std::vector<float> maxlen;  // [NP]
  std::vector<float> bp_ori_h, bp_ori_v, bp_h, bp_v;  // [(nph+1)*3], [(npv+1)*3]
  std::vector<float> pdf;                             // [20000]
  std::vector<float> pg, vg;                          // gaussian tables
  std::vector<int>   rnd;                             // rand() table
  int                pseq = 0, vseq = 0, rseq = 0;
  float              cur[3] = {0, 0, 0}, quat[4] = {1, 0, 0, 0};
  float              dt_last = 0.f, update_time = 0.f;
  int                update_counter = 0;
  float              expected_new_born = 0.f, new_born_each_object_weight = 0.f;
  bool               first = true;
  float              last_p[3];
  double             last_t;
  std::vector<float> born;  // input_cloud_with_velocity: x,y,z,nx,ny,nz,intensity per point
  std::vector<float> rotated;
  // velocityEstimationThread's function-local static (:1511): the previous frame's possibly-dynamic clusters
  struct ClusterFeature {  // :70-81
    float center_x = 0.f, center_y = 0.f, center_z = 0.f;
    int   point_num = 0, match_cluster_seq = -1;
    float vx = -10000.f, vy = -10000.f, vz = -10000.f, v = 0.f, intensity = 0.f;
  };
  std::vector<ClusterFeature> clusters_last;
  int                         dbg_clusters = 0, dbg_dynamic = 0, dbg_matched = 0;
  int                dbg_voxel_full = 0, dbg_pyr_full = 0, dbg_out = 0, dbg_vel_draws = 0;

  float *slot(int v, int p) { return &store[((size_t)v * S + p) * 9]; }
};

// Eigen generic quaternion product a*b (Eigen/src/Geometry/Quaternion.h, quat_product<..., generic>)
inline void qmul(const float a[4], const float b[4], float o[4]) {  // w,x,y,z
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
  o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}
// dsp_dynamic.h:1391-1411: att * (0,v) * att.inverse(); inverse = conjugate / squaredNorm
inline void rotate(const float *v, const float *q, float *o) {
  float vq[4] = {0.f, v[0], v[1], v[2]}, t[4], r[4];
  float n2    = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  float inv[4] = {q[0] / n2, -q[1] / n2, -q[2] / n2, -q[3] / n2};
  qmul(q, vq, t);
  qmul(t, inv, r);
  o[0] = r[1];
  o[1] = r[2];
  o[2] = r[3];
}

inline float dot3(float x, float y, float z, const float *n) { return x * n[0] + y * n[1] + z * n[2]; }

// :1413-1430
int inArea(Dsp &d, float x, float y, float z) {
  return dot3(x, y, z, &d.bp_h[0]) >= 0.f && dot3(x, y, z, &d.bp_h[d.nph * 3]) <= 0.f &&
         dot3(x, y, z, &d.bp_v[0]) <= 0.f && dot3(x, y, z, &d.bp_v[d.npv * 3]) >= 0.f;
}
// :1432-1452
int findH(Dsp &d, float x, float y, float z) {
  float last = 1.f;
  for (int i = 0; i < d.nph; i++) {
    float t = dot3(x, y, z, &d.bp_h[(i + 1) * 3]);
    if (last * t <= 0.f) return i;
    last = t;
  }
  return -1;
}
// :1454-1474
int findV(Dsp &d, float x, float y, float z) {
  float last = -1.f;
  for (int j = 0; j < d.npv; j++) {
    float t = dot3(x, y, z, &d.bp_v[(j + 1) * 3]);
    if (last * t <= 0.f) return j;
    last = t;
  }
  return -1;
}
// :1152-1166 (+ ifParticleIsOut :1198-1204)
int voxelIndex(Dsp &d, float px, float py, float pz, int &index) {
  if (px >= d.hx || px <= -d.hx || py >= d.hy || py <= -d.hy || pz >= d.hz || pz <= -d.hz) return 0;
  int x = (int)((px + d.hx) / d.res);
  int y = (int)((py + d.hy) / d.res);
  int z = (int)((pz + d.hz) / d.res);
  index = z * d.spec.W * d.spec.L + y * d.spec.L + x;
  if (index < 0 || index >= d.V) return 0;
  return 1;
}
// :1380-1389
float queryPdf(Dsp &d, float x, float mu, float sigma) {
  float c = (x - mu) / sigma;
  if (c > 9.9f)
    c = 9.9f;
  else if (c < -9.9f)
    c = -9.9f;
  return d.pdf[(int)(c * 1000 + 10000)];
}
float velGauss(Dsp &d) {  // :1259-1266
  float v = d.vg[d.vseq];
  d.vseq += 1;
  if (d.vseq >= (int)d.vg.size()) d.vseq = 0;
  d.dbg_vel_draws++;
  return v;
}
float posGauss(Dsp &d) {  // :1241-1248
  float v = d.pg[d.pseq];
  d.pseq += 1;
  if (d.pseq >= (int)d.pg.size()) d.pseq = 0;
  return v;
}
float randFloat(Dsp &d, float lo, float hi) {  // :1682-1684, rand() replaced by the table
  int r = d.rnd[d.rseq];
  d.rseq += 1;
  if (d.rseq >= (int)d.rnd.size()) d.rseq = 0;
  return lo + static_cast<float>(r) / (static_cast<float>(RAND_MAX / (hi - lo)));
}

// :1271-1290
int addAParticle(Dsp &d, const float p[7], int vi) {  // p: px,py,pz,vx,vy,vz,weight
  for (int i = 0; i < d.S; i++) {
    float *s = d.slot(vi, i);
    if (s[0] < 0.1f) {
      s[0] = 15.f;
      s[1] = p[3];
      s[2] = p[4];
      s[3] = p[5];
      s[4] = p[0];
      s[5] = p[1];
      s[6] = p[2];
      s[7] = p[6];
      s[8] = d.update_time;
      return 1;
    }
  }
  return 0;
}

// :1295-1372
int moveParticle(Dsp &d, int nv, int cv, int cp, float *ori) {
  int ni = cp;
  if (nv != cv) {
    ori[0]    = 0.f;
    int moved = 0;
    for (int i = 0; i < d.S; ++i) {
      float *s = d.slot(nv, i);
      if (s[0] < 0.1f) {
        ni    = i;
        moved = 1;
        s[0]  = 7.f;
        for (int k = 1; k < 9; ++k) s[k] = ori[k];
        break;
      }
    }
    if (!moved) return -1;
  }
  float *s = d.slot(nv, ni);
  if (inArea(d, s[4], s[5], s[6])) {
    int h  = findH(d, s[4], s[5], s[6]);
    int v  = findV(d, s[4], s[5], s[6]);
    int pi = h * d.npv + v;
    int ok = 0;
    for (int j = 0; j < d.SP; j++) {
      int *e = &d.pyr[((size_t)pi * d.SP + j) * 3];
      if (e[0] == 0) {
        e[0] |= 1;
        e[1] = nv;
        e[2] = ni;
        ok   = 1;
        break;
      }
    }
    if (!ok) {
      s[0] = 0.f;
      return -2;
    }
    if (fabs(s[1] * s[2] * s[3]) < 1e-6) {
    } else {
      s[1] += velGauss(d);
      s[2] += velGauss(d);
      s[3] = 0.f;
    }
  }
  return 1;
}

// :663-748 (LIMIT_MOVEMENT_IN_XY_PLANE = 1, CONSIDER_LOCALIZATION_UNCERTAINTY undefined)
void mapPrediction(Dsp &d, float ox, float oy, float oz, float dt) {
  d.update_time += dt;
  d.update_counter += 1;
  for (size_t i = 0; i < d.pyr.size(); i += 3) d.pyr[i] &= 0;
  for (int v = 0; v < d.V; ++v) {
    for (int p = 0; p < d.S; p++) {
      float *s = d.slot(v, p);
      if (s[0] > 0.1f && s[0] < 6.f) {
        s[0] = 1.f;
        if (fabs(s[1] * s[2] * s[3]) < 1e-6) {
        } else {
          s[1] += velGauss(d);
          s[2] += velGauss(d);
          s[3] += velGauss(d);
        }
        s[3] = 0.f;
        s[4] += dt * s[1] + ox;
        s[5] += dt * s[2] + oy;
        s[6] += dt * s[3] + oz;
        int nv;
        if (voxelIndex(d, s[4], s[5], s[6], nv)) {
          int f = moveParticle(d, nv, v, p, s);
          if (f == -2) {
            d.dbg_pyr_full++;
            continue;
          } else if (f == -1) {
            d.dbg_voxel_full++;
            continue;
          }
        } else {
          s[0] = 0.f;
          d.dbg_out++;
        }
      }
    }
  }
}

// :750-849
void mapUpdate(Dsp &d) {
  const int OM = d.P.obs_max_per_pyramid;
  float     sig = d.P.sigma_observation, Pd = d.P.p_detection;
  for (int i = 0; i < d.NP; ++i) {
    for (int j = 0; j < d.nobs[i]; ++j) {
      float *o = &d.pc[((size_t)i * OM + j) * 5];
      for (int n = 0; n < d.nbr[i * 10]; ++n) {
        int pi = d.nbr[i * 10 + n + 1];
        for (int q = 0; q < d.SP; ++q) {
          int *e = &d.pyr[((size_t)pi * d.SP + q) * 3];
          if (e[0] & 1) {
            float *s  = d.slot(e[1], e[2]);
            float  gk = queryPdf(d, s[4], o[0], sig) * queryPdf(d, s[5], o[1], sig) *
                       queryPdf(d, s[6], o[2], sig);
            o[3] += Pd * s[7] * gk;
          }
        }
      }
      o[3] += (d.expected_new_born + d.P.kappa);
    }
  }
  for (int i = 0; i < d.NP; i++) {
    for (int q = 0; q < d.SP; q++) {
      int *e = &d.pyr[((size_t)i * d.SP + q) * 3];
      if (e[0] & 1) {
        int    nn = d.nbr[i * 10];
        float *s  = d.slot(e[1], e[2]);
        float  px = s[4], py = s[5], pz = s[6];
        float  len = sqrtf(px * px + py * py + pz * pz);
        if (d.maxlen[i] > 0.f && len > d.maxlen[i] + d.P.obstacle_thickness) continue;
        float sum = 0.f;
        for (int n = 0; n < nn; ++n) {
          int ni = d.nbr[i * 10 + n + 1];
          for (int z = 0; z < d.nobs[ni]; ++z) {
            float *o  = &d.pc[((size_t)ni * OM + z) * 5];
            float  gk = queryPdf(d, px, o[0], sig) * queryPdf(d, py, o[1], sig) * queryPdf(d, pz, o[2], sig);
            sum += Pd * gk / o[3];
          }
        }
        s[7] *= ((1 - Pd) + sum);
        s[8] = d.update_time;
      }
    }
  }
}

// :852-990
void newBorn(Dsp &d) {
  const int OM   = d.P.obs_max_per_pyramid;
  float     norm = 0.f;
  for (int i = 0; i < d.NP; i++)
    for (int j = 0; j < d.nobs[i]; j++) norm += 1.f / d.pc[((size_t)i * OM + j) * 5 + 3];
  float     w_new      = d.P.newborn_weight * norm;
  const int nb         = d.P.newborn_num;
  const int min_static = (int)((float)nb * 0.15f);
  const int model_gen  = (int)((float)nb * 0.8f);
  for (size_t k = 0; k + 7 <= d.born.size(); k += 7) {
    const float *pt = &d.born[k];
    float        cx = pt[0] - d.cur[0], cy = pt[1] - d.cur[1], cz = pt[2] - d.cur[2];
    int          vi;
    float        ws = 0.f, wd = 0.f, wsd = 0.f;
    if (voxelIndex(d, cx, cy, cz, vi)) {
      for (int kk = 0; kk < d.S; ++kk) {
        float *s = d.slot(vi, kk);
        if (s[0] > 0.9f && s[0] < 14.f) {
          float va = fabs(s[1]) + fabs(s[2]) + fabs(s[3]);
          if (va < 0.1f)
            ws += s[7];
          else if (va < 0.5f)
            wsd += s[7];
          else
            wd += s[7];
        }
      }
    } else {
      continue;
    }
    float tot = ws + wd + wsd;
    float m_s = ws / tot, m_d = wd / tot, m_sd = wsd / tot;
    float p_s = (m_s + m_s + m_sd) * 0.5f, p_d = (m_d + m_d + m_sd) * 0.5f;
    float np  = p_s + p_d;
    float psn = p_s / np;
    // (int)(float) of NaN is INT_MIN on x86 (cvttss2si); max() with min_static hides it.  Made
    // explicit here so that the result does not depend on the conversion instruction.
    float f_static = (float)model_gen * psn;
    int   n_static = (f_static == f_static) ? (int)f_static : min_static;
    n_static       = n_static > min_static ? n_static : min_static;
    for (int p = 0; p < nb; p++) {
      float q[7];
      q[0] = cx + posGauss(d);
      q[1] = cy + posGauss(d);
      q[2] = cz + posGauss(d);
      int qi;
      if (voxelIndex(d, q[0], q[1], q[2], qi)) {
        if (p < n_static) {
          q[3] = q[4] = q[5] = 0.f;
        } else if (pt[3] > -100.f && p < model_gen) {
          if (pt[6] > 0.01f) {
            q[3] = pt[3] + 4 * velGauss(d);
            q[4] = pt[4] + 4 * velGauss(d);
            q[5] = pt[5] + 4 * velGauss(d);
          } else {
            q[3] = q[4] = q[5] = 0.f;
          }
        } else {
          if (pt[6] > 0.01f) {
            q[3] = randFloat(d, -1.5f, 1.5f);
            q[4] = randFloat(d, -1.5f, 1.5f);
            q[5] = randFloat(d, -0.5f, 0.5f);
          } else {
            q[3] = q[4] = q[5] = 0.f;
          }
        }
        q[5] = 0.f;
        q[6] = w_new;
        addAParticle(d, q, qi);
      }
    }
  }
}

// :993-1130
void occupancyAndResample(Dsp &d) {
  const int MAXP = d.P.max_particle_num_voxel;
  for (int v = 0; v < d.V; ++v) {
    float wsum = 0.f, vxs = 0.f, vys = 0.f, vzs = 0.f;
    int   n = 0, n_old = 0;
    for (int p = 0; p < d.S; p++) {
      float *s = d.slot(v, p);
      if (s[0] > 0.1f) {
        if (s[7] < 1e-3) {
          s[0] = 0.f;
        } else {
          if (s[0] < 10.f) {
            ++n_old;
            vxs += s[1];
            vys += s[2];
            vzs += s[3];
            for (int t = 0; t < d.T; ++t) {
              float pt = d.P.prediction_times[t];
              float fx = s[4] + s[1] * pt, fy = s[5] + s[2] * pt, fz = s[6] + s[3] * pt;
              int   pi;
              if (voxelIndex(d, fx, fy, fz, pi)) d.objnum[(size_t)pi * d.OD + 4 + t] += s[7];
            }
          }
          s[0] = 1.f;
          ++n;
          wsum += s[7];
        }
      }
    }
    float *o = &d.objnum[(size_t)v * d.OD];
    o[0]     = wsum;
    if (n_old > 0) {
      o[1] = vxs / (float)n_old;
      o[2] = vys / (float)n_old;
      o[3] = vzs / (float)n_old;
    } else {
      o[1] = o[2] = o[3] = 0.f;
    }
    if (n < 5) continue;
    int   n_after = n > MAXP ? MAXP : n;
    float w_after = wsum / (float)n_after;
    float acc_ori = 0.f, acc_new = w_after * 0.5f;
    for (int p = 0; p < d.S; ++p) {
      float *s = d.slot(v, p);
      if (s[0] > 0.7f) {
        acc_ori += s[7];
        if (acc_ori > acc_new) {
          s[7] = w_after;
          acc_new += w_after;
          int full = 0, p_i = 0;
          while (acc_ori > acc_new) {
            int found = 0;
            if (!full) {
              for (; p_i < d.S; ++p_i) {
                float *c = d.slot(v, p_i);
                if (c[0] < 0.1f) {
                  c[0] = 0.6f;
                  for (int k = 1; k < 9; k++) c[k] = s[k];
                  found = 1;
                  break;
                }
              }
            }
            if (!found) {
              s[7] += w_after;
              full = 1;
            }
            acc_new += w_after;
          }
        } else {
          s[0] = 0.f;
        }
      }
    }
  }
}

}  // namespace

extern "C" {

// DSPMap::DSPMap + setInitParameters (:118-163, 566-632) with the RiskVoxel::init settings.
void *orc_dsp_create(const SogmSpec *spec, const SogmDspParams *P, const float *p_gauss,
                     const float *v_gauss, int n_gauss, const int32_t *rand_tab, int n_rand) {
  Dsp *d  = new Dsp;
  d->spec = *spec;
  d->P    = *P;
  d->V    = spec->L * spec->W * spec->H;
  d->T    = spec->T;
  d->OD   = 4 + d->T;
  d->S    = P->max_particle_num_voxel * 2;                      // SAFE_PARTICLE_NUM_VOXEL
  const int ar = P->angle_resolution;
  const int pyramid_num = 360 * 180 / ar / ar;                  // PYRAMID_NUM
  const int safe_num    = (int)(d->V * P->max_particle_num_voxel + 1e5);  // SAFE_PARTICLE_NUM
  d->SP  = safe_num / pyramid_num * 2;                          // SAFE_PARTICLE_NUM_PYRAMID
  d->nph = P->half_fov_h * 2 / ar;
  d->npv = P->half_fov_v * 2 / ar;
  d->NP  = d->nph * d->npv;
  d->res = spec->resolution;
  d->hx  = (d->res * (float)spec->L) * 0.5f;
  d->hy  = (d->res * (float)spec->W) * 0.5f;
  d->hz  = (d->res * (float)spec->H) * 0.5f;
  d->store.assign((size_t)d->V * d->S * 9, 0.f);
  d->objnum.assign((size_t)d->V * d->OD, 0.f);
  d->pyr.assign((size_t)d->NP * d->SP * 3, 0);
  d->nbr.assign((size_t)d->NP * 10, 0);
  d->pc.assign((size_t)d->NP * P->obs_max_per_pyramid * 5, 0.f);
  d->nobs.assign(d->NP, 0);
  d->maxlen.assign(d->NP, -1.f);
  const float arr = (float)ar / 180.f * 3.14159265358979323846f;  // angle_resolution_rad :589
  d->bp_ori_h.resize((d->nph + 1) * 3);
  d->bp_ori_v.resize((d->npv + 1) * 3);
  d->bp_h = d->bp_ori_h;
  d->bp_v = d->bp_ori_v;
  int hs = -P->half_fov_h / ar, he = -hs;
  for (int i = hs; i <= he; i++) {
    d->bp_ori_h[(i + he) * 3 + 0] = -sinf((float)i * arr);
    d->bp_ori_h[(i + he) * 3 + 1] = cosf((float)i * arr);
    d->bp_ori_h[(i + he) * 3 + 2] = 0.f;
  }
  int vs = -P->half_fov_v / ar, ve = -vs;
  for (int i = vs; i <= ve; i++) {
    d->bp_ori_v[(i + ve) * 3 + 0] = sinf((float)i * arr);
    d->bp_ori_v[(i + ve) * 3 + 1] = 0.f;
    d->bp_ori_v[(i + ve) * 3 + 2] = cosf((float)i * arr);
  }
  for (int i = 0; i < d->NP; i++) {  // findPyramidNeighborIndexInFOV :1206-1227
    int h0 = i / d->npv, v0 = i % d->npv, n = 0;
    for (int a = -1; a <= 1; ++a)
      for (int b = -1; b <= 1; ++b) {
        int h = h0 + a, v = v0 + b;
        if (h >= 0 && h < d->nph && v >= 0 && v < d->npv) d->nbr[i * 10 + 1 + n++] = h * d->npv + v;
      }
    d->nbr[i * 10] = n;
  }
  d->pdf.resize(20000);  // calculateNormalPDFBuffer :1373-1378, standardNormalPDF :1368-1371
  for (int i = 0; i < 20000; ++i) {
    float value = (float)(i - 10000) * 0.001f;
    d->pdf[i]   = (1.f / (sqrtf(2.f * 1.57079632679489661923f))) * expf(-powf(value, 2) / (2));
  }
  d->pg.assign(p_gauss, p_gauss + n_gauss);
  d->vg.assign(v_gauss, v_gauss + n_gauss);
  d->rnd.assign(rand_tab, rand_tab + n_rand);
  return d;
}

void orc_dsp_destroy(void *h) { delete (Dsp *)h; }

// Minimum-cost assignment of rows to columns (rows <= cols), potentials form of the Hungarian method with a fixed
// scan order.  Stands for munkres-cpp's Munkres<float>::solve (external, /usr/local/lib/libmunkres.a, absent):
// any exact solver returns the same assignment when the optimum is unique, which it is for real-valued cluster
// distances; ties are resolved by this scan order on both sides (oracle and kernel).  asg[r] = column of row r.
void assign_min_cost(const std::vector<float> &cost, int rows, int cols, std::vector<int> &asg) {
  const float      INF = 3.0e38f;
  std::vector<float> u(rows + 1, 0.f), v(cols + 1, 0.f), minv(cols + 1);
  std::vector<int>   p(cols + 1, 0), way(cols + 1, 0);
  std::vector<char>  used(cols + 1);
  for (int i = 1; i <= rows; ++i) {
    p[0]   = i;
    int j0 = 0;
    for (int j = 0; j <= cols; ++j) {
      minv[j] = INF;
      used[j] = 0;
    }
    do {
      used[j0]    = 1;
      const int i0 = p[j0];
      float     delta = INF;
      int       j1 = 0;
      for (int j = 1; j <= cols; ++j)
        if (!used[j]) {
          const float cur = cost[(size_t)(i0 - 1) * cols + (j - 1)] - u[i0] - v[j];
          if (cur < minv[j]) {
            minv[j] = cur;
            way[j]  = j0;
          }
          if (minv[j] < delta) {
            delta = minv[j];
            j1    = j;
          }
        }
      for (int j = 0; j <= cols; ++j)
        if (used[j]) {
          u[p[j]] += delta;
          v[j] -= delta;
        } else {
          minv[j] -= delta;
        }
      j0 = j1;
    } while (p[j0] != 0);
    do {
      const int j1 = way[j0];
      p[j0]        = p[j1];
      j0           = j1;
    } while (j0);
  }
  asg.assign(rows, -1);
  for (int j = 1; j <= cols; ++j)
    if (p[j] > 0) asg[p[j] - 1] = j - 1;
}

// velocityEstimationThread (dsp_dynamic.h:1487-1678): ground split, pcl::EuclideanClusterExtraction (tolerance
// 2 x voxel_filtered_resolution, 5 <= size <= 10000), per-cluster centre, association with the previous frame's
// possibly-dynamic clusters (gated distance cost, optimal assignment), velocity = centre displacement / dt, and
// input_cloud_with_velocity = [points of possibly-dynamic clusters, cluster by cluster][ground points][points of
// static clusters].  Third-party pieces restated from their published behaviour (PCL / FLANN / munkres-cpp are
// absent here: parity unpinned):
//   * pcl::extractEuclideanClusters: seeds in index order, breadth-first growth by radius search, a cluster's
//     indices sorted ascending, clusters below min size dropped (their points stay "processed");
//   * radius search = FLANN L2_Simple squared distance in float, accepted when strictly below
//     (float)(tolerance^2) with the tolerance held as double (KdTreeFLANN::radiusSearch);
//   * EuclideanClusterExtraction::extract then orders the clusters with
//     std::sort(clusters.rbegin(), clusters.rend(), comparePointClusters) (size ascending over the REVERSED range
//     = largest first; equal sizes in whatever order libstdc++'s introsort leaves them — called here directly);
//   * the visualisation intensity generateRandomFloat(0.1, 1) (:1531) is replaced by the constant 0.55 and draws
//     nothing from rand(): in the reference that draw races with the filter thread's own rand() calls
//     (std::thread at :305); only "intensity > 0.01" is ever consumed (:903-913).
void velocityEstimation(Dsp &d) {
  const int n = (int)d.rotated.size() / 3;
  if (n == 0) return;  // :1488
  d.born.clear();
  const float vres = 0.15f;  // voxel_filtered_resolution (:105)
  std::vector<float> stat;   // static_points xyz
  std::vector<float> ng;     // non_ground_points xyz
  for (int s = 0; s < n; ++s) {
    const float x = d.rotated[s * 3] + d.cur[0], y = d.rotated[s * 3 + 1] + d.cur[1], z = d.rotated[s * 3 + 2] + d.cur[2];
    std::vector<float> &dst = z > vres ? ng : stat;
    dst.push_back(x);
    dst.push_back(y);
    dst.push_back(z);
  }
  std::vector<Dsp::ClusterFeature> dyn;
  const int m = (int)ng.size() / 3;
  if (m > 0) {
    const double tol = (double)(2 * vres);
    const float  r2  = (float)(tol * tol);
    std::vector<std::vector<int>> clusters;
    std::vector<char>             processed(m, 0);
    for (int i = 0; i < m; ++i) {
      if (processed[i]) continue;
      std::vector<int> q(1, i);
      processed[i] = 1;
      for (size_t sq = 0; sq < q.size(); ++sq) {
        const float *a = &ng[(size_t)q[sq] * 3];
        for (int j = 0; j < m; ++j) {
          if (processed[j]) continue;
          const float *b  = &ng[(size_t)j * 3];
          float        d2 = 0.f;
          for (int k = 0; k < 3; ++k) {
            const float df = a[k] - b[k];
            d2 += df * df;
          }
          if (d2 < r2) {
            processed[j] = 1;
            q.push_back(j);
          }
        }
      }
      if (q.size() >= 5 && q.size() <= 10000) {
        std::sort(q.begin(), q.end());
        clusters.push_back(q);
      }
    }
    std::sort(clusters.rbegin(), clusters.rend(),
              [](const std::vector<int> &a, const std::vector<int> &b) { return a.size() < b.size(); });
    d.dbg_clusters = (int)clusters.size();
    std::vector<char> possibly_dynamic;
    for (const auto &ci : clusters) {
      Dsp::ClusterFeature c;
      c.intensity = 0.55f;
      for (int idx : ci) {
        c.center_x += ng[(size_t)idx * 3];
        c.center_y += ng[(size_t)idx * 3 + 1];
        c.center_z += ng[(size_t)idx * 3 + 2];
        ++c.point_num;
      }
      c.center_x /= (float)c.point_num;
      c.center_y /= (float)c.point_num;
      c.center_z /= (float)c.point_num;
      if (ci.size() > 200 || c.center_z > 1.5) {  // DYNAMIC_CLUSTER_MAX_POINT_NUM / _MAX_CENTER_HEIGHT
        for (int idx : ci)
          for (int k = 0; k < 3; ++k) stat.push_back(ng[(size_t)idx * 3 + k]);
        possibly_dynamic.push_back(0);
      } else {
        dyn.push_back(c);
        possibly_dynamic.push_back(1);
      }
    }
    d.dbg_dynamic = (int)dyn.size();
    d.dbg_matched = 0;
    const float distance_gate = 1.5f, maximum_velocity = 5.f;
    const int   point_num_gate = 100;
    if (!d.clusters_last.empty() && !dyn.empty() && d.dt_last > 0.00001 && d.dt_last < 10.0) {
      const int R = (int)dyn.size(), C = (int)d.clusters_last.size();
      std::vector<float> cost((size_t)R * C), gate((size_t)R * C);
      for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) {
          const Dsp::ClusterFeature &a = dyn[r], &b = d.clusters_last[c];
          const float sq = (a.center_x - b.center_x) * (a.center_x - b.center_x) +
                           (a.center_y - b.center_y) * (a.center_y - b.center_y) +
                           (a.center_z - b.center_z) * (a.center_z - b.center_z);
          const float dist = sqrtf(sq);
          if (abs(a.point_num - b.point_num) > point_num_gate || dist >= distance_gate) {
            gate[(size_t)r * C + c] = 0.f;
            cost[(size_t)r * C + c] = distance_gate * 5000.f;
          } else {
            gate[(size_t)r * C + c] = 1.f;
            cost[(size_t)r * C + c] = dist / distance_gate * 1000.f;
          }
        }
      // optimal assignment (rows <= cols for the solver: transpose when there are more new clusters than old)
      std::vector<int> col_of_row(R, -1);
      if (R <= C) {
        assign_min_cost(cost, R, C, col_of_row);
      } else {
        std::vector<float> ct((size_t)C * R);
        for (int r = 0; r < R; ++r)
          for (int c = 0; c < C; ++c) ct[(size_t)c * R + r] = cost[(size_t)r * C + c];
        std::vector<int> row_of_col;
        assign_min_cost(ct, C, R, row_of_col);
        for (int c = 0; c < C; ++c)
          if (row_of_col[c] >= 0) col_of_row[row_of_col[c]] = c;
      }
      for (int r = 0; r < R; ++r) {
        const int c = col_of_row[r];
        if (c < 0 || !(gate[(size_t)r * C + c] > 0.01f)) continue;  // :1591
        Dsp::ClusterFeature &f = dyn[r];
        f.match_cluster_seq = c;
        f.vx = (f.center_x - d.clusters_last[c].center_x) / d.dt_last;
        f.vy = (f.center_y - d.clusters_last[c].center_y) / d.dt_last;
        f.vz = (f.center_z - d.clusters_last[c].center_z) / d.dt_last;
        f.v  = sqrtf(f.vx * f.vx + f.vy * f.vy + f.vz * f.vz);
        f.intensity = d.clusters_last[c].intensity;
        if (f.v > maximum_velocity) {
          f.v  = 0.f;
          f.vx = f.vy = f.vz = 0.f;
        }
        ++d.dbg_matched;
      }
    }
    // velocity allocation to points (:1625-1650)
    int dseq = 0;
    for (size_t ci = 0; ci < clusters.size(); ++ci) {
      if (!possibly_dynamic[ci]) continue;
      for (int idx : clusters[ci]) {
        for (int k = 0; k < 3; ++k) d.born.push_back(ng[(size_t)idx * 3 + k]);
        d.born.push_back(dyn[dseq].vx);
        d.born.push_back(dyn[dseq].vy);
        d.born.push_back(dyn[dseq].vz);
        d.born.push_back(dyn[dseq].intensity);
      }
      ++dseq;
    }
  }
  for (size_t k = 0; k + 3 <= stat.size(); k += 3) {  // :1653-1663
    for (int j = 0; j < 3; ++j) d.born.push_back(stat[k + j]);
    for (int j = 0; j < 4; ++j) d.born.push_back(0.f);
  }
  d.clusters_last = dyn;  // :1665 (also when empty)
}

// DSPMap::update (:165-364).  pts: n x 3 sensor-frame points; labels: n x 4 {vx,vy,vz,intensity}
// standing for velocityEstimationThread's output (points are born in the given order), or NULL: the
// velocity estimation runs here (velocityEstimation above).
int orc_dsp_update(void *h, int n, const float *pts, const float *labels, float px, float py,
                   float pz, double stamp, float qw, float qx, float qy, float qz) {
  Dsp &d = *(Dsp *)h;
  if (d.first) {
    d.last_p[0] = px;
    d.last_p[1] = py;
    d.last_p[2] = pz;
    d.last_t    = stamp;
    d.first     = false;
  }
  if (fabs(qw) > 1.001f || fabs(qx) > 1.001f || fabs(qy) > 1.001f || fabs(qz) > 1.001f) return 0;
  float dx = px - d.last_p[0], dy = py - d.last_p[1], dz = pz - d.last_p[2];
  float dt = (float)(stamp - d.last_t);
  if (fabs(dx) > 10.f || fabs(dy) > 10.f || fabs(dz) > 10.f || dt < 0.f || dt > 10.f) return 0;
  d.cur[0] = d.last_p[0] = px;
  d.cur[1] = d.last_p[1] = py;
  d.cur[2] = d.last_p[2] = pz;
  d.last_t               = stamp;
  d.dt_last              = dt;
  d.quat[0]              = qw;
  d.quat[1]              = qx;
  d.quat[2]              = qy;
  d.quat[3]              = qz;
  for (int i = 0; i < d.nph + 1; i++) rotate(&d.bp_ori_h[i * 3], d.quat, &d.bp_h[i * 3]);
  for (int j = 0; j < d.npv + 1; j++) rotate(&d.bp_ori_v[j * 3], d.quat, &d.bp_v[j * 3]);
  const int OM = d.P.obs_max_per_pyramid;
  for (int i = 0; i < d.NP; i++) {
    d.nobs[i]   = 0;
    d.maxlen[i] = -1.f;
  }
  d.rotated.clear();
  int valid = 0;
  for (int s = 0; s < n; ++s) {
    float r[3];
    rotate(&pts[s * 3], d.quat, r);
    d.rotated.push_back(r[0]);
    d.rotated.push_back(r[1]);
    d.rotated.push_back(r[2]);
    if (inArea(d, r[0], r[1], r[2])) {
      int    ph = findH(d, r[0], r[1], r[2]), pv = findV(d, r[0], r[1], r[2]);
      int    pi = ph * d.npv + pv, seq = d.nobs[pi];
      float  len = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
      float *o   = &d.pc[((size_t)pi * OM + seq) * 5];
      o[0]       = r[0];
      o[1]       = r[1];
      o[2]       = r[2];
      o[3]       = 0.f;
      o[4]       = len;
      if (d.maxlen[pi] < len) d.maxlen[pi] = len;
      d.nobs[pi] += 1;
      if (d.nobs[pi] >= OM) d.nobs[pi] = OM - 1;
      ++valid;
    }
  }
  d.expected_new_born          = d.P.newborn_weight * (float)valid * (float)d.P.newborn_num;
  d.new_born_each_object_weight = d.P.newborn_weight * (float)d.P.newborn_num;
  // velocityEstimationThread (:1487-1678) replaced by the supplied labels; like the reference it
  // leaves input_cloud_with_velocity untouched when the cloud is empty (:1488).
  if (!labels) {
    velocityEstimation(d);
  } else if (n > 0) {
    d.born.clear();
    for (int s = 0; s < n; ++s) {
      d.born.push_back(d.rotated[s * 3 + 0] + d.cur[0]);
      d.born.push_back(d.rotated[s * 3 + 1] + d.cur[1]);
      d.born.push_back(d.rotated[s * 3 + 2] + d.cur[2]);
      for (int k = 0; k < 4; k++) d.born.push_back(labels[s * 4 + k]);
    }
  }
  mapPrediction(d, -dx, -dy, -dz, dt);
  if (n >= 0) mapUpdate(d);
  if (n >= 0) newBorn(d);
  occupancyAndResample(d);
  return 1;
}

// getOccupancyMapWithFutureStatus (:445-469) + the zeroing loop of RiskVoxel::publishMap
// (risk_voxel.cpp:141-153).  out_vt [V][T].  The reference indexes temp_risk_map with
// getVoxelIndex(Vector3i offset) (map.h:176-179) for offsets in [-inf_step, inf_step]^3, i.e. with
// possibly NEGATIVE indices (undefined behaviour); only in-bounds indices are applied here.
int orc_dsp_publish(void *h, float *out_vt, float threshold, int inf_step) {
  Dsp &d   = *(Dsp *)h;
  int  occ = 0;
  for (int i = 0; i < d.V; i++) {
    float *o = &d.objnum[(size_t)i * d.OD];
    if (o[0] > threshold) ++occ;
    for (int n = 0; n < d.T; ++n) out_vt[(size_t)i * d.T + n] = o[4 + n];
    for (int j = 4; j < d.OD; ++j) o[j] = 0.f;
  }
  for (int x = -inf_step; x <= inf_step; x++)
    for (int y = -inf_step; y <= inf_step; y++)
      for (int z = -inf_step; z <= inf_step; z++) {
        int idx = z * d.spec.L * d.spec.W + y * d.spec.L + x;
        if (idx < 0 || idx >= d.V) continue;
        for (int t = 0; t < 3 && t < d.T; t++) out_vt[(size_t)idx * d.T + t] = 0.f;
      }
  return occ;
}

// raw state for parity checks: store [V][S][9], objnum [V][4+T]
void orc_dsp_state(void *h, float *store, float *objnum, int *counters) {
  Dsp &d = *(Dsp *)h;
  if (store) memcpy(store, d.store.data(), d.store.size() * sizeof(float));
  if (objnum) memcpy(objnum, d.objnum.data(), d.objnum.size() * sizeof(float));
  if (counters) {
    counters[0] = d.dbg_voxel_full;
    counters[1] = d.dbg_pyr_full;
    counters[2] = d.dbg_out;
    counters[3] = d.dbg_vel_draws;
    counters[4] = d.pseq;
    counters[5] = d.vseq;
    counters[6] = d.rseq;
    counters[7] = d.S;
    counters[8] = d.SP;
    counters[9] = d.NP;
  }
}
// per-pyramid observation table of the last update: nobs[NP], pc [NP][OM][5]
// input_cloud_with_velocity of the last update: rows {x, y, z, vx, vy, vz, intensity}; returns the row count;
// counters3 = {clusters, possibly dynamic, matched}
int orc_dsp_born(void *h, float *out, int cap, int *counters3) {
  Dsp      &d = *(Dsp *)h;
  const int n = (int)d.born.size() / 7;
  if (out)
    for (int i = 0; i < n * 7 && i < cap * 7; ++i) out[i] = d.born[i];
  if (counters3) {
    counters3[0] = d.dbg_clusters;
    counters3[1] = d.dbg_dynamic;
    counters3[2] = d.dbg_matched;
  }
  return n;
}

void orc_dsp_observations(void *h, int *nobs, float *pc, float *maxlen) {
  Dsp &d = *(Dsp *)h;
  if (nobs) memcpy(nobs, d.nobs.data(), d.nobs.size() * sizeof(int));
  if (pc) memcpy(pc, d.pc.data(), d.pc.size() * sizeof(float));
  if (maxlen) memcpy(maxlen, d.maxlen.data(), d.maxlen.size() * sizeof(float));
}
}
