#!/usr/bin/env python
"""An INDEPENDENT second restatement of the particle SOGM — dsp_map::DSPMap::update with its four stages (observation
binning, mapPrediction, mapUpdate, mapAddNewBornParticlesByObservation, mapOccupancyCalculationAndResample) as RiskVoxel
configures it — written from the reference's text (plan_env/include/plan_env/dsp_dynamic.h, map_parameters.h,
src/risk_voxel.cpp:42-50) WITHOUT reading oracle/.  The time(0)-seeded tables (Gaussian position / velocity noise, rand()) and
the output of velocityEstimationThread (the new-born list: points + {vx, vy, vz, intensity}) are INPUTS, as they are for
sogm_update_dsp with labels.  Four updates of a synthetic flight on the 66 x 66 x 20 x 6 map; after every update the particle
store (flag / velocity / position of every slot: SHA-256 of the float bits; weights: sampled values and sums, compared to a
tolerance because the normal-PDF table and the pyramid normals come from this interpreter's exp / sin, not glibc's), the
observation tables and the per-voxel object numbers are committed as tests/golden/dsp_independent.json.  A CPU test holds
the C++ oracle to them, a GPU test holds sogm_update_dsp to them directly (tests/test_dsp_independent.py).

Restated, block by block (dsp_dynamic.h):
  setInitParameters            :564-625  half map lengths, the FOV boundary-plane normals (h: (-sin, cos, 0), v: (sin, 0, cos) of
                                         i x angle_resolution_rad), pyramid neighbours (findPyramidNeighborIndexInFOV :1205-1225),
                                         the 20 000-entry normal-PDF table (:1373-1385: 1 / sqrtf(2 x pi/2) x expf(-v^2 / 2))
  update                       :165-330  static last pose / stamp (first call: deltas 0), the quaternion / jump checks, boundary
                                         normals rotated by the sensor quaternion (rotateVectorByQuaternion :1396-1418: Eigen's
                                         quaternion product q v q^-1 in float), observation binning (ifInPyramidsArea :1424,
                                         findPointPyramid*Index :1440-1480; the count is capped at 99 AFTER the write: the last
                                         slot is overwritten), expected_new_born_objects
  mapPrediction                :663-748  flags in (0.1, 6): reset to 1, LIMIT_MOVEMENT_IN_XY_PLANE: vz = 0 (so |vx vy vz| < 1e-6:
                                         no velocity noise is ever drawn), position += dt v + odom delta (no localisation noise:
                                         CONSIDER_LOCALIZATION_UNCERTAINTY is not defined in this header), getParticleVoxelsIndex
                                         (:1149-1165, strict bounds :1196-1203, float division, truncation), moveParticle
                                         (:1288-1365: first empty slot of the new voxel gets flag 7; first empty slot of the
                                         pyramid list; no slot -> the particle vanishes)
  mapUpdate                    :750-851  C_k per observation over the neighbour pyramids' listed particles in list order, then
                                         the weight update per listed particle (occlusion test, sum over the neighbours'
                                         observations in order); queryNormalPDF :1387-1394
  mapAddNewBornParticlesByObservation :853-992  the Dempster-Shafer split from the voxel's weights (0/0 = NaN -> the casts and
                                         max() keep the minimum of 3 static particles), 20 particles per point: position noise
                                         (3 draws each, always), static / labelled / random velocity branches (4 x velocity noise;
                                         generateRandomFloat :1682 = min + rand() / (float)(RAND_MAX / (max - min))), vz = 0,
                                         addAParticle (:1268-1286: first slot with flag < 0.1 gets flag 15)
  mapOccupancyCalculationAndResample :994-1130  weights < 1e-3 removed, object numbers, the future status per prediction time,
                                         flags back to 1, the systematic resample with copies into the first empty slots (0.6)
Run from the repo root:   python tests/golden/make_dsp_fixture.py      (a few minutes: the sweeps are Python loops)
"""
import hashlib
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from helpers import dsp_fixture_inputs  # noqa: E402  (the INPUTS: shared with the tests, which never import this script)

f32 = np.float32
NX, NY, NZ, T = 66, 66, 20, 6
RES = f32(0.15)
MAXP = 7
S = 2 * MAXP                               # SAFE_PARTICLE_NUM_VOXEL
V = NX * NY * NZ
HFH, HFV, ARES = 43, 29, 1
NPH, NPV = HFH * 2 // ARES, HFV * 2 // ARES
NP = NPH * NPV
SP = (int(V * MAXP + 1e5) // (360 * 180 // ARES // ARES)) * 2    # SAFE_PARTICLE_NUM_PYRAMID
OMAX = 100
THICK = f32(0.3)
PRED = [f32(0.3), f32(0.6), f32(0.9), f32(1.2), f32(1.5), f32(1.8)]
RAND_MAX = 2147483647


def fl(x):
    return f32(x)


class DSP:
    def __init__(self, tables, sigma_obs=0.05, p_det=0.95, kappa=0.01, nb_weight=0.0001, nb_num=20):
        self.pg, self.vg, self.rnd = tables
        self.pseq = self.vseq = self.rseq = 0
        self.sigma = f32(sigma_obs)
        self.pdet, self.kappa, self.nbw, self.nbn = f32(p_det), f32(kappa), f32(nb_weight), nb_num
        self.hx = fl(fl(RES * f32(NX)) * f32(0.5))
        self.hy = fl(fl(RES * f32(NY)) * f32(0.5))
        self.hz = fl(fl(RES * f32(NZ)) * f32(0.5))
        ang = f32(float(f32(ARES) / f32(180.0)) * 3.14159265358979323846)     # (float / float) x a double literal, stored as float
        self.bh0 = [(f32(-math.sin(float(f32(i) * ang))), f32(math.cos(float(f32(i) * ang))), f32(0)) for i in range(-HFH, HFH + 1)]
        self.bv0 = [(f32(math.sin(float(f32(i) * ang))), f32(0), f32(math.cos(float(f32(i) * ang)))) for i in range(-HFV, HFV + 1)]
        self.nei = []
        for i in range(NP):
            h0, v0 = i // NPV, i % NPV
            self.nei.append([h * NPV + v for h in (h0 - 1, h0, h0 + 1) for v in (v0 - 1, v0, v0 + 1) if 0 <= h < NPH and 0 <= v < NPV])
        c = f32(1.0) / f32(math.sqrt(float(f32(2.0 * 1.57079632679489661923))))
        vals = (np.arange(20000, dtype=np.int64) - 10000).astype(f32) * f32(0.001)
        self.pdf = (c * np.exp(-(vals.astype(np.float64) ** 2).astype(f32).astype(np.float64) / 2.0).astype(f32)).astype(f32)
        self.store = np.zeros((V, S, 9), f32)
        self.obj = np.zeros((V, 4 + T), f32)
        self.pyr = np.zeros((NP, SP, 3), np.int32)
        self.pc = np.zeros((NP, OMAX, 5), f32)
        self.nobs = np.zeros(NP, np.int32)
        self.maxlen = np.full(NP, -1.0, f32)
        self.last = None
        self.update_time = f32(0)
        self.cur = [f32(0)] * 3

    # ---- helpers
    def rotate(self, v, q):
        w, x, y, z = [f32(c) for c in q]

        def mul(a, b):
            aw, ax, ay, az = a
            bw, bx, by, bz = b
            return (fl(fl(fl(aw * bw) - fl(ax * bx)) - fl(ay * by)) - fl(az * bz),
                    fl(fl(fl(aw * bx) + fl(ax * bw)) + fl(ay * bz)) - fl(az * by),
                    fl(fl(fl(aw * by) + fl(ay * bw)) + fl(az * bx)) - fl(ax * bz),
                    fl(fl(fl(aw * bz) + fl(az * bw)) + fl(ax * by)) - fl(ay * bx))
        n2 = fl(fl(fl(x * x) + fl(y * y)) + fl(z * z)) + fl(w * w)
        inv = (fl(w / n2), fl(-x / n2), fl(-y / n2), fl(-z / n2))
        r = mul(mul((w, x, y, z), (f32(0), f32(v[0]), f32(v[1]), f32(v[2]))), inv)
        return (f32(r[1]), f32(r[2]), f32(r[3]))

    @staticmethod
    def dot(x, y, z, n):
        return fl(fl(fl(x * n[0]) + fl(y * n[1])) + fl(z * n[2]))

    def in_fov(self, x, y, z):
        return (self.dot(x, y, z, self.bh[0]) >= 0 and self.dot(x, y, z, self.bh[NPH]) <= 0 and
                self.dot(x, y, z, self.bv[0]) <= 0 and self.dot(x, y, z, self.bv[NPV]) >= 0)

    def pyr_h(self, x, y, z):
        last = f32(1.0)
        for i in range(NPH):
            t = self.dot(x, y, z, self.bh[i + 1])
            if fl(last * t) <= 0:
                return i
            last = t
        return -1

    def pyr_v(self, x, y, z):
        last = f32(-1.0)
        for j in range(NPV):
            t = self.dot(x, y, z, self.bv[j + 1])
            if fl(last * t) <= 0:
                return j
            last = t
        return -1

    def voxel_index(self, px, py, pz):
        if px >= self.hx or px <= -self.hx or py >= self.hy or py <= -self.hy or pz >= self.hz or pz <= -self.hz:
            return -1
        x = int(fl(fl(px + self.hx) / RES))
        y = int(fl(fl(py + self.hy) / RES))
        z = int(fl(fl(pz + self.hz) / RES))
        idx = z * NY * NX + y * NX + x
        return idx if 0 <= idx < V else -1

    def gauss_p(self):
        v = self.pg[self.pseq]
        self.pseq = (self.pseq + 1) % len(self.pg)
        return f32(v)

    def gauss_v(self):
        v = self.vg[self.vseq]
        self.vseq = (self.vseq + 1) % len(self.vg)
        return f32(v)

    def rand_float(self, lo, hi):
        r = int(self.rnd[self.rseq])
        self.rseq = (self.rseq + 1) % len(self.rnd)
        return fl(f32(lo) + fl(f32(r) / fl(f32(RAND_MAX) / fl(f32(hi) - f32(lo)))))

    def pdf_q(self, x, mu):
        c = fl(fl(x - mu) / self.sigma)
        c = np.where(c > f32(9.9), f32(9.9), np.where(c < f32(-9.9), f32(-9.9), c)).astype(f32)
        return self.pdf[(c * f32(1000) + f32(10000)).astype(f32).astype(np.int64)]

    # ---- DSPMap::update
    def update(self, points, labels, pos, quat, stamp):
        pos = [f32(v) for v in pos]
        if self.last is None:
            self.last = (pos[0], pos[1], pos[2], float(stamp))
        if any(abs(float(f32(c))) > 1.001 for c in quat):
            return 0
        d = [fl(pos[i] - self.last[i]) for i in range(3)]
        dt = f32(float(stamp) - self.last[3])
        if any(abs(float(v)) > 10.0 for v in d) or dt < 0 or dt > 10:
            return 0
        self.cur = list(pos)
        self.last = (pos[0], pos[1], pos[2], float(stamp))
        self.bh = [self.rotate(n, quat) for n in self.bh0]
        self.bv = [self.rotate(n, quat) for n in self.bv0]
        self.nobs[:] = 0
        self.maxlen[:] = -1.0
        rot, valid = [], 0
        for p in np.asarray(points, f32).reshape(-1, 3):
            r = self.rotate(p, quat)
            rot.append(r)
            if self.in_fov(*r):
                pi = self.pyr_h(*r) * NPV + self.pyr_v(*r)
                k = self.nobs[pi]
                ln = f32(math.sqrt(float(fl(fl(fl(r[0] * r[0]) + fl(r[1] * r[1])) + fl(r[2] * r[2])))))
                self.pc[pi, k] = (r[0], r[1], r[2], f32(0), ln)
                if self.maxlen[pi] < ln:
                    self.maxlen[pi] = ln
                self.nobs[pi] += 1
                if self.nobs[pi] >= OMAX:
                    self.nobs[pi] = OMAX - 1
                valid += 1
        self.expected = fl(fl(self.nbw * f32(valid)) * f32(self.nbn))
        # the new-born list as velocityEstimationThread leaves it: the rotated point + the sensor position, the label
        born = [((fl(r[0] + self.cur[0]), fl(r[1] + self.cur[1]), fl(r[2] + self.cur[2])), lab) for r, lab in zip(rot, np.asarray(labels, f32).reshape(-1, 4))]
        self.predict(fl(-d[0]), fl(-d[1]), fl(-d[2]), dt)
        self.map_update()
        self.add_new_born(born)
        self.occupancy_and_resample()
        return 1

    def predict(self, ox, oy, oz, dt):
        self.update_time = fl(self.update_time + dt)
        self.pyr[:, :, 0] &= 0
        st = self.store
        vs, ps = np.nonzero((st[:, :, 0] > f32(0.1)) & (st[:, :, 0] < f32(6.0)))     # (row-major: v ascending, then p)
        for v, p in zip(vs.tolist(), ps.tolist()):
            q = st[v, p]
            if not (q[0] > f32(0.1) and q[0] < f32(6.0)):
                continue
            q[0] = 1.0
            if abs(float(fl(fl(q[1] * q[2]) * q[3]))) < 1e-6:
                pass
            else:
                q[1] = fl(q[1] + self.gauss_v())
                q[2] = fl(q[2] + self.gauss_v())
                q[3] = fl(q[3] + self.gauss_v())
            q[3] = 0.0
            q[4] = fl(q[4] + fl(fl(dt * q[1]) + ox))
            q[5] = fl(q[5] + fl(fl(dt * q[2]) + oy))
            q[6] = fl(q[6] + fl(fl(dt * q[3]) + oz))
            nv = self.voxel_index(q[4], q[5], q[6])
            if nv >= 0:
                self.move(nv, v, p)
            else:
                q[0] = 0.0

    def move(self, nv, v, p):
        st = self.store
        ni = p
        if nv != v:
            src = st[v, p].copy()
            st[v, p, 0] = 0.0
            for i in range(S):
                if st[nv, i, 0] < f32(0.1):
                    ni = i
                    st[nv, i, 0] = 7.0
                    st[nv, i, 1:9] = src[1:9]
                    break
            else:
                return -1
        q = st[nv, ni]
        if self.in_fov(q[4], q[5], q[6]):
            pi = self.pyr_h(q[4], q[5], q[6]) * NPV + self.pyr_v(q[4], q[5], q[6])
            for j in range(SP):
                if self.pyr[pi, j, 0] == 0:
                    self.pyr[pi, j] = (1, nv, ni)
                    break
            else:
                q[0] = 0.0
                return -2
            if abs(float(fl(fl(q[1] * q[2]) * q[3]))) < 1e-6:
                pass
            else:
                q[1] = fl(q[1] + self.gauss_v())
                q[2] = fl(q[2] + self.gauss_v())
                q[3] = 0.0
        return 1

    def map_update(self):
        st, pc = self.store, self.pc
        # C_k: per observation, the neighbour pyramids in list order, their listed particles in slot order — one float sum each
        for i in np.nonzero(self.nobs > 0)[0].tolist():
            n = int(self.nobs[i])
            acc = pc[i, :n, 3].copy()
            for pi in self.nei[i]:
                for s in range(SP):
                    if self.pyr[pi, s, 0] & 1:
                        q = st[self.pyr[pi, s, 1], self.pyr[pi, s, 2]]
                        gk = (self.pdf_q(q[4], pc[i, :n, 0]) * self.pdf_q(q[5], pc[i, :n, 1])).astype(f32)
                        gk = (gk * self.pdf_q(q[6], pc[i, :n, 2])).astype(f32)
                        acc = (acc + ((self.pdet * q[7]).astype(f32) * gk).astype(f32)).astype(f32)
            pc[i, :n, 3] = (acc + fl(self.expected + self.kappa)).astype(f32)
        for i in range(NP):
            for s in range(SP):
                if not (self.pyr[i, s, 0] & 1):
                    continue
                q = st[self.pyr[i, s, 1], self.pyr[i, s, 2]]
                px, py, pz = q[4], q[5], q[6]
                dist = f32(math.sqrt(float(fl(fl(fl(px * px) + fl(py * py)) + fl(pz * pz)))))
                if self.maxlen[i] > 0 and dist > fl(self.maxlen[i] + THICK):
                    continue
                tot = f32(0)
                for ni in self.nei[i]:
                    n = int(self.nobs[ni])
                    if n == 0:
                        continue
                    gk = (self.pdf_q(px, pc[ni, :n, 0]) * self.pdf_q(py, pc[ni, :n, 1])).astype(f32)
                    gk = (gk * self.pdf_q(pz, pc[ni, :n, 2])).astype(f32)
                    terms = ((self.pdet * gk).astype(f32) / pc[ni, :n, 3]).astype(f32)
                    for t_ in terms:
                        tot = fl(tot + t_)
                q[7] = fl(q[7] * fl(fl(f32(1) - self.pdet) + tot))
                q[8] = self.update_time

    def add_new_born(self, born):
        st = self.store
        norm = f32(0)
        for i in np.nonzero(self.nobs > 0)[0].tolist():
            for j in range(int(self.nobs[i])):
                norm = fl(norm + fl(f32(1) / self.pc[i, j, 3]))
        w_new = fl(self.nbw * norm)
        n_min, n_model = int(f32(self.nbn) * f32(0.15)), int(f32(self.nbn) * f32(0.8))
        with np.errstate(invalid="ignore", divide="ignore"):
            for (pt, lab) in born:
                pcx, pcy, pcz = fl(pt[0] - self.cur[0]), fl(pt[1] - self.cur[1]), fl(pt[2] - self.cur[2])
                vi = self.voxel_index(pcx, pcy, pcz)
                if vi < 0:
                    continue
                ws = wd = wsd = f32(0)
                for k in range(S):
                    q = st[vi, k]
                    if q[0] > f32(0.9) and q[0] < f32(14.0):
                        vabs = fl(fl(abs(q[1]) + abs(q[2])) + abs(q[3]))
                        if vabs < f32(0.1):
                            ws = fl(ws + q[7])
                        elif vabs < f32(0.5):
                            wsd = fl(wsd + q[7])
                        else:
                            wd = fl(wd + q[7])
                tw = fl(fl(ws + wd) + wsd)
                m_s, m_d, m_sd = fl(ws / tw), fl(wd / tw), fl(wsd / tw)
                p_s = fl(fl(fl(m_s + m_s) + m_sd) * f32(0.5))
                p_d = fl(fl(fl(m_d + m_d) + m_sd) * f32(0.5))
                p_s_n = fl(p_s / fl(p_s + p_d))
                x = fl(f32(n_model) * p_s_n)
                n_static = -2147483648 if not np.isfinite(x) else int(x)      # (int)NaN on x86-64
                n_static = max(n_min, n_static)
                for p in range(self.nbn):
                    ppx, ppy, ppz = fl(pcx + self.gauss_p()), fl(pcy + self.gauss_p()), fl(pcz + self.gauss_p())
                    nv = self.voxel_index(ppx, ppy, ppz)
                    if nv < 0:
                        continue
                    if p < n_static:
                        vx = vy = vz = f32(0)
                    elif lab[0] > f32(-100.0) and p < n_model:
                        if lab[3] > f32(0.01):
                            vx = fl(lab[0] + fl(f32(4) * self.gauss_v()))
                            vy = fl(lab[1] + fl(f32(4) * self.gauss_v()))
                            vz = fl(lab[2] + fl(f32(4) * self.gauss_v()))
                        else:
                            vx = vy = vz = f32(0)
                    else:
                        if lab[3] > f32(0.01):
                            vx, vy, vz = self.rand_float(-1.5, 1.5), self.rand_float(-1.5, 1.5), self.rand_float(-0.5, 0.5)
                        else:
                            vx = vy = vz = f32(0)
                    vz = f32(0)
                    for i in range(S):
                        if st[nv, i, 0] < f32(0.1):
                            st[nv, i] = (15.0, vx, vy, vz, ppx, ppy, ppz, w_new, self.update_time)
                            break

    def occupancy_and_resample(self):
        st, obj = self.store, self.obj
        obj[:, :4] = 0      # (entries 0-3 are assigned for every voxel below; 4.. accumulate until the consumer clears them)
        for v in np.nonzero((st[:, :, 0] > f32(0.1)).any(axis=1))[0].tolist():
            wsum = vxs = vys = vzs = f32(0)
            n = old = 0
            for p in range(S):
                q = st[v, p]
                if q[0] > f32(0.1):
                    if float(q[7]) < 1e-3:
                        q[0] = 0.0
                    else:
                        if q[0] < f32(10.0):
                            old += 1
                            vxs, vys, vzs = fl(vxs + q[1]), fl(vys + q[2]), fl(vzs + q[3])
                            for t in range(T):
                                fv = self.voxel_index(fl(q[4] + fl(q[1] * PRED[t])), fl(q[5] + fl(q[2] * PRED[t])), fl(q[6] + fl(q[3] * PRED[t])))
                                if fv >= 0:
                                    obj[fv, 4 + t] = fl(obj[fv, 4 + t] + q[7])
                        q[0] = 1.0
                        n += 1
                        wsum = fl(wsum + q[7])
            obj[v, 0] = wsum
            if old > 0:
                obj[v, 1], obj[v, 2], obj[v, 3] = fl(vxs / f32(old)), fl(vys / f32(old)), fl(vzs / f32(old))
            if n < 5:
                continue
            n_after = MAXP if n > MAXP else n
            w_after = fl(wsum / f32(n_after))
            acc_o, acc_n = f32(0), fl(w_after * f32(0.5))
            for p in range(S):
                q = st[v, p]
                if q[0] > f32(0.7):
                    acc_o = fl(acc_o + q[7])
                    if acc_o > acc_n:
                        q[7] = w_after
                        acc_n = fl(acc_n + w_after)
                        full, pi = False, 0
                        while acc_o > acc_n:
                            found = False
                            if not full:
                                while pi < S:
                                    if st[v, pi, 0] < f32(0.1):
                                        st[v, pi, 0] = 0.6
                                        st[v, pi, 1:9] = q[1:9]
                                        found = True
                                        break
                                    pi += 1
                            if not found:
                                q[7] = fl(q[7] + w_after)
                                full = True
                            acc_n = fl(acc_n + w_after)
                    else:
                        q[0] = 0.0


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    tables, seq = dsp_fixture_inputs()
    g = DSP(tables)
    out = []
    for k, s in enumerate(seq):
        ok = g.update(s["points"], s["labels"], s["pos"], s["quat"], s["stamp"])
        st = g.store
        occ = st[:, :, 0] > f32(0.1)
        w = st[:, :, 7][occ].astype(np.float64)
        idx = np.flatnonzero(occ.ravel())
        pick = idx[:: max(1, len(idx) // 300)][:300]
        rec = {"update": k, "ok": ok, "in_sha256": sha(np.concatenate([s["points"].ravel(), s["labels"].ravel(), np.asarray(s["pos"], f32), np.asarray(s["quat"], f32)])),
               "n_particles": int(occ.sum()), "flags_sha256": sha(st[:, :, 0]), "vel_pos_sha256": sha(np.where(occ[:, :, None], st[:, :, 1:7], 0)),
               "cursors": [g.pseq, g.vseq, g.rseq], "nobs_sha256": sha(g.nobs), "nobs_total": int(g.nobs.sum()),
               "obs_xyzl_sha256": sha(np.where((np.arange(OMAX)[None, :] < g.nobs[:, None])[:, :, None], g.pc[:, :, [0, 1, 2, 4]], 0)),
               "maxlen_sha256": sha(g.maxlen), "weight_sum": float(w.sum()), "weight_min": float(w.min()) if len(w) else 0.0,
               "weight_max": float(w.max()) if len(w) else 0.0,
               "weight_samples": [[int(i), float(st.reshape(-1, 9)[i, 7])] for i in pick],
               "ck_sum": float(np.where(np.arange(OMAX)[None, :] < g.nobs[:, None], g.pc[:, :, 3], 0).astype(np.float64).sum()),
               "obj0_sum": float(g.obj[:, 0].astype(np.float64).sum()), "future_sum": g.obj[:, 4:].astype(np.float64).sum(axis=0).tolist(),
               "occupied_voxels": int((g.obj[:, 0] > 0).sum())}
        g.obj[:, 4:] = 0      # the consumer (getOccupancyMapWithFutureStatus, :454-476) clears the future status after reading it
        out.append(rec)
        print({k_: v for k_, v in rec.items() if k_ not in ("weight_samples",)}, flush=True)
    with open(os.path.join(HERE, "dsp_independent.json"), "w") as f:
        json.dump({"grid": [NX, NY, NZ, T], "updates": out}, f)


if __name__ == "__main__":
    main()
