#!/usr/bin/env python
"""An INDEPENDENT second restatement of the neighbour overlay at the end of FakeParticleRiskVoxel::updateMap — other
agents' body particles along their shared Bezier trajectories, added to the future slices — written straight from the
reference's text WITHOUT reading oracle/.  The increments on an empty parity-size grid are committed as
tests/golden/overlay_independent.json; a CPU test holds the C++ oracle (`orc_project_neighbours`) to them cell by cell, a
GPU test holds `sogm_project_neighbours` to them directly (tests/test_overlay_independent.py).  Two separately written
readings have to agree; nothing here pins either to the reference itself.

Restated, block by block:
  the overlay loop                 plan_env/src/fake_particle_risk_voxel.cpp:175-218  (is_swarm_traj_valid carried over the
                                   slices; t = stamp + time_resolution_ (float) * t_idx; particles - pose_ in double)
  addParticlesToRiskMap            plan_env/src/risk_base.cpp:199-208 (cast to float, strict isInRange, += risk)
  ParticleATC::getWaypoints        traj_coordinator/src/particles.cpp:316-344 (strict time_start < t0 < time_end; a
                                   trajectory that has not started yet yields NO waypoints; one that has ended yields its
                                   last point and `false`)
  ParticleATC::getParticlesWithRisk  :346-422, replan_risk_rate 0 (sim_fake.yaml:88): every particle weighs 1.0; no
                                   waypoints = `false` = the agent is dropped for every later slice of this update
  loadParticles(pts, pt, idx)      :301-306    particlesCallback :89-108 (Point32: the offsets are float32 cast to double)
  initEgoParticles                 :62-75      trajectoryCallback :131-191 (time_end = time_start + the durations, summed)
  Bezier / BernsteinPiece          traj_utils/include/traj_utils/bernstein.hpp:38-47,164-177; src/bernstein.cpp:25-34,
                                   80-88,184-196 (piece i spans [t, t + t_i] with t accumulated; s = (t - t0_) / (tf_ - t0_);
                                   pos = cpts' * A * [1 s s^2 s^3 s^4]', evaluated left to right)
What the text does that one might not expect, all kept: a neighbour whose trajectory starts AT or after the slice time of
slice 0 is dropped for the whole update (`time_start < t0` is strict and an empty waypoint list returns false), even if it
starts before slice 1; a trajectory that ends inside the horizon is overlaid up to its last slice strictly before the end.
Run from the repo root:   python tests/golden/make_overlay_fixture.py
"""
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
spec_ = importlib.util.spec_from_file_location("make_map_fixture", os.path.join(HERE, "make_map_fixture.py"))
mm = importlib.util.module_from_spec(spec_)
spec_.loader.exec_module(mm)
f32 = np.float32
L, W, H, T = mm.L, mm.W, mm.H, mm.T
A4 = [[1, -4, 6, -4, 1], [0, 4, -12, 12, -4], [0, 0, 6, -12, 6], [0, 0, 0, 4, -4], [0, 0, 0, 0, 1]]


def ego_particles(sx=0.4, sy=0.4, sz=0.45):
    out, step = [], 0.15
    x = -sx / 2
    while x <= sx / 2:
        y = -sy / 2
        while y <= sy / 2:
            z = -sz / 2
            while z <= sz / 2:
                out.append((x, y, z))
                z += step
            y += step
        x += step
    return [tuple(float(f32(v)) for v in p) for p in out]     # as received: geometry_msgs/Point32


class Traj:
    def __init__(self, rec):
        self.id, self.t_start = rec["id"], rec["time_start"]
        self.dur = list(rec["duration"])
        t_end = self.t_start
        for d in self.dur:
            t_end += d
        self.t_end = t_end
        self.T = 0.0
        for d in self.dur:
            self.T += d
        self.pieces, t = [], 0.0
        for i, d in enumerate(self.dur):
            self.pieces.append((rec["cpts"][5 * i:5 * i + 5], t, t + d))
            t += d

    def locate(self, t):
        for i, d in enumerate(self.dur):
            t -= d
            if t < 0:
                return i
        return len(self.dur) - 1

    def pos(self, t):
        cp, t0, tf = self.pieces[self.locate(t)]
        s = (t - t0) / (tf - t0)
        S = [1.0] + [s ** i for i in range(1, 5)]
        out = []
        for r in range(3):
            Mr = []
            for c in range(5):
                a = 0.0
                for k in range(5):
                    a += cp[k][r] * A4[k][c]
                Mr.append(a)
            p = 0.0
            for c in range(5):
                p += Mr[c] * S[c]
            out.append(p)
        return out


def waypoints(trajs, body, idx, ego, t0):
    tr = next((x for x in trajs if x.id == idx), None)
    if tr is None or idx == ego:
        return False, []
    if tr.t_start < t0 and tr.t_end > t0:
        p = tr.pos(t0 - tr.t_start)
        return True, [(p[0] + e[0], p[1] + e[1], p[2] + e[2]) for e in body]
    if tr.t_start > t0:
        return True, []
    if tr.t_end < t0:
        return False, [tuple(tr.pos(tr.T))]
    return False, []


def overlay(m, trajs, body, n_rbts, ego, stamp):
    inc = {}
    valid = [True] * n_rbts
    pose = [float(v) for v in m.pose]
    for k in range(T):
        t = stamp + float(f32(mm.TIME_RES * f32(k)))
        parts = []
        for i in range(n_rbts):
            if i == ego or not valid[i]:
                continue
            ok, pts = waypoints(trajs, body, i, ego, t)
            if not ok or not pts:                     # getParticlesWithRisk: false, nothing handed back
                valid[i] = False
                continue
            parts.extend(pts)                         # rate 0: the way points themselves, weight 1.0
        for p in parts:
            q = np.array([p[0] - pose[0], p[1] - pose[1], p[2] - pose[2]], np.float64).astype(f32)
            if not m.in_range_f(q):
                continue
            v = m.voxel_index_f(q)
            if v < L * W * H:
                inc[(v, k)] = inc.get((v, k), 0) + 1
    return inc


def main():
    rng = np.random.default_rng(0x0E1A)
    body = ego_particles()
    stamp = 250.0
    cases = []
    for c in range(4):
        m = mm.FakeMap()
        m.pose = (rng.uniform(-3, 3, 3) * np.array([1, 1, 0.1]) + np.array([0, 0, 1.0])).astype(f32)
        n_rbts, ego = 7, int(rng.integers(0, 7))
        recs = []
        for i in range(n_rbts):
            if i == 5:
                continue                               # no record from drone 5
            M = int(rng.integers(1, 6))
            dur = rng.uniform(0.2, 0.6, M).tolist()
            p0 = m.pose.astype(np.float64) + rng.uniform(-3.5, 3.5, 3) * np.array([1, 1, 0.2])
            v = rng.uniform(-1.5, 1.5, 3) * np.array([1, 1, 0.1])
            cp, tt = [], 0.0
            for j in range(M):
                for q in range(5):
                    cp.append((p0 + v * (tt + dur[j] * q / 4.0) + rng.normal(0, 0.05, 3) * (0 < q < 4)).tolist())
                tt += dur[j]
            ts = stamp - float(rng.uniform(0.05, 0.5))
            if i == 1:
                ts = stamp                              # starts exactly at the map stamp: dropped at slice 0
            if i == 2:
                ts = stamp + 0.1                        # starts between slice 0 and 1: dropped at slice 0 as well
            if i == 3:
                dur = [0.25, 0.3]                       # ends inside the horizon
                cp, ts = cp[:10] if len(cp) >= 10 else (cp + cp)[:10], stamp - 0.1
            recs.append({"id": i, "time_start": ts, "duration": dur, "cpts": cp})
        rng.shuffle(recs)                               # the table order is not the id order
        trajs = [Traj(r) for r in recs]
        inc = overlay(m, trajs, body, n_rbts, ego, stamp)
        cells = sorted((int(v), int(k), int(n)) for (v, k), n in inc.items())
        per_slice = [sum(n for _, k, n in cells if k == kk) for kk in range(T)]
        print(f"case {c}: ego {ego}, {len(cells)} cells, particles per slice {per_slice}")
        cases.append({"pose": [float(x) for x in m.pose], "stamp": stamp, "n_robots": n_rbts, "ego": ego, "records": recs,
                      "cells": cells, "particles_per_slice": per_slice})
    out = {"what": "neighbour overlay of FakeParticleRiskVoxel::updateMap restated independently (tests/golden/make_overlay_fixture.py): "
                   "cells = (voxel index, slice, particles added) on an empty grid; body = float32-rounded ego particles",
           "grid": [L, W, H, T], "body": body, "cases": cases}
    path = os.path.join(HERE, "overlay_independent.json")
    json.dump(out, open(path, "w"))
    print("written", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
