#!/usr/bin/env python
"""An INDEPENDENT second restatement of getObstaclePoints(points, t_start, t_end, lower_corner, higher_corner) — the box scan
that feeds FIRI its obstacle points — written straight from the reference's text WITHOUT reading oracle/, run on the grids of
the independent map restatement (make_map_fixture.FakeMap: the same three poses as map_independent.json); counts, SHA-256
and the first points of every box are committed as tests/golden/obstacle_points_independent.json.  A CPU test holds the C++
oracle (`orc_obstacle_points`) to them and a GPU test holds `sogm_obstacle_points` to them directly
(tests/test_obstacle_points_independent.py).  Two separately written readings have to agree; nothing here pins either to
the reference itself.

Restated:
  MapBase::getObstaclePoints  plan_env/src/map.cpp:463-530      (the fake map's: plain threshold)
  RiskBase::getObstaclePoints plan_env/src/risk_base.cpp:295-337 (threshold lowered by risk_thres_vox_decay_ * j)
  MapBase::getVoxelPosition   plan_env/include/plan_env/map.h:186-194
As the text has it: every slice above the threshold emits the voxel's centre again (no break); the box corners are truncated
toward zero in DOUBLE arithmetic (double corner - float pose + float range, / float resolution), then clamped; the slice
loop is INCLUSIVE of idx_end and idx_end may equal PREDICTION_TIMES, so risk_maps_[i][T] is read — one past the row's end:
undefined behaviour that, in the flat float[VOXEL_NUM][PREDICTION_TIMES], lands on risk_maps_[i + 1][0].  The fixture holds
BOTH readings: "n" / "sha256" / "first" with slice T empty (what oracle and kernel document as their one deviation here,
oracle/map_oracle.cpp "deviation: the reference reads slice T") and "n_alias" = the count under the aliasing reading, so the
size of the deviation is on record (it only differs for windows that reach the end of the prediction horizon).
Run from the repo root:   python tests/golden/make_obstacle_points_fixture.py
"""
import hashlib
import importlib.util
import json
import math
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
VOX_DECAY = f32(0.2)     # risk_base.cpp:23


def trunc(x):
    return int(x)        # C's double -> int conversion: toward zero


def obstacle_points(m, stamp, t_start, t_end, lc, hc, decayed, alias):
    flat = m.risk.ravel()                       # float risk_maps_[VOXEL_NUM][PREDICTION_TIMES], row-major
    tres = float(mm.TIME_RES)
    i0 = math.floor((t_start - stamp) / tres)
    i1 = math.ceil((t_end - stamp) / tres)
    i0 = 0 if i0 < 0 else i0
    i0 = T if i0 > T else i0
    i1 = T if i1 > T else i1
    i1 = 0 if i1 < 0 else i1
    res, rng, pose = float(mm.RES), (float(m.rx), float(m.ry), float(m.rz)), [float(v) for v in m.pose]
    lo = [trunc((lc[k] - pose[k] + rng[k]) / res) for k in range(3)]
    hi = [trunc((hc[k] - pose[k] + rng[k]) / res) for k in range(3)]
    hi = [min(hi[0], L - 1), min(hi[1], W - 1), min(hi[2], H - 1)]
    lo = [max(v, 0) for v in lo]
    pts = []
    for z in range(lo[2], hi[2] + 1):
        for y in range(lo[1], hi[1] + 1):
            for x in range(lo[0], hi[0] + 1):
                i = x + y * L + z * L * W
                for j in range(i0, i1 + 1):
                    k = i * T + j
                    if j >= T and not alias:
                        continue
                    v = flat[k] if k < flat.size else f32(0.0)
                    thr = f32(mm.RISK_THRESHOLD - VOX_DECAY * f32(j)) if decayed else mm.RISK_THRESHOLD
                    if v > thr:
                        pts.append(m.voxel_position(i).astype(np.float64))
    return np.array(pts, np.float64).reshape(-1, 3)


def main():
    scene = importlib.import_module("pred-occ-planner_amd.scene")
    fx = json.load(open(os.path.join(HERE, "map_independent.json")))
    sc = scene.make_scene(fx["agents"], 4.95, seed=fx["seed"], moving=True)
    cyl = [{"type": 3, "x": float(r[0]), "y": float(r[1]), "w": float(r[2]), "vx": float(r[3]), "vy": float(r[4])}
           for r in sc["cylinders"]]
    rng = np.random.default_rng(0x0B57)
    stamp = 100.0
    cases = []
    for a, case in enumerate(fx["cases"]):
        m = mm.FakeMap()
        m.update_map(sc["cloud"], cyl, np.asarray(case["pose"], f32))
        occ = np.flatnonzero(m.risk.ravel()) // T
        boxes = []
        for b in range(14):
            c = m.voxel_position(int(occ[rng.integers(0, len(occ))])).astype(np.float64) + rng.uniform(-0.5, 0.5, 3)
            half = rng.uniform(0.2, 2.5, 3)
            lc, hc = c - half, c + half
            t0 = stamp + rng.uniform(-0.3, 1.0)
            t1 = t0 + rng.uniform(0.0, 0.6)
            if b == 0:      # a box beyond the map
                lc, hc = c + 50.0, c + 51.0
            if b == 1:      # a window beyond the last slice: idx_start = idx_end = PREDICTION_TIMES
                t0, t1 = stamp + 99.0, stamp + 100.0
            if b == 2:      # ends exactly on the last boundary: slices T - 1 and T
                t0, t1 = stamp + (T - 1) * float(mm.TIME_RES) + 0.01, stamp + T * float(mm.TIME_RES) + 0.5
            if b == 3:      # the whole map, the whole horizon
                lc, hc, t0, t1 = np.array(m.pose, np.float64) - 20.0, np.array(m.pose, np.float64) + 20.0, stamp - 1.0, stamp + 10.0
            if b == 4:      # before the map's stamp
                t0, t1 = stamp - 2.0, stamp - 1.0
            rec = {"lc": lc.tolist(), "hc": hc.tolist(), "t0": float(t0), "t1": float(t1)}
            for name, dec in (("base", False), ("risk", True)):
                p = obstacle_points(m, stamp, t0, t1, lc, hc, dec, False)
                rec[name] = {"n": int(len(p)), "sha256": hashlib.sha256(p.tobytes()).hexdigest(), "first": p[:4].tolist(),
                             "n_alias": int(len(obstacle_points(m, stamp, t0, t1, lc, hc, dec, True)))}
            boxes.append(rec)
        cases.append({"pose": case["pose"], "stamp": stamp, "boxes": boxes})
        print(f"agent {a}:", [(b["base"]["n"], b["base"]["n_alias"], b["risk"]["n"]) for b in boxes])
    out = {"what": "getObstaclePoints (boxed overload) restated independently on the independent map's grids "
                   "(tests/golden/make_obstacle_points_fixture.py); base = MapBase (fake map), risk = RiskBase (decayed threshold)",
           "seed": fx["seed"], "agents": fx["agents"], "grid": [L, W, H, T], "cases": cases}
    path = os.path.join(HERE, "obstacle_points_independent.json")
    json.dump(out, open(path, "w"))
    print("written", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
