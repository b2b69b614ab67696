#!/usr/bin/env python
"""An INDEPENDENT second restatement of MapBase::filterPointCloud (plan_env/src/map.cpp:107-132) — pcl::VoxelGrid, the
camera -> body axis swap, isInRange, the cap of 5000 — written from the reference's text and from the published algorithm of
pcl::VoxelGrid<PointXYZ>::applyFilter (PCL 1.10, filters/impl/voxel_grid.hpp: the library itself is absent here) WITHOUT
reading oracle/.  Outputs for six synthetic camera-frame clouds are committed as tests/golden/filter_independent.json; a CPU
test holds the C++ oracle to them, a GPU test holds sogm_filter_point_cloud to them directly (tests/test_filter_independent.py).
Two separately written readings have to agree; nothing here pins either to PCL.

Restated:
  sor.setLeafSize(filter_res_ x 3); sor.filter(*cloud_out)                             map.cpp:111-114
    inverse_leaf_size_ = 1 / leaf (float)                                              voxel_grid.h setLeafSize
    min / max over the FINITE points (getMinMax3D); min_b = floor(min * inv), max_b = floor(max * inv) (float product,
    floor, to int); div_b = max_b - min_b + 1; divb_mul = (1, div_b.x, div_b.x * div_b.y)
    every finite point: ijk = floor(p * inv) - min_b, idx = ijk . divb_mul; sort by idx; per run of equal idx ONE output
    point = the float centroid (a Vector4f sum of the run, divided by the float count), runs in ascending idx
    (the sort is std::sort on idx alone: the order INSIDE a run is unspecified, so is the last bit of a float sum — the
    sums here run in input order; the tests compare centroids to 1e-4 and everything else exactly)
  for every output point: x = p.z, y = -p.x, z = -p.y; isInRange: strict |.| < local_update_range (map.h:153-157), the range
    being MAP_LENGTH_VOXEL_NUM / 2.f * resolution_ for x / y and MAP_HEIGHT_VOXEL_NUM / 2 * resolution_ — an INTEGER
    division — for z (map.cpp:46-48); kept points appended until 5000 are there (:121-128)
Run from the repo root:   python tests/golden/make_filter_fixture.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32
L, W, H, RES, LEAF, CAP = 66, 66, 20, f32(0.15), f32(0.15), 5000   # the parity grid (config.make_spec("parity"))


sys.path.insert(0, os.path.dirname(HERE))
from helpers import filter_fixture_clouds as clouds  # noqa: E402  (the INPUTS: shared with the tests, which never import this script)


def voxel_grid(pts, leaf):
    inv = f32(1.0) / leaf
    fin = np.isfinite(pts).all(axis=1)
    p = pts[fin]
    if len(p) == 0:
        return np.zeros((0, 3), f32)
    mn, mx = p.min(axis=0), p.max(axis=0)
    min_b = np.floor(mn * inv).astype(np.int64)
    max_b = np.floor(mx * inv).astype(np.int64)
    div_b = max_b - min_b + 1
    mul = np.array([1, div_b[0], div_b[0] * div_b[1]], np.int64)
    ijk = np.floor(p * inv).astype(np.int64) - min_b
    idx = ijk @ mul
    order = np.argsort(idx, kind="stable")      # input order inside a leaf
    out = []
    i = 0
    while i < len(order):
        j = i
        s = np.zeros(3, f32)
        while j < len(order) and idx[order[j]] == idx[order[i]]:
            s = (s + p[order[j]]).astype(f32)
            j += 1
        out.append((s / f32(j - i)).astype(f32))
        i = j
    return np.asarray(out, f32).reshape(-1, 3)


def filter_point_cloud(raw, cap=CAP):
    rx, ry, rz = f32(L / 2.0) * RES, f32(W / 2.0) * RES, f32(H // 2) * RES
    kept = []
    for q in voxel_grid(raw, LEAF):
        x, y, z = q[2], -q[0], -q[1]
        if -rx < x < rx and -ry < y < ry and -rz < z < rz:
            kept.append((x, y, z))
            if len(kept) >= cap:
                break
    return np.asarray(kept, f32).reshape(-1, 3)


def main():
    cases = []
    for name, raw in clouds():
        out = filter_point_cloud(raw)
        small = filter_point_cloud(raw, cap=64)
        cases.append({"name": name, "n_in": int(len(raw)), "in_sha256": hashlib.sha256(raw.tobytes()).hexdigest(),
                      "n_out": int(len(out)), "n_out_cap64": int(len(small)),
                      # every output point rounded to 1e-3 (leaf 0.15: identifies the leaf and its order), and exact values of a few
                      "out_mm": np.round(out.astype(np.float64) * 1000).astype(np.int64).ravel().tolist()[:3 * 400],
                      "out_first": out[:8].astype(np.float64).ravel().tolist(),
                      "out_sum": out.astype(np.float64).sum(axis=0).tolist() if len(out) else [0.0, 0.0, 0.0]})
    with open(os.path.join(HERE, "filter_independent.json"), "w") as f:
        json.dump({"grid": [L, W, H], "resolution": float(RES), "leaf": float(LEAF), "cap": CAP, "cases": cases}, f)
    print("wrote filter_independent.json:", [(c["name"], c["n_in"], c["n_out"]) for c in cases])


if __name__ == "__main__":
    main()
