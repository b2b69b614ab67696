#!/usr/bin/env python
"""An INDEPENDENT second restatement of the depth front end — GridMap::projectDepthImage, raycastProcess (with RayCaster),
clearAndInflateLocalMap as updateOccupancyCallback runs them for one depth frame — written from the reference's text WITHOUT
reading oracle/.  Three frames of a moving camera on the small test map (12 x 12 x 3 m at 0.1 m) are flown through it; the
SHA-256 of the fp64 log-odds buffer and of the inflated occupancy, the local bounds and a few counts after every frame are
committed as tests/golden/gridmap_independent.json.  A CPU test holds the C++ oracle to them, a GPU test holds
sogm_gridmap_update to them directly (tests/test_gridmap_independent.py).  Nothing here pins either to the reference binary
(DESIGN.md section 4): two separately written readings have to agree, bit for bit.

Restated, block by block (plan_env/src/grid_map.cpp unless said otherwise):
  initMap derived values        :66-100   resolution_inv_, map_origin_ = (-x/2, -y/2, ground_height_), the logits, unknown_flag_
                                          0.01, map_voxel_num_ = ceil(size / res), buffers: log-odds at clamp_min - unknown_flag
  depthPoseCallback             :632-661  a camera outside the map (isInMap, 1e-4 margins, grid_map.h:359-370) updates nothing
  updateOccupancyCallback       :609-630  project, raycast, and clearAndInflateLocalMap only if the raycast ran
  projectDepthImage, filter on  :244-304  the first frame only sets has_first_depth_; samples v, u from the margin in steps of
                                          skip_pixel_; depth = *row_ptr * (1 / k); the pointer is ADVANCED before the zero test,
                                          which therefore reads the NEXT sample; < mindist: skipped; 0 or > maxdist: depth =
                                          max_ray_length_ + 0.1; pt = R * ((u - cx) d / fx, (v - cy) d / fy, d) + camera
  raycastProcess                :313-441  per projected point, in order: outside the map -> closetPointInMap (:443-460), clipped
                                          to max_ray_length_, a MISS at the end; inside and farther than max_ray_length_ -> clipped,
                                          a miss; else a HIT; setCacheOccupancy (:191-208: counters, first touch queues the voxel);
                                          the ray is skipped if its end voxel was a ray end before in this frame (flag_rayend_);
                                          RayCaster from the END point back to the camera (raycast.cpp:242-335: setInput, step,
                                          intbound, signum; the first voxel it returns is the end voxel itself, the camera's voxel
                                          is never returned), every voxel a miss, stop at the first voxel traversed before in this
                                          frame (flag_traverse_) — after marking it a miss; bounds from the (clipped) end points,
                                          the camera and ground_height_; the queue in first-touch order: hit if hits >= misses,
                                          the clamp / continue rules, "outside the local range -> reset to clamp_min first"
  clearAndInflateLocalMap       :462-566  the three shells of cleared cells, inflation by ceil(inflation / res) cells in a full
                                          cube (grid_map.h:391-420) through the FLAT address with only a range check on it (rows
                                          wrap), the virtual ceiling at floor((ceil - origin.z) * res_inv) - 1
Python floats are IEEE doubles and every expression keeps the reference's operation order (Eigen's fixed-size 3 x 3 times vector:
(r0 x + r1 y) + r2 z; squaredNorm (x x + y y) + z z; v / length * max + camera component-wise).
Run from the repo root:   python tests/golden/make_gridmap_fixture.py      (about a minute: the rays are walked in Python)
"""
import hashlib
import importlib
import json
import math
import os
import sys
from collections import deque

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from helpers import gridmap_fixture_frames, gridmap_fixture_params  # noqa: E402  (the INPUTS, shared with the tests)


def logit(x):
    return math.log(x / (1 - x))


class GridMapRestated:
    def __init__(self, p):
        self.p = p
        self.res = p["resolution"]
        self.res_inv = 1 / self.res
        xs, ys, zs = p["map_size"]
        self.ceil_h = p["virtual_ceil_height"]
        if self.ceil_h - p["ground_height"] > zs:
            self.ceil_h = p["ground_height"] + zs
        self.origin = (-xs / 2.0, -ys / 2.0, p["ground_height"])
        self.size = (xs, ys, zs)
        self.hit, self.miss = logit(p["p_hit"]), logit(p["p_miss"])
        self.cmin, self.cmax, self.occ_log = logit(p["p_min"]), logit(p["p_max"]), logit(p["p_occ"])
        self.unknown = 0.01
        self.nv = [int(math.ceil(self.size[i] / self.res)) for i in range(3)]
        self.bmin = self.origin
        self.bmax = tuple(self.origin[i] + self.size[i] for i in range(3))
        n = self.nv[0] * self.nv[1] * self.nv[2]
        self.occ = np.full(n, self.cmin - self.unknown, np.float64)
        self.inf = np.zeros(n, np.int8)
        self.cnt = np.zeros(n, np.int16)        # count_hit_and_miss_ (short)
        self.cnt_hit = np.zeros(n, np.int16)
        self.f_end = np.full(n, -1, np.int8)    # flag_rayend_ (char)
        self.f_trav = np.full(n, -1, np.int8)
        self.raycast_num = 0
        self.has_first = False
        self.lb_min, self.lb_max = [0, 0, 0], [0, 0, 0]
        self.queue = deque()

    # grid_map.h inline helpers
    def in_map(self, q):
        return not (q[0] < self.bmin[0] + 1e-4 or q[1] < self.bmin[1] + 1e-4 or q[2] < self.bmin[2] + 1e-4 or
                    q[0] > self.bmax[0] - 1e-4 or q[1] > self.bmax[1] - 1e-4 or q[2] > self.bmax[2] - 1e-4)

    def pos_to_index(self, q):
        return [int(math.floor((q[i] - self.origin[i]) * self.res_inv)) for i in range(3)]

    def addr(self, i):
        return i[0] * self.nv[1] * self.nv[2] + i[1] * self.nv[2] + i[2]

    def bound(self, i):
        return [max(min(i[k], self.nv[k] - 1), 0) for k in range(3)]

    def set_cache(self, q, occ):
        i = self.pos_to_index(q)
        a = self.addr(i)
        if a < 0 or a >= len(self.cnt):
            return -1        # (outside the arrays: undefined behaviour in the reference; no such touch in these frames)
        self.cnt[a] += 1
        if self.cnt[a] == 1:
            self.queue.append(i)
        if occ == 1:
            self.cnt_hit[a] += 1
        return a

    def closest_in_map(self, pt, cam):
        diff = [pt[i] - cam[i] for i in range(3)]
        max_tc = [self.bmax[i] - cam[i] for i in range(3)]
        min_tc = [self.bmin[i] - cam[i] for i in range(3)]
        min_t = 1000000
        for i in range(3):
            if abs(diff[i]) > 0:
                t1 = max_tc[i] / diff[i]
                if 0 < t1 < min_t:
                    min_t = t1
                t2 = min_tc[i] / diff[i]
                if 0 < t2 < min_t:
                    min_t = t2
        return [cam[i] + (min_t - 1e-3) * diff[i] for i in range(3)]

    def project(self, img, cam, R):
        p = self.p
        rows, cols = img.shape
        pts = []
        if not self.has_first:
            self.has_first = True
            return pts
        flat = img.ravel()
        m, skip = p["depth_filter_margin"], p["skip_pixel"]
        inv_factor = 1.0 / p["k_depth_scaling_factor"]
        for v in range(m, rows - m, skip):
            ptr = v * cols + m
            for u in range(m, cols - m, skip):
                depth = int(flat[ptr]) * inv_factor
                ptr += skip
                if flat[ptr] == 0:                      # (the NEXT sample's value)
                    depth = p["max_ray_length"] + 0.1
                elif depth < p["depth_filter_mindist"]:
                    continue
                elif depth > p["depth_filter_maxdist"]:
                    depth = p["max_ray_length"] + 0.1
                c0 = (u - p["cx"]) * depth / p["fx"]
                c1 = (v - p["cy"]) * depth / p["fy"]
                c2 = depth
                pts.append([(R[i][0] * c0 + R[i][1] * c1) + R[i][2] * c2 + cam[i] for i in range(3)])
        return pts

    def raycast(self, pts, cam):
        if not pts:
            return False
        p = self.p
        self.raycast_num += 1
        rn = np.int8(self.raycast_num) if self.raycast_num < 128 else None   # (char flags: never equal from frame 128 on)
        mn = [self.bmax[0], self.bmax[1], self.bmax[2]]
        mx = [self.bmin[0], self.bmin[1], self.bmin[2]]
        maxlen, res = p["max_ray_length"], self.res

        def norm(d):
            return math.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])

        for pt in pts:
            if not self.in_map(pt):
                pt = self.closest_in_map(pt, cam)
                d = [pt[i] - cam[i] for i in range(3)]
                length = norm(d)
                if length > maxlen:
                    pt = [d[i] / length * maxlen + cam[i] for i in range(3)]
                vox = self.set_cache(pt, 0)
            else:
                d = [pt[i] - cam[i] for i in range(3)]
                length = norm(d)
                if length > maxlen:
                    pt = [d[i] / length * maxlen + cam[i] for i in range(3)]
                    vox = self.set_cache(pt, 0)
                else:
                    vox = self.set_cache(pt, 1)
            for i in range(3):
                mx[i] = max(mx[i], pt[i])
                mn[i] = min(mn[i], pt[i])
            if vox >= 0 and rn is not None and self.f_end[vox] == rn:
                continue
            if vox >= 0:
                self.f_end[vox] = rn if rn is not None else np.int8(self.raycast_num & 0x7F)   # (the assignment truncates to char)
            # RayCaster::setInput(start = pt / res, end = cam / res)
            s = [pt[i] / res for i in range(3)]
            e = [cam[i] / res for i in range(3)]
            x = [int(math.floor(s[i])) for i in range(3)]
            ex = [int(math.floor(e[i])) for i in range(3)]
            dd = [ex[i] - x[i] for i in range(3)]
            step = [0 if v == 0 else (-1 if v < 0 else 1) for v in dd]

            def intbound(sv, ds):
                if ds < 0:
                    return intbound(-sv, -ds)
                sv = math.fmod(math.fmod(sv, 1) + 1, 1)
                return (1 - sv) / ds if ds != 0 else math.inf

            tmax = [intbound(s[i], float(dd[i])) for i in range(3)]
            tdel = [(step[i] / dd[i]) if dd[i] != 0 else math.inf for i in range(3)]
            while True:
                ray = (x[0], x[1], x[2])
                if x == ex:
                    break                                   # step() returns false: the camera's voxel is not used
                if tmax[0] < tmax[1]:
                    k = 0 if tmax[0] < tmax[2] else 2
                else:
                    k = 1 if tmax[1] < tmax[2] else 2
                x[k] += step[k]
                tmax[k] += tdel[k]
                tmp = [(ray[i] + 0.5) * res for i in range(3)]
                vox = self.set_cache(tmp, 0)
                if vox < 0:
                    continue
                if rn is not None and self.f_trav[vox] == rn:
                    break
                self.f_trav[vox] = rn if rn is not None else np.int8(self.raycast_num & 0x7F)
        for i in range(3):
            mn[i] = min(mn[i], cam[i])
            mx[i] = max(mx[i], cam[i])
        mx[2] = max(mx[2], p["ground_height"])
        self.lb_max = self.bound(self.pos_to_index(mx))
        self.lb_min = self.bound(self.pos_to_index(mn))
        lo = self.bound(self.pos_to_index([cam[i] - p["local_update_range"][i] for i in range(3)]))
        hi = self.bound(self.pos_to_index([cam[i] + p["local_update_range"][i] for i in range(3)]))
        while self.queue:
            i = self.queue.popleft()
            a = self.addr(i)
            upd = self.hit if self.cnt_hit[a] >= self.cnt[a] - self.cnt_hit[a] else self.miss
            self.cnt_hit[a] = self.cnt[a] = 0
            if upd >= 0 and self.occ[a] >= self.cmax:
                continue
            elif upd <= 0 and self.occ[a] <= self.cmin:
                self.occ[a] = self.cmin
                continue
            if not all(lo[k] <= i[k] <= hi[k] for k in range(3)):
                self.occ[a] = self.cmin
            self.occ[a] = min(max(self.occ[a] + upd, self.cmin), self.cmax)
        return True

    def clear_and_inflate(self):
        p, nv = self.p, self.nv
        occ3 = self.occ.reshape(nv)
        inf3 = self.inf.reshape(nv)
        mg = p["local_map_margin"]
        cut0 = self.bound([self.lb_min[k] - mg for k in range(3)])
        cut1 = self.bound([self.lb_max[k] + mg for k in range(3)])
        m0 = self.bound([cut0[k] - 5 for k in range(3)])
        m1 = self.bound([cut1[k] + 5 for k in range(3)])
        val = self.cmin - self.unknown
        X, Y, Z = slice(m0[0], m1[0] + 1), slice(m0[1], m1[1] + 1), slice(m0[2], m1[2] + 1)
        occ3[X, Y, m0[2]:cut0[2]] = val
        occ3[X, Y, cut1[2] + 1:m1[2] + 1] = val
        occ3[X, m0[1]:cut0[1], Z] = val
        occ3[X, cut1[1] + 1:m1[1] + 1, Z] = val
        occ3[m0[0]:cut0[0], Y, Z] = val
        occ3[cut1[0] + 1:m1[0] + 1, Y, Z] = val
        step = int(math.ceil(p["obstacles_inflation"] / self.res))
        B = tuple(slice(self.lb_min[k], self.lb_max[k] + 1) for k in range(3))
        inf3[B] = 0
        n = nv[0] * nv[1] * nv[2]
        xs, ys, zs = np.nonzero(occ3[B] > self.occ_log)
        for x, y, z in zip(xs + self.lb_min[0], ys + self.lb_min[1], zs + self.lb_min[2]):
            for dx in range(-step, step + 1):
                for dy in range(-step, step + 1):
                    for dz in range(-step, step + 1):
                        a = (x + dx) * nv[1] * nv[2] + (y + dy) * nv[2] + (z + dz)     # the flat address: rows wrap
                        if 0 <= a < n:
                            self.inf[a] = 1
        if self.ceil_h > -0.5:
            cz = int(math.floor((self.ceil_h - self.origin[2]) * self.res_inv)) - 1
            inf3[B[0], B[1], cz] = 1

    def update(self, img, cam, R):
        cam = [float(c) for c in cam]
        if not self.in_map(cam):
            return 0
        pts = self.project(img, cam, [[float(v) for v in row] for row in R])
        if self.raycast(pts, cam):
            self.clear_and_inflate()
            return 1
        return 0


def main():
    p = gridmap_fixture_params()
    g = GridMapRestated(p)
    frames = []
    for k, (img, cam, R) in enumerate(gridmap_fixture_frames()):
        upd = g.update(img, cam, np.asarray(R).reshape(3, 3))
        frames.append({"frame": k, "updated": upd, "in_sha256": hashlib.sha256(img.tobytes() + np.asarray(cam).tobytes() + np.asarray(R).tobytes()).hexdigest(),
                       "occ_sha256": hashlib.sha256(g.occ.tobytes()).hexdigest(),
                       "inflate_sha256": hashlib.sha256(g.inf.tobytes()).hexdigest(),
                       "bounds": [int(v) for v in g.lb_min + g.lb_max],
                       "cells_known": int((g.occ >= g.cmin).sum()), "cells_occupied": int((g.occ > g.occ_log).sum()),
                       "cells_inflated": int(g.inf.sum()), "occ_sum": float(g.occ.sum())})
        print(frames[-1])
    with open(os.path.join(HERE, "gridmap_independent.json"), "w") as f:
        json.dump({"voxels": g.nv, "frames": frames}, f, indent=1)


if __name__ == "__main__":
    main()
