#!/usr/bin/env python
"""An INDEPENDENT second restatement of FakeRiskHybridAstar::search — written straight from the reference's text in plain
Python floats (IEEE double, libm through `math`), WITHOUT reading oracle/ — run on the independent numpy map of
tests/golden/make_map_fixture.py.  The order in which it takes nodes from the open set (pool ids, first 200 per attempt),
the node and iteration counts and the return codes are committed as tests/golden/astar_independent.json; the CPU tests
hold the C++ oracle (oracle/astar_oracle.cpp) to them (tests/test_astar_independent.py).  Like the map fixture it does not
pin the oracle to the REFERENCE (Eigen / ROS absent: nothing here can), it makes two separately written readings agree.

Restated, statement by statement:
  FakeRiskHybridAstar::reset / search        path_searching/src/fake_risk_hybrid_a_star.cpp:84-100, 114-432
  estimateHeuristic, cubic, quartic          :434-469, 533-591
  computeShotTraj                            :470-531
  posToIndex, timeToIndex, stateTransit      :802-831
  NodeComparator, NodeHashTable              path_searching/include/path_searching/grid_node.h:46-51, path_node.h:70-97
  std::priority_queue = std::push_heap / std::pop_heap of libstdc++ (bits/stl_heap.h: __push_heap, __adjust_heap) — the
      sift order matters because nodes' f-scores are rewritten while they sit in the heap
  FakeBaselinePlanner::replan's two attempts plan_manager/src/baseline_fake.cpp:280-291
  parameters                                 plan_manager/config/sim_fake.yaml:12-26
Summation orders where the text leaves them to Eigen (3-term dot products, the 6x6 phi * state product) are written left to
right; the comparison that matters is the ORDER of the open set, which survives last-bit differences unless two f-scores
tie (the test would show it).  Run from the repo root:   python tests/golden/make_astar_fixture.py
"""
import importlib
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_map_fixture import FakeMap  # noqa: E402  (the independent map restatement)

NO_PATH, INIT_ERR, SEARCH_ERR, REACH_HORIZON, REACH_END, NEAR_END = range(6)   # dyn_a_star.h:15
NOT_EXPAND, IN_OPEN_SET, IN_CLOSE_SET = 0, 1, 2

# sim_fake.yaml:12-26
MAX_TAU, MAX_VEL, MAX_ACC, W_TIME, HORIZON, LAMBDA_HEU = 2.0, 2.0, 6.0, 5.0, 5.0, 5.0
RESOLUTION, TIME_RESOLUTION, ALLOCATE_NUM, CHECK_NUM, TOLERANCE = 0.15, 0.3, 10000, 1, 1
TIE_BREAKER = 1.0 + 1.0 / 10000          # setParam :76
INV_RES = 1.0 / RESOLUTION               # init :39-40
INV_TRES = 1.0 / TIME_RESOLUTION


def csqrt(x):
    """C's sqrt: NaN for negative arguments"""
    return math.sqrt(x) if x >= 0 else float("nan")


def cbrt(x):
    return float(np.cbrt(x))


def cubic(a, b, c, d):   # :533-560
    a2 = b / a
    a1 = c / a
    a0 = d / a
    Q = (3 * a1 - a2 * a2) / 9
    R = (9 * a1 * a2 - 27 * a0 - 2 * a2 * a2 * a2) / 54
    D = Q * Q * Q + R * R
    if D > 0:
        S = cbrt(R + csqrt(D))
        T = cbrt(R - csqrt(D))
        return [-a2 / 3 + (S + T)]
    elif D == 0:
        S = cbrt(R)
        return [-a2 / 3 + S + S, -a2 / 3 - S]
    else:
        theta = math.acos(R / csqrt(-Q * Q * Q))
        return [2 * csqrt(-Q) * math.cos(theta / 3) - a2 / 3,
                2 * csqrt(-Q) * math.cos((theta + 2 * math.pi) / 3) - a2 / 3,
                2 * csqrt(-Q) * math.cos((theta + 4 * math.pi) / 3) - a2 / 3]


def quartic(a, b, c, d, e):   # :562-591
    dts = []
    a3 = b / a
    a2 = c / a
    a1 = d / a
    a0 = e / a
    ys = cubic(1, -a2, a1 * a3 - 4 * a0, 4 * a2 * a0 - a1 * a1 - a3 * a3 * a0)
    y1 = ys[0]
    r = a3 * a3 / 4 - a2 + y1
    if r < 0:
        return dts
    R = csqrt(r)
    if R != 0:
        D = csqrt(0.75 * a3 * a3 - R * R - 2 * a2 + 0.25 * (4 * a3 * a2 - 8 * a1 - a3 * a3 * a3) / R)
        E = csqrt(0.75 * a3 * a3 - R * R - 2 * a2 - 0.25 * (4 * a3 * a2 - 8 * a1 - a3 * a3 * a3) / R)
    else:
        D = csqrt(0.75 * a3 * a3 - 2 * a2 + 2 * csqrt(y1 * y1 - 4 * a0))
        E = csqrt(0.75 * a3 * a3 - 2 * a2 - 2 * csqrt(y1 * y1 - 4 * a0))
    if not math.isnan(D):
        dts.append(-a3 / 4 + R / 2 + D / 2)
        dts.append(-a3 / 4 + R / 2 - D / 2)
    if not math.isnan(E):
        dts.append(-a3 / 4 - R / 2 + E / 2)
        dts.append(-a3 / 4 - R / 2 - E / 2)
    return dts


def dot3(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def estimate_heuristic(x1, x2):   # :434-469; returns (value, optimal_time)
    dp = [x2[i] - x1[i] for i in range(3)]
    v0 = x1[3:6]
    v1 = x2[3:6]
    c1 = -36 * dot3(dp, dp)
    c2 = 24 * dot3([v0[i] + v1[i] for i in range(3)], dp)
    c3 = -4 * (dot3(v0, v0) + dot3(v0, v1) + dot3(v1, v1))
    c4 = 0
    c5 = W_TIME
    ts = quartic(c5, c4, c3, c2, c1)
    v_max = MAX_VEL * 0.5
    t_bar = max(abs(x1[i] - x2[i]) for i in range(3)) / v_max
    ts.append(t_bar)
    cost = 100000000
    t_d = t_bar
    for t in ts:
        if t < t_bar:
            continue
        c = -c1 / (3 * t * t * t) - c2 / (2 * t * t) - c3 / t + W_TIME * t
        if c < cost:
            cost = c
            t_d = t
    return 1.0 * (1 + TIE_BREAKER) * cost, t_d


def state_transit(s0, um, tau):   # :816-831: phi = I with phi(i, i+3) = tau; state1 = phi * state0 + integral
    s1 = [0.0] * 6
    for i in range(3):
        s1[i] = (s0[i] + tau * s0[i + 3]) + 0.5 * math.pow(tau, 2) * um[i]
        s1[i + 3] = s0[i + 3] + tau * um[i]
    return s1


class Node:
    __slots__ = ("pid", "parent", "input", "state", "duration", "time", "time_idx", "index", "f", "g", "node_state")

    def __init__(self, pid):
        self.pid = pid
        self.parent = None
        self.node_state = NOT_EXPAND
        self.f = self.g = 0.0
        self.time = 0.0
        self.time_idx = 0
        self.index = (0, 0, 0)
        self.state = [0.0] * 6
        self.input = [0.0] * 3
        self.duration = 0.0


class Heap:
    """std::priority_queue<PathNodePtr, vector, NodeComparator>: comp(a, b) = a->f > b->f, evaluated on the CURRENT f"""

    def __init__(self):
        self.v = []

    @staticmethod
    def comp(a, b):
        return a.f > b.f

    def _push_heap(self, hole, top, value):
        v = self.v
        parent = (hole - 1) // 2
        while hole > top and self.comp(v[parent], value):
            v[hole] = v[parent]
            hole = parent
            parent = (hole - 1) // 2
        v[hole] = value

    def push(self, node):
        self.v.append(node)
        self._push_heap(len(self.v) - 1, 0, node)

    def top(self):
        return self.v[0]

    def pop(self):
        v = self.v
        if len(v) > 1:                 # std::pop_heap: the last element's value sifts down from the root
            value = v[-1]
            v[-1] = v[0]
            n = len(v) - 1             # __adjust_heap(first, 0, len = n, value)
            hole = 0
            child = 0
            while child < (n - 1) // 2:
                child = 2 * (child + 1)
                if self.comp(v[child], v[child - 1]):
                    child -= 1
                v[hole] = v[child]
                hole = child
            if (n & 1) == 0 and child == (n - 2) // 2:
                child = 2 * (child + 1)
                v[hole] = v[child - 1]
                hole = child - 1
            self._push_heap(hole, 0, value)
        v.pop()

    def empty(self):
        return not self.v


class Search:
    def __init__(self, fmap, pop_cap=200):
        self.map = fmap
        self.pool = [Node(i) for i in range(ALLOCATE_NUM)]
        self.use_node_num = 0
        self.pop_cap = pop_cap
        self.reset()

    def reset(self):   # :84-100
        self.table = {}
        self.node_path = []
        self.open = Heap()
        for i in range(self.use_node_num):
            self.pool[i].parent = None
            self.pool[i].node_state = NOT_EXPAND
        self.use_node_num = 0
        self.iter_num = 0
        self.is_shot_succ = False
        self.pops = []

    def pos_to_index(self, pt):
        return tuple(int(math.floor((pt[i] - self.map_center[i]) * INV_RES)) for i in range(3))

    def time_to_index(self, t):
        return int(math.floor((t - self.time_origin) * INV_TRES))

    def insert(self, idx, time_idx, node):   # unordered_map::insert keeps an existing entry
        self.table.setdefault((idx[0], idx[1], idx[2], time_idx), node)

    def retrieve_path(self, node):
        path = [node]
        while node.parent is not None:
            node = node.parent
            path.append(node)
        self.node_path = path[::-1]

    def compute_shot_traj(self, s1, s2, t_d):   # :470-531
        p0, v0, v1 = s1[0:3], s1[3:6], s2[3:6]
        dp = [s2[i] - p0[i] for i in range(3)]
        dv = [v1[i] - v0[i] for i in range(3)]
        a = [1.0 / 6.0 * (-12.0 / (t_d * t_d * t_d) * (dp[i] - v0[i] * t_d) + 6 / (t_d * t_d) * dv[i]) for i in range(3)]
        b = [0.5 * (6.0 / (t_d * t_d) * (dp[i] - v0[i] * t_d) - 2 / t_d * dv[i]) for i in range(3)]
        t_delta = t_d / 10
        time = t_delta
        while time <= t_d:
            tp = [math.pow(time, j) for j in range(4)]
            coord = [((p0[i] * tp[0] + v0[i] * tp[1]) + b[i] * tp[2]) + a[i] * tp[3] for i in range(3)]
            if self.map.clear_occupancy_dt(coord, time) != 0:
                return False
            time += t_delta
        self.is_shot_succ = True
        return True

    def search(self, start_pt, start_v, start_a, end_pt, end_v, init, dynamic, time_start):   # :114-432
        assert dynamic
        self.map_center = [float(x) for x in self.map.pose]    # getMapCenter().cast<double>()
        cur = self.pool[0]
        cur.parent = None
        cur.state = list(start_pt) + list(start_v)
        cur.index = self.pos_to_index(start_pt)
        cur.g = 0.0
        end_state = list(end_pt) + list(end_v)
        end_index = self.pos_to_index(end_pt)
        cur.f = LAMBDA_HEU * estimate_heuristic(cur.state, end_state)[0]
        cur.node_state = IN_OPEN_SET
        self.open.push(cur)
        self.use_node_num += 1
        self.time_origin = time_start
        cur.time = time_start
        cur.time_idx = self.time_to_index(time_start)
        self.insert(cur.index, cur.time_idx, cur)
        init_search = init
        while not self.open.empty():
            cur = self.open.top()
            d = [cur.state[i] - start_pt[i] for i in range(3)]
            reach_horizon = math.sqrt(dot3(d, d)) >= HORIZON
            near_end = all(abs(cur.index[i] - end_index[i]) <= TOLERANCE for i in range(3))
            exceed_time = cur.time >= MAX_TAU
            if reach_horizon or near_end or exceed_time:
                self.terminate = cur.pid
                self.retrieve_path(cur)
                if near_end:
                    t_goal = estimate_heuristic(cur.state, end_state)[1]
                    self.compute_shot_traj(cur.state, end_state, t_goal)
            if reach_horizon:
                return REACH_END if self.is_shot_succ else REACH_HORIZON
            if near_end:
                if self.is_shot_succ:
                    return REACH_END
                elif cur.parent is not None:
                    return NEAR_END
                else:
                    return NO_PATH
            if exceed_time:
                return REACH_HORIZON
            self.open.pop()
            cur.node_state = IN_CLOSE_SET
            self.iter_num += 1
            if len(self.pops) < self.pop_cap:
                self.pops.append(cur.pid)
            res = 1 / 2.0
            cur_state = list(cur.state)
            tmp_expand = []
            inputs = []
            if init_search:
                inputs.append(list(start_a))
                init_search = False
            else:
                ax = -MAX_ACC
                while ax <= MAX_ACC + 1e-3:
                    ay = -MAX_ACC
                    while ay <= MAX_ACC + 1e-3:
                        az = -0.5 * MAX_ACC
                        while az <= 0.5 * MAX_ACC + 1e-3:
                            inputs.append([ax, ay, az])
                            az += MAX_ACC * res
                        ay += MAX_ACC * res
                    ax += MAX_ACC * res
            durations = [TIME_RESOLUTION]
            for um in inputs:
                for tau in durations:
                    pro_state = state_transit(cur_state, um, tau)
                    pro_t = cur.time + tau
                    pro_id = self.pos_to_index(pro_state[0:3])
                    pro_t_id = self.time_to_index(pro_t)
                    pro_node = self.table.get((pro_id[0], pro_id[1], pro_id[2], pro_t_id))
                    if pro_node is not None and pro_node.node_state == IN_CLOSE_SET:
                        continue
                    if abs(pro_state[3]) > MAX_VEL or abs(pro_state[4]) > MAX_VEL or abs(pro_state[5]) > MAX_VEL:
                        continue
                    if pro_id == cur.index and pro_t_id - cur.time_idx == 0:
                        continue
                    is_occ = False
                    for k in range(1, CHECK_NUM + 1):
                        dt = tau * float(k) / float(CHECK_NUM)
                        xt = state_transit(cur_state, um, dt)
                        t = cur.time + dt
                        if self.map.clear_occupancy_dt(xt[0:3], t) != 0:
                            is_occ = True
                            break
                    if is_occ:
                        continue
                    tmp_g = (dot3(um, um) + W_TIME) * tau + cur.g
                    tmp_f = tmp_g + LAMBDA_HEU * estimate_heuristic(pro_state, end_state)[0]
                    prune = False
                    for en in tmp_expand:
                        if pro_id == en.index and pro_t_id == en.time_idx:
                            prune = True
                            if tmp_f < en.f:
                                en.f = tmp_f
                                en.g = tmp_g
                                en.state = pro_state
                                en.input = um
                                en.duration = tau
                                en.time = cur.time + tau
                            break
                    if not prune:
                        if pro_node is None:
                            pro_node = self.pool[self.use_node_num]
                            pro_node.index = pro_id
                            pro_node.state = pro_state
                            pro_node.f = tmp_f
                            pro_node.g = tmp_g
                            pro_node.input = um
                            pro_node.duration = tau
                            pro_node.parent = cur
                            pro_node.node_state = IN_OPEN_SET
                            pro_node.time = cur.time + tau
                            pro_node.time_idx = self.time_to_index(pro_node.time)
                            self.open.push(pro_node)
                            # insert(Eigen::Vector3i, int time_idx, ...) called with the DOUBLE pro_node->time: truncated
                            self.insert(pro_id, int(pro_node.time), pro_node)
                            tmp_expand.append(pro_node)
                            self.use_node_num += 1
                            if self.use_node_num == ALLOCATE_NUM:
                                return NO_PATH
                        elif pro_node.node_state == IN_OPEN_SET:
                            if tmp_g < pro_node.g:
                                pro_node.state = pro_state
                                pro_node.f = tmp_f
                                pro_node.g = tmp_g
                                pro_node.input = um
                                pro_node.duration = tau
                                pro_node.parent = cur
                                pro_node.time = cur.time + tau
                        else:
                            return SEARCH_ERR
        return NO_PATH


def replan_search(fmap, start_pva, goal, t_after_map, pop_cap=200):
    """baseline_fake.cpp:280-291: reset, search(init = true); on NO_PATH reset and search(init = false)"""
    s = Search(fmap, pop_cap)
    attempts = []
    for init in (True, False):
        s.reset()
        s.terminate = -1
        ret = s.search(start_pva[0:3], start_pva[3:6], start_pva[6:9], goal, [0.0, 0.0, 0.0], init, True, t_after_map)
        attempts.append({"init": init, "ret": ret, "pops": list(s.pops), "iter_num": s.iter_num,
                         "use_node_num": s.use_node_num, "terminate_node": s.terminate,
                         "path_nodes": [n.pid for n in s.node_path]})
        if ret != NO_PATH:
            break
    return attempts


POCKETS = [(1.6, 1.4), (2.2, 1.8), (1.2, 2.4), (2.8, 1.2)]   # (half_width, depth) of the wall pockets


def build_cases():
    """scene of the map fixture; per agent (i) start states / goals in the open field, (ii) the same scene plus a U-shaped
    wall pocket (tests/helpers.pocket_cloud) between start and goal, which makes the search long"""
    pop_scene = importlib.import_module("pred-occ-planner_amd.scene")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import pocket_cloud
    seed, A = 0x5A17, 3
    sc = pop_scene.make_scene(A, 4.95, seed=seed, moving=True)
    cyl = [{"type": 3, "x": float(r[0]), "y": float(r[1]), "w": float(r[2]), "vx": float(r[3]), "vy": float(r[4])}
           for r in sc["cylinders"]]
    rng = np.random.default_rng(seed + 1)
    cases = []
    for a in range(A):
        pose = sc["poses"][a].astype(np.float32)
        m = FakeMap()
        m.update_map(sc["cloud"], cyl, pose)
        for j in range(4):
            ang = rng.uniform(0, 2 * math.pi)
            dist = [9.0, 9.0, 0.8, 2.5][j]
            start = [float(pose[0]) + rng.uniform(-0.3, 0.3), float(pose[1]) + rng.uniform(-0.3, 0.3), 1.0 + rng.uniform(-0.2, 0.4)]
            vel = [rng.uniform(-1.2, 1.2), rng.uniform(-1.2, 1.2), rng.uniform(-0.2, 0.2)]
            acc = [rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-0.5, 0.5)]
            goal = [start[0] + dist * math.cos(ang), start[1] + dist * math.sin(ang), 1.0 + rng.uniform(-0.2, 0.4)]
            cases.append((a, m, pose, start + vel + acc, goal, 0.05 if j % 2 == 0 else 0.13, None))
        for j, (hw, depth) in enumerate(POCKETS):
            mp = FakeMap()
            mp.update_map(np.concatenate([sc["cloud"], pocket_cloud(pose, hw, depth)]), cyl, pose)
            start = [float(pose[0]) + rng.uniform(-0.2, 0.2), float(pose[1]) + rng.uniform(-0.4, 0.4), 1.0 + rng.uniform(-0.2, 0.4)]
            vel = [rng.uniform(0.3, 1.5), rng.uniform(-0.6, 0.6), rng.uniform(-0.1, 0.1)]
            acc = [rng.uniform(-1, 2), rng.uniform(-1, 1), rng.uniform(-0.3, 0.3)]
            goal = [start[0] + 9.0, start[1] + rng.uniform(-1.0, 1.0), 1.0 + rng.uniform(-0.2, 0.4)]
            cases.append((a, mp, pose, start + vel + acc, goal, 0.05 if j % 2 == 0 else 0.13, [hw, depth]))
    return seed, A, sc, cases


def main():
    seed, A, sc, cases = build_cases()
    out_cases = []
    for (a, m, pose, pva, goal, t0, pocket) in cases:
        att = replan_search(m, pva, goal, t0)
        print(f"agent {a} pocket {pocket}: goal {np.round(goal, 2).tolist()} t0 {t0}: " +
              " | ".join(f"init={x['init']} ret={x['ret']} iters={x['iter_num']} nodes={x['use_node_num']}" for x in att))
        out_cases.append({"agent": a, "pose": [float(x) for x in pose], "start_pva": [float(x) for x in pva],
                          "goal": [float(x) for x in goal], "t_after_map": t0, "pocket": pocket, "attempts": att})
    out = {"what": "FakeRiskHybridAstar::search restated independently in Python (tests/golden/make_astar_fixture.py) on the "
                   "independent map of make_map_fixture.py; scene = pred-occ-planner_amd.scene.make_scene(3, 4.95, seed, "
                   "moving=True); pops = pool ids in the order they leave the open set (first 200 per attempt)",
           "seed": seed, "agents": A, "pop_cap": 200, "cases": out_cases}
    with open(os.path.join(ROOT, "tests", "golden", "astar_independent.json"), "w") as f:
        json.dump(out, f)
    print("written tests/golden/astar_independent.json")


if __name__ == "__main__":
    main()
