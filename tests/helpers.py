"""Shared test helpers (not collected)."""
import importlib

import numpy as np


def hard_cases(pop, A, seed, field=7.0):
    """Scene whose agents start at random places INSIDE the obstacle field with random velocity
    (so that A* has to route around cylinders and other agents)."""
    sc = pop.scene.make_scene(A, 4.95, seed=seed, moving=True, circle_radius=8.0)
    rng = np.random.default_rng(seed)
    cyl = sc["cylinders"]
    starts = np.zeros((A, 3))
    for a in range(A):
        for _ in range(1000):
            p = rng.uniform(-field, field, 2)
            d = np.hypot(cyl[:, 0] - p[0], cyl[:, 1] - p[1]) - cyl[:, 2] * 0.5
            if d.min() > 0.9 and (a == 0 or np.hypot(*(starts[:a, :2] - p).T).min() > 1.6):
                break
        starts[a] = (p[0], p[1], rng.uniform(0.6, 1.8))
    goals = starts + np.concatenate([rng.uniform(-9, 9, (A, 2)), np.zeros((A, 1))], axis=1)
    goals[:, 2] = 1.0
    vel = rng.uniform(-1.0, 1.0, (A, 3)) * np.array([1, 1, 0.2])
    acc = rng.uniform(-2.0, 2.0, (A, 3)) * np.array([1, 1, 0.2])
    sc["starts"], sc["goals"] = starts, goals
    sc["poses"] = starts.astype(np.float32).copy()
    pva = np.concatenate([starts, vel, acc], axis=1)
    return sc, pva


def pocket_cloud(pose, half_width, depth, back=0.6, step=0.1):
    """Static wall points (no cylinder record: velocity zero) forming a U-shaped pocket that opens towards -x around `pose`:
    a front wall at x = pose.x + depth and two side walls at y = pose.y +- half_width reaching back to x = pose.x - back,
    floor to ceiling.  A search towards +x has to explore the pocket before it backs out: hundreds of expansions
    (tests/golden/make_astar_fixture.py, tests/test_astar_independent.py)."""
    zs = np.arange(0.05, 3.0, step)
    ys = np.arange(-half_width, half_width + 1e-6, step)
    xs = np.arange(-back, depth + 1e-6, step)
    front = np.stack(np.meshgrid([depth], ys, zs, indexing="ij"), -1).reshape(-1, 3)
    left = np.stack(np.meshgrid(xs, [-half_width], zs, indexing="ij"), -1).reshape(-1, 3)
    right = np.stack(np.meshgrid(xs, [half_width], zs, indexing="ij"), -1).reshape(-1, 3)
    pts = np.concatenate([front, left, right]) + np.asarray(pose, np.float64)[None, :] * np.array([1.0, 1.0, 0.0])
    return pts.astype(np.float32)


def oracle_grids(pop, orc, spec, sc, recs):
    cyl = pop.scene.cylinders_to_struct(sc["cylinders"])
    body = pop.scene.received_body_particles()
    out = []
    A = sc["n_agents"]
    for a in range(A):
        g = orc.update_gt(spec, sc["cloud"], cyl, len(sc["cylinders"]), sc["poses"][a])
        if recs is not None:
            orc.project_neighbours(spec, g, recs, A, a, body, sc["poses"][a], sc["stamps"][a])
        out.append(g)
    return out


def approaching_cylinder_scene(pop, gx, cy, vy):
    """One agent at the origin (z = 1), goal `gx` metres ahead, one cylinder beside the goal at (gx, cy) moving
    with velocity (0, vy): slice 0 and the later slices differ around the goal — the situation in which
    FakeRiskHybridAstar's shot check (with time) and RiskHybridAstar's (slice 0) give different return codes."""
    sc = pop.scene.make_scene(1, 4.95, seed=1, n_cyl=0)
    sc["starts"][0] = (0, 0, 1)
    sc["goals"][0] = (gx, 0, 1)
    sc["poses"][0] = (0, 0, 1)
    cyl = np.array([[gx, cy, 0.6, 0.0, vy]])
    sc["cylinders"] = cyl
    zs = np.arange(0, 40) * 0.1
    x, y, w = cyl[0, :3]
    r = w * 0.5
    gxx, gyy = np.meshgrid(np.arange(int(np.floor((x - r) / 0.1)), int(np.ceil((x + r) / 0.1)) + 1) * 0.1,
                           np.arange(int(np.floor((y - r) / 0.1)), int(np.ceil((y + r) / 0.1)) + 1) * 0.1,
                           indexing="ij")
    d = np.hypot(gxx - x, gyy - y)
    m = (d <= r) & (d > r - 0.15)
    sx, sy = gxx[m], gyy[m]
    sc["cloud"] = np.ascontiguousarray(
        np.stack([np.repeat(sx, zs.size), np.repeat(sy, zs.size), np.tile(zs, sx.size)], axis=1), np.float32)
    return sc


class OracleCompute:
    """The CPU oracle standing in for driver.HipCompute (the four kernels calls of a tick), so that the rank-local
    bookkeeping of driver.SwarmTick.step() — tick inputs, map update from the EXCHANGED records, replan, latest-wins
    merge, the all-gather — runs on CPU over gloo (tests/test_driver_gloo.py).  Test infrastructure only."""
    device = "cpu"
    ctx = None

    def __init__(self, pop, orc, spec, scene, lo, hi, timeline=None):
        self.pop, self.orc, self.spec, self.scene, self.lo, self.hi = pop, orc, spec, scene, lo, hi
        self.timeline = timeline  # scene.WorldTimeline: the sensor frame of every tick (None: the frozen scene)
        self.abi = pop._abi
        self.ap, self.pp, self.qs = (pop.config.make_astar_params(), pop.config.make_planner_params(True),
                                     pop.config.make_qp_settings())
        self.cyl = pop.scene.cylinders_to_struct(scene["cylinders"])
        self.body = pop.scene.received_body_particles()
        self.grids, self.swarm, self.pub = [None] * (hi - lo), None, None
        self.overlay_sums = []   # (tick stamp, agent, sum of the SOGM, hash of the occupied cells) per agent-update

    def set_swarm(self, all_records, A_tot, now):
        self.swarm = (all_records, A_tot)

    def set_publish(self, own, next_table):
        self.pub = (own, next_table)

    def _records(self, t):
        n = t.shape[0]
        return (self.abi.SogmTrajRecord * n).from_buffer_copy(t.numpy().tobytes())

    def tick_inputs(self, own, stamp, hover, now, t_start, pva, poses):
        """k_tick_inputs (csrc/sogm_map.hip) restated: start state = the executed trajectory at stamp + 0.02, or
        the hover state; hover <- position with zero velocity / acceleration."""
        import importlib
        drv = importlib.import_module("pred-occ-planner_amd.driver")
        recs = self._records(own)
        ts = stamp + drv.REPLAN_START_TIME
        for i in range(own.shape[0]):
            r = recs[i]
            if r.n_pieces > 0:
                d = np.array(r.duration[:r.n_pieces])
                c = np.array(r.cpts[:15 * r.n_pieces]).reshape(-1, 3)
                tt = min(max(ts - r.time_start, 0.0), d.sum())
                o = np.concatenate([self.orc.bezier_eval(d, c, tt, k) for k in range(3)])
            else:
                o = hover[i].numpy().copy()
            pva[i] = torch_from(o)
            hover[i, :3] = torch_from(o[:3])
            hover[i, 3:] = 0.0
            poses[i] = torch_from(o[:3].astype(np.float32))
        now.fill_(stamp)
        t_start.fill_(ts)

    def update_map(self, poses, now, all_records, A_tot, tick=0):
        recs = self._records(all_records)
        cloud, cyl = self.scene["cloud"], self.cyl
        if self.timeline is not None:
            f = self.timeline.frame(tick)
            cloud, cyl = f["cloud"], self.pop.scene.cylinders_to_struct(f["cylinders"])
        for i in range(self.hi - self.lo):
            pose = poses[i].numpy()
            g = self.orc.update_gt(self.spec, cloud, cyl, len(self.scene["cylinders"]), pose)
            self.orc.project_neighbours(self.spec, g, recs, A_tot, self.lo + i, self.body, pose, float(now[i]))
            self.grids[i] = g
            self.overlay_sums.append((round(float(now[i]), 6), self.lo + i, float(g.sum()),
                                      int(np.flatnonzero(g.ravel()).sum() % 1000003)))

    def replan(self, pva, goals, t_start, new, ok):
        import torch
        allr = self._records(self.swarm[0]) if self.swarm else None
        for i in range(self.hi - self.lo):
            a = self.lo + i
            pose = pva[i, :3].numpy().astype(np.float32)
            stamp = float(t_start[i]) - 0.02
            okk, rec, _ = self.orc.replan(self.spec, self.ap, self.pp, self.qs, self.grids[i], pose, stamp,
                                          pva[i].numpy(), goals[i].numpy(), float(t_start[i]), a)
            if okk and allr is not None:  # isSafeAfterOpt, the last step of replan() (baseline_fake.cpp:453-460)
                okk = self.orc.safe_after_opt(np.asarray(rec.cpts[:15 * rec.n_pieces]), rec.n_pieces, allr,
                                              self.swarm[1], a, stamp)
            ok[i] = int(bool(okk))
            new[i] = torch.from_numpy(np.frombuffer(bytes(rec), dtype=np.uint8).copy())
        if self.pub is not None and self.pub[0] is not None:  # sogm_planner_set_publish restated
            own, table = self.pub
            own.copy_(new.where(ok.bool().unsqueeze(1), own))
            if table is not None:
                table.copy_(own)

    def merge_latest(self, new, ok, own, all_records):
        """k_merge_latest restated: a successful replan replaces the agent's record, a failed one keeps it; the
        swarm table (single process only) is refreshed in the same pass."""
        sel = ok.bool().unsqueeze(1)
        own.copy_(new.where(sel, own))
        if all_records is not None:
            all_records.copy_(own)

    def close(self):
        pass


def torch_from(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a))



def filter_fixture_clouds():
    """camera frame: x right, y down, z forward"""
    out = []
    rng = np.random.default_rng(20260929)
    a = np.stack([rng.uniform(-4, 4, 30000), rng.uniform(-1.2, 1.2, 30000), rng.uniform(0.3, 4.5, 30000)], 1).astype(np.float32)
    out.append(("uniform box", a))
    b = a[:12000].copy()
    b[::531] = np.nan                      # invalid depth pixels
    b[5::977, 2] = np.inf
    out.append(("with non-finite points", b))
    # a wall 3 m ahead sampled like a depth image (dense: many points per leaf) + points beyond every range
    u, v = np.meshgrid(np.linspace(-2.5, 2.5, 260), np.linspace(-1.0, 1.0, 110))
    wall = np.stack([u.ravel(), v.ravel(), np.full(u.size, 3.0) + 0.02 * np.sin(7 * u.ravel())], 1).astype(np.float32)
    far = np.float32([[0.0, 0.0, 9.0], [0.2, 0.1, 9.3], [6.0, 0.0, 1.0], [0.0, -1.6, 2.0]])
    out.append(("wall + out of range", np.concatenate([wall, far])))
    out.append(("empty", np.zeros((0, 3), np.float32)))
    # > 5000 leaves inside the range: the cap
    g = np.stack(np.meshgrid(np.arange(-30, 30), np.arange(-6, 6), np.arange(3, 28), indexing="ij"), -1).reshape(-1, 3)
    out.append(("the cap", (g * 0.15 + 0.07).astype(np.float32)))
    # coordinates exactly on leaf boundaries and exactly on the range
    e = np.float32([[0.15, 0.0, 0.3], [0.3, 0.15, 0.45], [-0.15, -0.15, 0.15], [0.0, 0.0, 4.95], [0.0, 0.0, 4.9499998],
                    [4.95, 0.0, 1.0], [-4.9499998, 0.0, 1.0], [0.0, 1.5, 1.0], [0.0, 1.4999999, 1.0], [0.0, -1.5, 1.0]])
    out.append(("boundaries", e))
    return out


def gridmap_fixture_params():
    """the small test map of tests/test_gridmap.py as a plain dict (inputs of tests/golden/make_gridmap_fixture.py)"""
    return {"resolution": 0.1, "map_size": [12.0, 12.0, 3.0], "local_update_range": [4.0, 4.0, 2.0], "obstacles_inflation": 0.099,
            "fx": 387.0, "fy": 387.0, "cx": 320.0, "cy": 240.0, "depth_filter_maxdist": 5.0, "depth_filter_mindist": 0.2,
            "k_depth_scaling_factor": 1000.0, "p_hit": 0.70, "p_miss": 0.35, "p_min": 0.12, "p_max": 0.97, "p_occ": 0.80,
            "max_ray_length": 4.5, "virtual_ceil_height": 2.5, "ground_height": -0.01, "use_depth_filter": 1,
            "depth_filter_margin": 2, "skip_pixel": 2, "local_map_margin": 5, "rows": 480, "cols": 640}


def gridmap_fixture_frames(n=9):
    """depth images + camera poses of a camera creeping through the small map (frame 0 only arms the depth filter; five hits
    take a cell from unknown to occupied)"""
    pop = importlib.import_module("pred-occ-planner_amd")
    out = []
    for k in range(n):
        img = pop.scene.make_depth_image(k % 2)
        cam, R = pop.scene.camera_pose(-3.0 + 0.02 * k, 0.05 * np.sin(0.4 * k), 1.0 + 0.005 * k, 0.04 * np.sin(0.5 * k))
        out.append((img, cam, R))
    return out



def dsp_fixture_inputs(n_updates=4, seed=7):
    """the injected tables (Gaussian position / velocity noise, rand()) and a synthetic sensor sequence with labels for
    tests/golden/make_dsp_fixture.py and tests/test_dsp_independent.py"""
    pop = importlib.import_module("pred-occ-planner_amd")
    dsp = importlib.import_module("pred-occ-planner_amd.dsp")
    tables = dsp.make_tables(11, n_gauss=1 << 18, n_rand=1 << 12)
    seq = pop.scene.make_dsp_sequence(seed, n_updates, half=(66 * 0.15 / 2, 66 * 0.15 / 2, 20 * 0.15 / 2))
    return tables, seq
