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


def oracle_grids(pop, orc, spec, sc, recs):
    cyl = pop.scene.cylinders_to_struct(sc["cylinders"])
    body = pop.scene.body_particles()
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
