"""Deterministic synthetic scenes (SURVEY.md §8 d).

The reference's obstacle generator (`map_generator dynamic_forest_seq`,
plan_manager/launch/simulator/simulator_fake.launch:19-53) lives in an absent submodule, so the
build owns the scene.  Parameters follow that launch file: obstacle radius U[0.5, 1.0] m, height 4 m,
speed U[0, 0.1] m/s, cloud lattice 0.10 m; agents on a circle with antipodal goals as in
plan_manager/launch/sim_fkpcp_4_case_4.launch:21-89.  Everything is a pure function of the seed
(splitmix64 -> uniform), numpy only — no torch, no GPU.
"""
import ctypes as C
import math

import numpy as np

from ._abi import SOGM_MAX_PIECES, SogmCylinder, SogmTrajRecord

_M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _M64

    def next_u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * ((self.next_u64() >> 11) * (1.0 / (1 << 53)))


def body_particles(size=(0.4, 0.4, 0.45)):
    """ParticleATC::initEgoParticles (traj_coordinator/src/particles.cpp:62-75): fp64 loops, STEP 0.15."""
    step = 0.15
    out = []
    x = -size[0] / 2
    while x <= size[0] / 2:
        y = -size[1] / 2
        while y <= size[1] / 2:
            z = -size[2] / 2
            while z <= size[2] / 2:
                out.append((x, y, z))
                z += step
            y += step
        x += step
    return np.asarray(out, dtype=np.float64)


def received_body_particles(size=(0.4, 0.4, 0.45)):
    """What a ParticleATC holds for ANOTHER drone (the only particles the neighbour overlay and isSafeAfterOpt place):
    particlesCallback (traj_coordinator/src/particles.cpp:89-108) copies them from a geometry_msgs/PolygonStamped, whose
    points are geometry_msgs/Point32 — the sender's fp64 ego particles rounded to float32 and cast back to double."""
    return body_particles(size).astype(np.float32).astype(np.float64)


def cylinders_to_struct(cyl):
    """(n, 5) float64 rows {x, y, w, vx, vy} -> ctypes array of SogmCylinder (type 3, height 4)."""
    arr = (SogmCylinder * max(len(cyl), 1))()
    for i, (x, y, w, vx, vy) in enumerate(cyl):
        c = arr[i]
        c.type = 3
        c.x, c.y, c.z, c.w, c.h = x, y, 2.0, w, 4.0
        c.vx, c.vy = vx, vy
        c.qw, c.qx, c.qy, c.qz = 1.0, 0.0, 0.0, 0.0
    return arr


def add_rings(scene, rings, first=True):
    """Ring obstacles (`Cylinder.type == 2`, fake_particle_risk_voxel.cpp:137-149; the map_generator of the
    reference's dynamic_forest_seq emits them next to the cylinders).  rings: rows {x, y, z, w (diameter), vx, vy,
    qw, qx, qy, qz}.  Appends the ring's circle (0.10 m spacing, 3 x 3 tube section) to the cloud and returns
    (SogmCylinder array, count) with the rings before (first=True) or after the scene's cylinders — the
    reference takes the FIRST record that contains a voxel."""
    rings = np.asarray(rings, np.float64).reshape(-1, 10)
    cyl = scene["cylinders"]
    n = len(cyl) + len(rings)
    arr = (SogmCylinder * max(n, 1))()
    base = cylinders_to_struct(cyl)
    off = len(rings) if first else 0
    for i in range(len(cyl)):
        C.memmove(C.addressof(arr[off + i]), C.addressof(base[i]), C.sizeof(SogmCylinder))
    roff = 0 if first else len(cyl)
    pts = [scene["cloud"]]
    for i, (x, y, z, w, vx, vy, qw, qx, qy, qz) in enumerate(rings):
        c = arr[roff + i]
        c.type = 2
        c.x, c.y, c.z, c.w, c.h = x, y, z, w, 0.1
        c.vx, c.vy = vx, vy
        c.qw, c.qx, c.qy, c.qz = qw, qx, qy, qz
        q = np.array([qw, qx, qy, qz])
        R = _quat_to_matrix(q / np.linalg.norm(q))
        r = 0.5 * w
        n_seg = max(8, int(math.ceil(2.0 * math.pi * r / 0.1)))
        th = np.arange(n_seg) * (2.0 * math.pi / n_seg)
        for dr in (-0.1, 0.0, 0.1):
            for dn in (-0.1, 0.0, 0.1):
                local = np.stack([(r + dr) * np.cos(th), (r + dr) * np.sin(th), np.full_like(th, dn)], axis=1)
                pts.append((local @ R.T + np.array([x, y, z])).astype(np.float32))
    scene["cloud"] = np.ascontiguousarray(np.concatenate(pts, axis=0), np.float32)
    return arr, n


def _quat_to_matrix(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def make_scene(n_agents, half_range, seed, moving=True, n_cyl=None, circle_radius=None,
               tick_time=100.0):
    """Returns a dict of numpy arrays.

    half_range: map half extent in metres (grid L/2 * 0.15) — used to size the field so every
    agent's window is populated.
    """
    rng = SplitMix64(seed)
    # agents evenly on a circle, >= 1.5 m apart, antipodal goals, z = 1
    if circle_radius is None:
        circle_radius = max(8.0, 1.5 * n_agents / (2.0 * math.pi))
    starts = np.zeros((n_agents, 3))
    goals = np.zeros((n_agents, 3))
    for a in range(n_agents):
        th = 2.0 * math.pi * a / n_agents + math.pi
        starts[a] = (circle_radius * math.cos(th), circle_radius * math.sin(th), 1.0)
        goals[a] = (-starts[a, 0], -starts[a, 1], 1.0)
    if n_agents == 1:
        starts[0] = (-8.0, 0.0, 1.0)
        goals[0] = (8.0, 0.0, 1.0)

    # Obstacles where the agents fly during a run: the annulus [R - inner, R + outer] around the
    # start circle (the whole disk for small swarms), at the launch file's density of 20 per 16x16 m.
    inner, outer = 30.0, half_range + 2.0
    r_lo, r_hi = max(circle_radius - inner, 0.0), circle_radius + outer
    if n_agents == 1:
        r_lo, r_hi = 0.0, 8.0 + outer
    if n_cyl is None:
        n_cyl = max(20, int(round(20.0 * math.pi * (r_hi ** 2 - r_lo ** 2) / 256.0)))
    cyl = []
    guard = 0
    # spatial hash of starts/goals for the clearance test (keeps generation O(n_cyl))
    pts_sg = np.concatenate([starts[:, :2], goals[:, :2]], axis=0)
    cell = 4.0
    table = {}
    for q in pts_sg:
        table.setdefault((int(math.floor(q[0] / cell)), int(math.floor(q[1] / cell))), []).append(q)
    while len(cyl) < n_cyl and guard < 50 * n_cyl + 1000:
        guard += 1
        rho = math.sqrt(rng.uniform(r_lo ** 2, r_hi ** 2))
        ang = rng.uniform(0.0, 2.0 * math.pi)
        x, y = rho * math.cos(ang), rho * math.sin(ang)
        w = rng.uniform(0.5, 1.0)
        speed = rng.uniform(0.0, 0.1) * (10.0 if moving else 0.0)  # up to 1 m/s when moving
        head = rng.uniform(0.0, 2.0 * math.pi)
        # keep 1 m (+ radius) clear around every start/goal
        cx, cy = int(math.floor(x / cell)), int(math.floor(y / cell))
        near = False
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for q in table.get((cx + dx, cy + dy), ()):
                    if math.hypot(q[0] - x, q[1] - y) < 1.0 + w + 0.6:
                        near = True
        if near:
            continue
        cyl.append((x, y, w, speed * math.cos(head), speed * math.sin(head)))
    cyl = np.asarray(cyl, dtype=np.float64).reshape(-1, 5)

    # cloud: cylinder shells on a 0.10 m lattice, z in [0, 4)
    pts, owner = [], []
    zs = np.arange(0, 40) * 0.1
    for ci, (x, y, w, _, _) in enumerate(cyl):
        r = w * 0.5
        k0x, k1x = int(math.floor((x - r) / 0.1)), int(math.ceil((x + r) / 0.1))
        k0y, k1y = int(math.floor((y - r) / 0.1)), int(math.ceil((y + r) / 0.1))
        gx, gy = np.meshgrid(np.arange(k0x, k1x + 1) * 0.1, np.arange(k0y, k1y + 1) * 0.1,
                             indexing="ij")
        d = np.hypot(gx - x, gy - y)
        m = (d <= r) & (d > r - 0.15)
        sx, sy = gx[m], gy[m]
        if sx.size == 0:
            continue
        col = np.stack([np.repeat(sx, zs.size), np.repeat(sy, zs.size),
                        np.tile(zs, sx.size)], axis=1)
        pts.append(col)
        owner.append(np.full((col.shape[0],), ci, np.int32))
    cloud = (np.concatenate(pts, axis=0) if pts else np.zeros((0, 3))).astype(np.float32)
    cloud_cyl = np.concatenate(owner) if owner else np.zeros((0,), np.int32)

    return {
        "n_agents": n_agents,
        "cloud": np.ascontiguousarray(cloud),
        "cloud_cyl": cloud_cyl,      # the cylinder every cloud point was sampled from (WorldTimeline moves it along)
        "cylinders": cyl,
        "starts": starts,
        "goals": goals,
        "poses": starts.astype(np.float32).copy(),
        "stamps": np.full((n_agents,), tick_time, dtype=np.float64),
        "ego_ids": np.arange(n_agents, dtype=np.int32),
        "circle_radius": circle_radius,
    }


class WorldTimeline:
    """The scene as a function of time: the sensor frames the fake-perception map receives, one per tick.

    The reference's simulator (map_generator dynamic_forest_seq, absent submodule) republishes the global cloud and the
    obstacle states while the obstacles move with their constant planar velocities (simulator_fake.launch:41); the map
    is rebuilt from the cloud and the states that arrived FOR THAT update (map.cpp:170-171,
    fake_particle_risk_voxel.cpp:244-264).  frame(k) is that input at tick k: every cylinder advanced by v * (k * dt) and
    its cloud points — contiguous runs of the cloud — moved with it.  Pure numpy, a function of (scene, k): the oracle
    flights of the tests and the device uploads of the driver use the same arrays.
    moving=False: every frame is the scene at tick 0 (the frozen world of the earlier rounds, kept for the tests that
    compare single builds)."""

    def __init__(self, scene, dt=0.1, moving=True):
        self.scene, self.dt, self.moving = scene, float(dt), bool(moving)
        self.cyl0 = np.asarray(scene["cylinders"], np.float64).reshape(-1, 5)
        self.cloud0 = np.ascontiguousarray(scene["cloud"], np.float32)
        self.owner = scene.get("cloud_cyl")
        if self.moving and (self.owner is None or len(self.owner) != len(self.cloud0)):
            raise ValueError("WorldTimeline(moving=True) needs scene['cloud_cyl'] (make_scene provides it)")

    def cylinders(self, k):
        """(n, 5) rows {x, y, w, vx, vy} at tick k (fp64: x0 + vx * (k * dt))"""
        c = self.cyl0.copy()
        if self.moving and len(c):
            t = float(k) * self.dt
            c[:, 0] = self.cyl0[:, 0] + self.cyl0[:, 3] * t
            c[:, 1] = self.cyl0[:, 1] + self.cyl0[:, 4] * t
        return c

    def cloud(self, k):
        """[n, 3] float32 at tick k: the point's tick-0 position plus its cylinder's displacement rounded to fp32"""
        if not self.moving or not len(self.cloud0) or k == 0:
            return self.cloud0
        t = float(k) * self.dt
        off = (self.cyl0[:, 3:5] * t).astype(np.float32)          # per cylinder
        out = self.cloud0.copy()
        out[:, :2] += off[self.owner]
        return out

    def frame(self, k):
        return {"cloud": self.cloud(k), "cylinders": self.cylinders(k)}


def straight_records(scene, speed=1.0, t_start=None, n_pieces=6, piece_dur=0.3):
    """Constant-velocity Bezier trajectories start->goal for every agent (neighbour overlay input
    before any replan has produced real ones).  Returns a ctypes array of SogmTrajRecord."""
    A = scene["n_agents"]
    recs = (SogmTrajRecord * A)()
    for a in range(A):
        r = recs[a]
        r.drone_id = int(scene["ego_ids"][a])
        r.n_pieces = n_pieces
        r.time_start = float(scene["stamps"][a] - 0.05 if t_start is None else t_start)
        d = scene["goals"][a] - scene["starts"][a]
        n = np.linalg.norm(d)
        u = d / n if n > 0 else d
        for p in range(n_pieces):
            r.duration[p] = piece_dur
            for j in range(5):
                s = (p + j / 4.0) * piece_dur * speed
                for k in range(3):
                    r.cpts[(p * 5 + j) * 3 + k] = scene["starts"][a][k] + u[k] * s
    return recs


def records_to_numpy(recs):
    """ctypes SogmTrajRecord array -> uint8 numpy buffer (for upload)."""
    return np.frombuffer(bytes(recs), dtype=np.uint8).copy()


def struct_to_numpy(arr):
    return np.frombuffer(bytes(arr), dtype=np.uint8).copy()


def crop_clouds(scene, lo, hi, half):
    """Per-agent crop of the global cloud: agent a sees the points within `half` metres (x and y)
    of its start — the role of the simulator's local sensing radius.  Returns (points, ranges) with
    the crops concatenated; ranges[a] = (begin, end) in the ABI's cloud_range layout."""
    cloud = scene["cloud"]
    order = np.argsort(cloud[:, 0], kind="stable")
    xs = cloud[order, 0]
    chunks, ranges, pos = [], [], 0
    for a in range(lo, hi):
        cx, cy = scene["starts"][a, 0], scene["starts"][a, 1]
        i0, i1 = np.searchsorted(xs, cx - half), np.searchsorted(xs, cx + half, side="right")
        sel = order[i0:i1]
        sel = np.sort(sel[np.abs(cloud[sel, 1] - cy) <= half])  # keep the global point order
        chunks.append(cloud[sel])
        ranges.append((pos, pos + len(sel)))
        pos += len(sel)
    pts = np.concatenate(chunks, axis=0) if chunks else np.zeros((0, 3), np.float32)
    return np.ascontiguousarray(pts, np.float32), np.asarray(ranges, np.int32).reshape(-1, 2)


def quat_yaw(yaw):
    return np.asarray([math.cos(yaw / 2.0), 0.0, 0.0, math.sin(yaw / 2.0)], np.float32)


def make_dsp_sequence(seed, n_updates, half=(4.95, 4.95, 1.5), n_pillars=6, dt=0.1, speed=0.6,
                      max_points=5000, wall=True):
    """Synthetic sensor-frame clouds for the particle-filter SOGM (SURVEY.md 8d "depth clouds"):
    a sensor flying along +x with a slow yaw oscillation past static pillars, one moving pillar
    (points labelled with its velocity, as velocityEstimationThread would) and one unmatched cluster
    (label vx = -10000), plus ground points and an optional side wall.  Points lie on a 0.15 m lattice
    (MapBase::filterPointCloud output), within 5 m and +-60 deg of the optical axis; the FOV test itself
    is the map's job.  Returns a list of dicts {points [n,3] f32 sensor frame, labels [n,4] f32,
    pos [3] f32, quat [4] f32 wxyz, stamp f64}."""
    rng = SplitMix64(seed)
    pillars = []
    for i in range(n_pillars):
        pillars.append([rng.uniform(1.5, 9.0), rng.uniform(-3.0, 3.0), rng.uniform(0.2, 0.45), 0.0, 0.0, 0])
    pillars.append([4.0, -2.5, 0.3, -0.2, 0.5, 1])    # moving, matched cluster
    pillars.append([6.0, 2.0, 0.3, 0.0, 0.0, 2])      # unmatched cluster (vx label -10000)
    out = []
    zs = np.arange(-8, 9) * 0.15
    for k in range(n_updates):
        t = 100.0 + k * dt
        pos = np.asarray([speed * k * dt, 0.1 * math.sin(0.7 * k * dt), 0.05 * math.sin(0.5 * k * dt)], np.float32)
        yaw = 0.25 * math.sin(0.9 * k * dt)
        q = quat_yaw(yaw)
        cy, sy = math.cos(yaw), math.sin(yaw)
        pts, lab = [], []
        for (x0, y0, r, vx, vy, kind) in pillars:
            cx, cyy = x0 + vx * k * dt, y0 + vy * k * dt
            for ang in np.arange(0.0, 2 * math.pi, 0.15 / r):
                wx, wy = cx + r * math.cos(ang), cyy + r * math.sin(ang)
                # visible half only (normal faces the sensor)
                if (wx - cx) * (pos[0] - wx) + (wy - cyy) * (pos[1] - wy) <= 0:
                    continue
                for z in zs:
                    pts.append((wx, wy, z))
                    lab.append((vx, vy, 0.0, 0.5) if kind == 1 else
                               (-10000.0, -10000.0, -10000.0, 0.7) if kind == 2 else (0.0, 0.0, 0.0, 0.0))
        for gx in np.arange(0.5, 5.0, 0.3):          # ground strip ahead of the sensor
            for gy in np.arange(-2.0, 2.01, 0.3):
                pts.append((pos[0] + gx, gy, -1.2))
                lab.append((0.0, 0.0, 0.0, 0.0))
        if wall:                                      # side wall seen at a grazing angle
            for wx in np.arange(pos[0] + 1.0, pos[0] + 5.0, 0.15):
                for z in zs[4:13]:
                    pts.append((wx, 1.6, z))
                    lab.append((0.0, 0.0, 0.0, 0.0))
        pts = np.asarray(pts, np.float64).reshape(-1, 3)
        lab = np.asarray(lab, np.float32).reshape(-1, 4)
        rel = pts - pos.astype(np.float64)
        # world-aligned -> sensor frame: rotate by -yaw
        sx = cy * rel[:, 0] + sy * rel[:, 1]
        syy = -sy * rel[:, 0] + cy * rel[:, 1]
        rng_ = np.sqrt(sx * sx + syy * syy + rel[:, 2] ** 2)
        keep = (rng_ < 5.0) & (sx > 0.2) & (np.abs(np.arctan2(syy, sx)) < math.radians(60.0))
        sp = np.stack([sx, syy, rel[:, 2]], axis=1)[keep].astype(np.float32)
        lab = lab[keep]
        if len(sp) > max_points:
            sp, lab = sp[:max_points], lab[:max_points]
        out.append({"points": np.ascontiguousarray(sp), "labels": np.ascontiguousarray(lab), "pos": pos,
                    "quat": q, "stamp": t})
    return out


def make_depth_cloud(seed, shape=(480, 640), fx=387.0, max_depth=4.4):
    """Back-projected synthetic depth image in the CAMERA frame (x right, y down, z forward), pin-hole
    fx = fy = 387, cx = 320, cy = 240 (GridMap::projectDepthImage's formula,
    plan_env/src/grid_map.cpp:231-235): a tilted wall, a pillar and a floor with 2 mm depth noise.
    Returns [H*W, 3] float32 — the input of MapBase::filterPointCloud."""
    rng = np.random.default_rng(seed)
    v, u = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
    depth = 3.5 + 0.002 * (u - shape[1] / 2) + rng.normal(0, 0.002, u.shape)
    pil = np.abs(u - 200 - 40 * (seed % 3)) < 35
    depth = np.where(pil, 1.6 + rng.normal(0, 0.002, u.shape), depth)
    floor = v > 0.83 * shape[0]
    depth = np.where(floor, 1.2 * fx / np.maximum(v - shape[0] / 2, 1), depth)
    depth = np.clip(depth, 0.3, max_depth)
    x = (u - shape[1] / 2) * depth / fx
    y = (v - shape[0] / 2) * depth / fx
    return np.stack([x, y, depth], axis=-1).reshape(-1, 3).astype(np.float32)


def make_depth_image(seed, shape=(480, 640), fx=387.0, scale=1000.0, holes=True):
    """uint16 depth image (depth * k_depth_scaling_factor) of the same synthetic view as make_depth_cloud:
    tilted wall, pillar, floor; optional zero-depth holes and a too-near patch to exercise the depth
    filter branches of GridMap::projectDepthImage (grid_map.cpp:262-271)."""
    rng = np.random.default_rng(seed)
    v, u = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
    depth = 3.5 + 0.002 * (u - shape[1] / 2) + rng.normal(0, 0.002, u.shape)
    pil = np.abs(u - 200 - 40 * (seed % 3)) < 35
    depth = np.where(pil, 1.6 + rng.normal(0, 0.002, u.shape), depth)
    floor = v > 0.83 * shape[0]
    depth = np.where(floor, 1.2 * fx / np.maximum(v - shape[0] / 2, 1), depth)
    far = (u > 0.8 * shape[1]) & (v < 0.3 * shape[0])
    depth = np.where(far, 7.5, depth)                     # beyond depth_filter_maxdist
    img = np.clip(depth * scale, 0, 65535).astype(np.uint16)
    if holes:
        img[rng.random(img.shape) < 0.01] = 0             # invalid pixels
        img[20:40, 30:60] = 100                            # 0.1 m: below depth_filter_mindist
    return img


def camera_pose(x, y, z, yaw):
    """Camera-to-world rotation (row-major 3x3) of a forward-looking camera (x right, y down, z forward)
    on a body with heading `yaw`, and its position."""
    c, s = math.cos(yaw), math.sin(yaw)
    body = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    cam2body = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    return np.asarray([x, y, z], np.float64), (body @ cam2body).astype(np.float64)
