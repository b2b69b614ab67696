"""Host-side mirror of the reference map classes for a BATCH of agents, over the C ABI.

Method names follow the reference (`updateMap`, `getClearOcccupancy` [sic], `getObstaclePoints`,
`getMapTime`, `getMapCenter`, `setCoordinator`-style neighbour overlay) —
plan_env/include/plan_env/fake_particle_risk_voxel.h:49-107, risk_base.h:33-110.  PyTorch is used
only as plumbing: device buffers and the current HIP stream.  All compute runs in libsogm_hip.so.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _abi
from ._abi import check, lib
from .scene import received_body_particles, struct_to_numpy


def _dev(x, dtype=None, device="cuda"):
    """numpy/ctypes -> device tensor (uint8 for structs)."""
    if isinstance(x, torch.Tensor):
        return x.to(device)
    if not isinstance(x, np.ndarray):
        x = struct_to_numpy(x)
    if dtype is not None:
        x = np.ascontiguousarray(x, dtype=dtype)
    return torch.from_numpy(np.ascontiguousarray(x)).to(device)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class SogmMap:
    """Batched FakeParticleRiskVoxel / RiskBase.  One instance = n_agents maps on one GPU."""

    def __init__(self, spec, n_agents, device=0, drone_size=(0.4, 0.4, 0.45)):
        self.spec = spec
        self.n_agents = n_agents
        self.device = device
        self._ctx = C.c_void_p()
        torch.cuda.set_device(device)
        check(lib().sogm_create(C.byref(spec), n_agents, device, C.byref(self._ctx)), "sogm_create")
        self.body = received_body_particles(drone_size)   # as another drone's ParticleATC receives them (Point32)
        check(lib().sogm_set_body_particles(
            self._ctx, self.body.ctypes.data_as(C.POINTER(C.c_double)), len(self.body)),
            "sogm_set_body_particles")
        self.V = spec.L * spec.W * spec.H
        self._keep = []  # tensors referenced by in-flight calls
        # SOGM_TUNING="key=value,key=value": tools and A/B scripts set the library's tuning knobs this way — applied
        # HERE, by the binding, through sogm_set_tuning (the library itself reads no tuning from the environment)
        for kv in filter(None, os.environ.get("SOGM_TUNING", "").split(",")):
            k, _, v = kv.partition("=")
            try:
                val = float(v)
            except ValueError:
                raise ValueError(f"SOGM_TUNING: item {kv!r} is not key=number") from None
            self.set_tuning(k.strip(), val)

    # ---- tuning knobs (sogm_abi.h: sogm_set_tuning) ----
    def set_tuning(self, key, value):
        check(lib().sogm_set_tuning(self._ctx, key.encode(), float(value)), f"sogm_set_tuning({key})")

    def get_tuning(self, key):
        out = C.c_double(0.0)
        check(lib().sogm_get_tuning(self._ctx, key.encode(), C.byref(out)), f"sogm_get_tuning({key})")
        return out.value

    # ---- lifetime ----
    def close(self):
        if self._ctx:
            torch.cuda.synchronize()
            lib().sogm_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ctx(self):
        return self._ctx

    def grid_bytes(self):
        return int(lib().sogm_grid_bytes(self._ctx))

    def set_overlap_clear(self, on=True, double_buffer=None, grids=None):
        """Tick pipelining.  Modes 3 / 2 (a pool of three / two grids, the default when HBM has room): replan()
        clears the spare grid swapped out by this tick's update with a narrow streaming kernel under the whole
        replan; with three grids the next update takes the grid cleared one tick earlier, so the clear is never on
        the tick's critical path.  Mode 1: the grid is cleared in place in two launches — a narrow head beside the
        FIRI kernels, the full-width rest under the QP stage.  `grids` (1 | 2 | 3) forces a mode; `double_buffer`
        False forces mode 1.  Returns the mode in effect."""
        if not on:
            check(lib().sogm_set_overlap_clear(self._ctx, 0), "sogm_set_overlap_clear")
            return 0
        if grids is None:
            if double_buffer is False:
                grids = 1
            else:
                free, _ = torch.cuda.mem_get_info()
                spare = (free - (16 << 30)) // max(self.grid_bytes(), 1)
                grids = 3 if spare >= 2 else (2 if spare >= 1 else 1)
                if double_buffer is True:
                    grids = max(grids, 2)
        rc = 0
        for mode in ([3, 2, 1] if grids >= 3 else [2, 1] if grids == 2 else [1]):
            rc = lib().sogm_set_overlap_clear(self._ctx, mode)
            if rc == 0:
                return mode
            if rc != _abi.SOGM_ERR_CAPACITY:
                check(rc, "sogm_set_overlap_clear")
        check(rc, "sogm_set_overlap_clear")  # even the last fallback (mode 1 needs no memory) failed: not silent
        return 0

    def isTrajSafe(self, records, t_now, check_duration):
        """BaselinePlanner::isTrajSafe for every agent's executed trajectory (device uint8 [A, 2064])."""
        out = torch.empty((self.n_agents,), dtype=torch.int32, device=t_now.device)
        check(lib().sogm_traj_safe(self._ctx, records.data_ptr(), t_now.data_ptr(), float(check_duration),
                                   out.data_ptr(), _stream()), "sogm_traj_safe")
        return out

    def filterPointCloud(self, raw_xyz, raw_range, filter_res=0.15, cap=5000):
        """MapBase::filterPointCloud (map.cpp:107-132) for every agent: (points [A, cap, 3], counts [A])."""
        out = torch.empty((self.n_agents, cap, 3), dtype=torch.float32, device=raw_xyz.device)
        cnt = torch.empty((self.n_agents,), dtype=torch.int32, device=raw_xyz.device)
        check(lib().sogm_filter_point_cloud(self._ctx, raw_xyz.data_ptr(), raw_range.data_ptr(), filter_res, cap,
                                            out.data_ptr(), cnt.data_ptr(), _stream()), "sogm_filter_point_cloud")
        return out, cnt

    # ---- sparse reset (sogm_abi.h: sogm_set_sparse_reset) ----
    def set_sparse_reset(self, on=True, log_capacity=0):
        check(lib().sogm_set_sparse_reset(self._ctx, 1 if on else 0, int(log_capacity)), "sogm_set_sparse_reset")

    def sparse_reset_state(self):
        """{enabled, log_capacity, tracked (current grid covered by its log), max_entries (largest per-agent count),
        total_entries (all agents)} of the current grid; resets / entries_per_reset: the sparse resets launched since the
        previous call."""
        out = (C.c_int32 * 8)()
        check(lib().sogm_sparse_reset_state(self._ctx, out), "sogm_sparse_reset_state")
        return {"enabled": bool(out[0]), "log_capacity": out[1], "tracked": bool(out[2]), "max_entries": out[3],
                "total_entries": out[4], "resets": out[5], "entries_per_reset": out[6],
                "zeroed_bytes_per_reset": int(out[7]) * 1024}

    def set_resample(self, replan_risk_rate, num_resample, normal_table):
        """ParticleATC's resample branch (sogm_set_resample): `normal_table` = device float32 tensor of standard
        normals, >= 3 * body particles * num_resample entries, kept alive here."""
        self._resample_table = normal_table
        check(lib().sogm_set_resample(self._ctx, float(replan_risk_rate), int(num_resample),
                                      normal_table.data_ptr() if normal_table is not None else None,
                                      int(normal_table.numel()) if normal_table is not None else 0), "sogm_set_resample")

    def map_traffic(self, reset=False):
        """Device-side counts of what the resets and stamps moved since the counters' last reset (sogm_map_traffic)."""
        out = (C.c_int64 * 6)()
        check(lib().sogm_map_traffic(self._ctx, out, 1 if reset else 0), "sogm_map_traffic")
        return {"resets": out[0], "reset_entries": out[1], "reset_bytes_zeroed": out[2], "stamps": out[3],
                "stamp_marks": out[4], "stamp_entries": out[5]}

    def device_clock(self):
        """(device wall clock in seconds, host perf_counter at the call's return): the device's 100 MHz clock read by a
        kernel on the current stream, synchronised (sogm_device_clock)"""
        import time
        out = C.c_int64(0)
        check(lib().sogm_device_clock(self._ctx, C.byref(out), _stream()), "sogm_device_clock")
        return out.value * 1e-8, time.perf_counter()

    def tick_clock(self):
        """device clock (s) at the start of the last update's first kernel and in the last replan's closing kernel"""
        out = (C.c_int64 * 2)()
        check(lib().sogm_tick_clock(self._ctx, out), "sogm_tick_clock")
        return out[0] * 1e-8, out[1] * 1e-8

    def grid_history(self):
        """How the current grid came to be: {slot, sparse_resets, dense_clears, prestamped} (sogm_grid_history)."""
        out = (C.c_int32 * 4)()
        check(lib().sogm_grid_history(self._ctx, out), "sogm_grid_history")
        return {"slot": out[0], "sparse_resets": out[1], "dense_clears": out[2], "prestamped": bool(out[3])}

    # ---- profiling (HIP events around each kernel, on the caller's stream) ----
    def set_profiling(self, on=True, slots=None):
        """Time every profiled kernel (on=True), none, or the given slots only (iterable of SOGM_PROF_* indices)."""
        if slots is not None:
            mask = 0
            for k in slots:
                mask |= 1 << int(k)
            check(lib().sogm_set_profiling_slots(self._ctx, mask), "sogm_set_profiling_slots")
            return
        check(lib().sogm_set_profiling(self._ctx, 1 if on else 0), "sogm_set_profiling")

    def profile_read(self):
        out = (C.c_double * _abi.PROF_N)()
        check(lib().sogm_profile_read(self._ctx, out), "sogm_profile_read")
        return list(out)

    def map_state(self, agent):
        """(getMapTime().toSec(), getMapCenter()) of one agent (risk_base.h:70,76)."""
        t = C.c_double(0.0)
        c = (C.c_float * 3)()
        check(lib().sogm_map_state(self.ctx, agent, C.byref(t), c, _stream()), "sogm_map_state")
        return t.value, np.array(list(c), np.float32)

    def profile_read_all(self, slot, cap=1024):
        """Durations (ms, oldest first) of every launch of `slot` since set_profiling(True) (last 1024 kept)."""
        out = (C.c_double * cap)()
        n = C.c_int(0)
        check(lib().sogm_profile_read_all(self._ctx, slot, out, cap, C.byref(n)), "sogm_profile_read_all")
        return [out[i] for i in range(n.value)]

    # ---- update ----
    def updateMap(self, cloud, cloud_range, cylinders, n_cyl, poses, stamps):
        """FakeParticleRiskVoxel::updateMap for every agent (device tensors)."""
        self._poses, self._stamps = poses, stamps
        check(lib().sogm_update_gt(self._ctx, cloud.data_ptr(), cloud_range.data_ptr(),
                                   cylinders.data_ptr() if n_cyl else None, n_cyl,
                                   poses.data_ptr(), stamps.data_ptr(), _stream()),
              "sogm_update_gt")

    def updateMapSwarm(self, cloud, cloud_range, cylinders, n_cyl, poses, stamps, records, n_records, ego_ids):
        """FakeParticleRiskVoxel::updateMap incl. its closing neighbour overlay, one call (sogm_update_gt_swarm);
        the maps equal those of updateMap + addOtherAgents."""
        self._poses, self._stamps = poses, stamps
        check(lib().sogm_update_gt_swarm(self._ctx, cloud.data_ptr(), cloud_range.data_ptr(),
                                         cylinders.data_ptr() if n_cyl else None, n_cyl, poses.data_ptr(),
                                         stamps.data_ptr(), records.data_ptr() if n_records else None, n_records,
                                         ego_ids.data_ptr(), _stream()), "sogm_update_gt_swarm")

    def updateWorld(self, world, poses, stamps, records=None, n_records=0, ego_ids=None):
        """FakeParticleRiskVoxel::updateMap (+ its closing overlay when `records` is given) from one World frame, the
        PassThrough crop done on the device around `poses` (sogm_update_world)."""
        self._poses, self._stamps, self._world = poses, stamps, world
        check(lib().sogm_update_world(self._ctx, C.byref(world.c), poses.data_ptr(), stamps.data_ptr(),
                                      records.data_ptr() if n_records else None, n_records,
                                      ego_ids.data_ptr() if ego_ids is not None else None, _stream()), "sogm_update_world")

    def prestamp_pending(self):
        """True if the last replan pre-stamped the next grid (sogm_planner_set_prestamp)."""
        return bool(lib().sogm_prestamp_pending(self._ctx))

    def prestamp_join(self):
        """the current stream waits for the end of the last replan's pre-stamp, if nothing has joined it yet"""
        check(lib().sogm_prestamp_join(self._ctx, _stream()), "sogm_prestamp_join")

    def updatePrestamped(self, records, n_records, ego_ids):
        """The update of a pre-stamped tick: grid swap + neighbour overlay (sogm_update_prestamped)."""
        check(lib().sogm_update_prestamped(self._ctx, records.data_ptr() if records is not None else None, n_records,
                                           ego_ids.data_ptr() if ego_ids is not None else None, _stream()),
              "sogm_update_prestamped")

    def addOtherAgents(self, records, n_records, ego_ids):
        """RiskBase::addOtherAgents / fake_particle_risk_voxel.cpp:178-218."""
        check(lib().sogm_project_neighbours(self._ctx, records.data_ptr(), n_records,
                                            ego_ids.data_ptr(), _stream()),
              "sogm_project_neighbours")

    def futureRiskCallback(self, grid_vt, poses, stamps):
        check(lib().sogm_set_future_risk(self._ctx, grid_vt.data_ptr(), poses.data_ptr(),
                                         stamps.data_ptr(), _stream()), "sogm_set_future_risk")

    # ---- readback ----
    def download(self, agent):
        """risk_maps_[V][T] of one agent (reference layout), numpy float32."""
        out = np.empty((self.V, self.spec.T), dtype=np.float32)
        check(lib().sogm_download_reference_layout(self._ctx, agent, out.ctypes.data),
              "sogm_download_reference_layout")
        return out

    # ---- queries ----
    def getClearOcccupancy(self, agent_idx, pos, t, t_is_index=False):
        """Batched getClearOcccupancy; agent_idx int32[n], pos float64[n,3], t float64[n]."""
        n = int(agent_idx.numel())
        out = torch.empty((n,), dtype=torch.int8, device=agent_idx.device)
        check(lib().sogm_query_clear(self._ctx, agent_idx.data_ptr(), pos.data_ptr(),
                                     t.data_ptr(), 1 if t_is_index else 0, n, out.data_ptr(),
                                     _stream()), "sogm_query_clear")
        return out

    def getObstaclePoints(self, agent_idx, box_lo, box_hi, t0, t1, cap=4096):
        n = int(agent_idx.numel())
        pts = torch.empty((n, cap, 3), dtype=torch.float64, device=agent_idx.device)
        cnt = torch.empty((n,), dtype=torch.int32, device=agent_idx.device)
        check(lib().sogm_obstacle_points(self._ctx, agent_idx.data_ptr(), box_lo.data_ptr(),
                                         box_hi.data_ptr(), t0.data_ptr(), t1.data_ptr(), n,
                                         pts.data_ptr(), cnt.data_ptr(), cap, _stream()),
              "sogm_obstacle_points")
        return pts, cnt


class World:
    """One sensor frame on the device (SogmWorld): cloud, block bounds (computed on the device by
    sogm_cloud_block_bounds), GT cylinders.  `cloud` numpy / tensor [n, 3] float32, `cylinders` numpy (n, 5) rows
    {x, y, w, vx, vy} or a ctypes SogmCylinder array."""

    def __init__(self, cloud, cylinders, n_cyl=None, block_points=256, device="cuda"):
        from .scene import cylinders_to_struct
        if not isinstance(cylinders, C.Array):
            n_cyl = len(cylinders)
            cylinders = cylinders_to_struct(cylinders)
        self.cloud = _dev(cloud if len(cloud) else np.zeros((1, 3), np.float32), np.float32, device)
        self.n_points = int(len(cloud))
        self.block_points = int(block_points)
        self.n_blocks = (self.n_points + self.block_points - 1) // self.block_points
        self.bounds = torch.empty((max(self.n_blocks, 1), 4), dtype=torch.float32, device=device)
        self.cylinders = _dev(cylinders, None, device)
        self.n_cyl = int(n_cyl)
        check(lib().sogm_cloud_block_bounds(self.cloud.data_ptr(), self.n_points, self.block_points,
                                            self.bounds.data_ptr(), _stream()), "sogm_cloud_block_bounds")
        self.c = _abi.SogmWorld(self.cloud.data_ptr(), self.bounds.data_ptr(),
                                self.cylinders.data_ptr() if self.n_cyl else None, self.n_points, self.n_blocks,
                                self.block_points, self.n_cyl)


def upload_scene(scene, device="cuda", cloud=None, cloud_range=None):
    """numpy scene -> dict of device tensors in the ABI's layouts.  By default every agent sees the
    whole cloud; pass per-agent crops (scene.crop_clouds) to bound each agent's scan."""
    from .scene import cylinders_to_struct
    A = scene["n_agents"]
    pts = scene["cloud"] if cloud is None else cloud
    n_pts = pts.shape[0]
    cyl = cylinders_to_struct(scene["cylinders"])
    rng = np.tile(np.asarray([[0, n_pts]], dtype=np.int32), (A, 1)) if cloud_range is None else cloud_range
    return {
        "cloud": _dev(pts if n_pts else np.zeros((1, 3), np.float32), np.float32, device),
        "cloud_range": _dev(rng, np.int32, device),
        "cylinders": _dev(cyl, None, device),
        "n_cyl": int(len(scene["cylinders"])),
        "poses": _dev(scene["poses"], np.float32, device),
        "stamps": _dev(scene["stamps"], np.float64, device),
        "ego_ids": _dev(scene["ego_ids"], np.int32, device),
    }
