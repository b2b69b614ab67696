"""Host mirror of GridMap's depth front end (plan_env/src/grid_map.cpp:210-583), batched over agents.
Forwards to sogm_gridmap_* (include/sogm_abi.h); no CPU path."""
import ctypes as C

import numpy as np
import torch

from ._abi import SogmGridMapParams, check, lib
from .sogm import _stream


def make_gridmap_params(rows=480, cols=640):
    """The reference ships no grid_map YAML (every default is -1, grid_map.cpp:19-41): EGO-planner-style
    values for a 40 x 40 x 3 m map at 0.1 m, D435-like intrinsics (SURVEY.md section 8 d)."""
    p = SogmGridMapParams()
    p.resolution = 0.1
    p.map_size[:] = [40.0, 40.0, 3.0]
    p.local_update_range[:] = [5.5, 5.5, 4.5]
    p.obstacles_inflation = 0.099
    p.fx = p.fy = 387.0
    p.cx, p.cy = 320.0, 240.0
    p.depth_filter_maxdist, p.depth_filter_mindist = 5.0, 0.2
    p.k_depth_scaling_factor = 1000.0
    p.p_hit, p.p_miss, p.p_min, p.p_max, p.p_occ = 0.70, 0.35, 0.12, 0.97, 0.80
    p.max_ray_length = 4.5
    p.virtual_ceil_height, p.ground_height = 2.5, -0.01
    p.use_depth_filter, p.depth_filter_margin, p.skip_pixel, p.local_map_margin = 1, 2, 2, 30
    p.rows, p.cols = rows, cols
    return p


class GridMap:
    def __init__(self, params, n_agents, device=0):
        self.params, self.n_agents = params, n_agents
        self._h = C.c_void_p()
        torch.cuda.set_device(device)
        check(lib().sogm_gridmap_create(C.byref(params), n_agents, device, C.byref(self._h)), "sogm_gridmap_create")
        self.nv = [int(np.ceil(params.map_size[i] / params.resolution)) for i in range(3)]
        self.N = self.nv[0] * self.nv[1] * self.nv[2]

    def close(self):
        if self._h:
            lib().sogm_gridmap_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self, depth, cam_pos, cam_rot):
        """depthPoseCallback + updateOccupancyCallback for every agent: depth uint16 (int16 storage) [A, rows,
        cols], cam_pos fp64 [A, 3], cam_rot fp64 [A, 9] row-major.  Returns updated flags [A] int32."""
        upd = torch.zeros((self.n_agents,), dtype=torch.int32, device=cam_pos.device)
        check(lib().sogm_gridmap_update(self._h, depth.data_ptr(), cam_pos.data_ptr(), cam_rot.data_ptr(),
                                        upd.data_ptr(), _stream()), "sogm_gridmap_update")
        return upd

    def getInflateOccupancy(self, agent_idx, pos):
        out = torch.empty((pos.shape[0],), dtype=torch.int8, device=pos.device)
        check(lib().sogm_gridmap_query_inflate(self._h, agent_idx.data_ptr(), pos.data_ptr(), pos.shape[0],
                                               out.data_ptr(), _stream()), "sogm_gridmap_query_inflate")
        return out

    def download(self, agent):
        occ = np.zeros(self.N, np.float64)
        inf = np.zeros(self.N, np.int8)
        bounds = np.zeros(6, np.int32)
        counters = np.zeros(4, np.int32)
        check(lib().sogm_gridmap_download(self._h, agent, occ.ctypes.data_as(C.c_void_p), inf.ctypes.data_as(C.c_void_p),
                                          bounds.ctypes.data_as(C.c_void_p), counters.ctypes.data_as(C.c_void_p)),
              "sogm_gridmap_download")
        return occ, inf, bounds, counters

    def force_frame(self, n):
        check(lib().sogm_gridmap_force_frame(self._h, n), "sogm_gridmap_force_frame")
