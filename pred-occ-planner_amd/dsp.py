"""Host mirror of dsp_map::DSPMap as owned by RiskVoxel (plan_env/src/risk_voxel.cpp:42-50,237-254),
batched over the agents of a SogmMap.  Forwards to sogm_dsp_* / sogm_update_dsp (include/sogm_abi.h);
no CPU path."""
import ctypes as C

import numpy as np
import torch

from ._abi import SOGM_DSP_MAX_T, SogmDspParams, check, lib
from .sogm import _stream


def make_dsp_params(T=6):
    """map_parameters.h:5-54 + RiskVoxel::init (risk_voxel.cpp:20-21,42-50)."""
    p = SogmDspParams()
    p.max_particle_num_voxel = 7
    p.half_fov_h, p.half_fov_v, p.angle_resolution = 43, 29, 1
    p.newborn_num = 20
    p.obs_max_per_pyramid = 100
    for t in range(SOGM_DSP_MAX_T):
        p.prediction_times[t] = round(0.3 * (t + 1), 1)  # prediction_future_time {0.3f .. 1.8f}
    p.sigma_observation = 0.05
    p.p_detection = 0.95
    p.kappa = 0.01
    p.newborn_weight = 0.0001
    p.obstacle_thickness = 0.3
    return p


def make_tables(seed, n_gauss=1 << 20, n_rand=1 << 16, p_std=0.05, v_std=0.05):
    """Stand-ins for generateGaussianRandomsVectorZeroCenter (dsp_dynamic.h:1229-1239; N(0, 0.05) after
    setPredictionVariance(0.05, 0.05), risk_voxel.cpp:43) and for rand(): reproducible tables."""
    rng = np.random.default_rng(seed)
    pg = (rng.standard_normal(n_gauss) * p_std).astype(np.float32)
    vg = (rng.standard_normal(n_gauss) * v_std).astype(np.float32)
    rnd = rng.integers(0, 2 ** 31 - 1, size=n_rand, dtype=np.int64).astype(np.int32)
    return pg, vg, rnd


class DspMap:
    def __init__(self, sogm_map, params, tables, max_points=5000):
        self.map = sogm_map
        self.params = params
        pg, vg, rnd = tables
        self._h = C.c_void_p()
        fp = C.POINTER(C.c_float)
        check(lib().sogm_dsp_create(sogm_map.ctx, C.byref(params), pg.ctypes.data_as(fp),
                                    vg.ctypes.data_as(fp), len(pg), rnd.ctypes.data_as(C.c_void_p),
                                    len(rnd), max_points, C.byref(self._h)), "sogm_dsp_create")
        self.max_points = max_points
        self.S = 2 * params.max_particle_num_voxel
        self.NP = (params.half_fov_h * 2 // params.angle_resolution) * (params.half_fov_v * 2 // params.angle_resolution)

    def close(self):
        if self._h:
            torch.cuda.synchronize()
            lib().sogm_dsp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self, points, labels, cloud_range, sensor_pos, sensor_quat, stamps, out_ok=None):
        """DSPMap::update for every agent; all arguments are device tensors (see sogm_abi.h).  labels=None runs
        velocityEstimationThread (clustering + association) on the GPU."""
        ok = out_ok if out_ok is not None else torch.zeros(self.map.n_agents, dtype=torch.int32, device=points.device)
        check(lib().sogm_update_dsp(self._h, points.data_ptr(), labels.data_ptr() if labels is not None else None,
                                    cloud_range.data_ptr(),
                                    sensor_pos.data_ptr(), sensor_quat.data_ptr(), stamps.data_ptr(),
                                    ok.data_ptr(), _stream()), "sogm_update_dsp")
        return ok

    def download_born(self, agent):
        """input_cloud_with_velocity of the last update: (rows [n, 7], counters {clusters, dynamic, matched, err})."""
        cap = self.max_points
        born = np.zeros((cap, 7), np.float32)
        n = C.c_int32(0)
        cnt = (C.c_int32 * 4)()
        check(lib().sogm_dsp_download_born(self._h, agent, born.ctypes.data_as(C.c_void_p), cap, C.byref(n), cnt),
              "sogm_dsp_download_born")
        return born[:n.value].copy(), list(cnt)

    def publish(self):
        """RiskVoxel::publishMap's map half: future status -> SOGM grid (+ inflate-kernel zeroing)."""
        n = torch.zeros(self.map.n_agents, dtype=torch.int32, device="cuda")
        check(lib().sogm_dsp_publish(self._h, n.data_ptr(), _stream()), "sogm_dsp_publish")
        return n

    def download_state(self, agent):
        V, T = self.map.V, self.map.spec.T
        store = np.zeros((V, self.S, 9), np.float32)
        objnum = np.zeros((V, 4 + T), np.float32)
        counters = np.zeros(16, np.int32)
        check(lib().sogm_dsp_download_state(self._h, agent, store.ctypes.data_as(C.c_void_p),
                                            objnum.ctypes.data_as(C.c_void_p),
                                            counters.ctypes.data_as(C.c_void_p)), "sogm_dsp_download_state")
        return store, objnum, counters

    def download_observations(self, agent):
        OM = self.params.obs_max_per_pyramid
        nobs = np.zeros(self.NP, np.int32)
        pc = np.zeros((self.NP, OM, 5), np.float32)
        ml = np.zeros(self.NP, np.float32)
        check(lib().sogm_dsp_download_observations(self._h, agent, nobs.ctypes.data_as(C.c_void_p),
                                                   pc.ctypes.data_as(C.c_void_p), ml.ctypes.data_as(C.c_void_p)),
              "sogm_dsp_download_observations")
        return nobs, pc, ml
