"""ctypes binding of include/sogm_abi.h (the C-ABI drop-in boundary).

The HIP library is REQUIRED: importing this module without ``libsogm_hip.so`` raises — there is no
CPU fallback in the product path.  Plain-data records mirror the C structs field for field.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SOGM_LIB_PATH: another build of the same ABI, for same-box A/B runs of two library versions — tools/micro/ab.sh)
LIB_PATH = os.environ.get("SOGM_LIB_PATH") or os.path.join(_HERE, "libsogm_hip.so")

SOGM_ABI_VERSION = 6  # include/sogm_abi.h; load_library() refuses a library of another version
SOGM_MAX_PIECES = 16
SOGM_MAP_FAKE = 0
SOGM_MAP_RISKBASE = 1
SOGM_MAP_RISKVOXEL = 2
SOGM_STORE_F32, SOGM_STORE_F16 = 0, 1
SOGM_LAYOUT_TILED = 16  # OR-ed into SogmSpec.storage: 2 x 2 x 2 cell tiles
SOGM_DSP_MAX_T = 16

SOGM_OK = 0
SOGM_ERR_INVALID_ARG = -1
SOGM_ERR_NO_DEVICE = -2
SOGM_ERR_HIP = -3
SOGM_ERR_CAPACITY = -4
SOGM_ERR_STATE = -5
SOGM_ERR_COMM = -6
SOGM_COMM_ID_BYTES = 128
PROF_CLEAR, PROF_STAMP, PROF_SPLAT, PROF_ASTAR, PROF_CORRIDOR, PROF_QP, PROF_CLEAR_HEAD, PROF_EXCHANGE, PROF_N = range(9)

COUNTER_NAMES = ("replan_ok", "fail_search", "fail_corridor", "fail_qp", "fail_unsafe", "corridor_capacity",
                 "pieces_capacity", "deconflict_capacity")

# ASTAR_RET (path_searching/include/path_searching/dyn_a_star.h:15)
ASTAR_NO_PATH, ASTAR_INIT_ERR, ASTAR_SEARCH_ERR, ASTAR_REACH_HORIZON, ASTAR_REACH_END, ASTAR_NEAR_END = range(6)


class SogmSpec(C.Structure):
    _fields_ = [("L", C.c_int32), ("W", C.c_int32), ("H", C.c_int32), ("T", C.c_int32),
                ("resolution", C.c_float), ("time_resolution", C.c_float),
                ("risk_threshold", C.c_float), ("clearance", C.c_float),
                ("ground_height", C.c_float), ("ceiling_height", C.c_float),
                ("risk_threshold_region", C.c_float), ("risk_thres_reg_decay", C.c_float),
                ("risk_thres_vox_decay", C.c_float), ("map_kind", C.c_int32),
                ("storage", C.c_int32)]


class SogmDspParams(C.Structure):
    _fields_ = [("max_particle_num_voxel", C.c_int32), ("half_fov_h", C.c_int32),
                ("half_fov_v", C.c_int32), ("angle_resolution", C.c_int32),
                ("newborn_num", C.c_int32), ("obs_max_per_pyramid", C.c_int32),
                ("prediction_times", C.c_float * SOGM_DSP_MAX_T),
                ("sigma_observation", C.c_float), ("p_detection", C.c_float), ("kappa", C.c_float),
                ("newborn_weight", C.c_float), ("obstacle_thickness", C.c_float)]


class SogmGridMapParams(C.Structure):
    _fields_ = [("resolution", C.c_double), ("map_size", C.c_double * 3), ("local_update_range", C.c_double * 3),
                ("obstacles_inflation", C.c_double), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("depth_filter_maxdist", C.c_double), ("depth_filter_mindist", C.c_double),
                ("k_depth_scaling_factor", C.c_double), ("p_hit", C.c_double), ("p_miss", C.c_double),
                ("p_min", C.c_double), ("p_max", C.c_double), ("p_occ", C.c_double), ("max_ray_length", C.c_double),
                ("virtual_ceil_height", C.c_double), ("ground_height", C.c_double),
                ("use_depth_filter", C.c_int32), ("depth_filter_margin", C.c_int32), ("skip_pixel", C.c_int32),
                ("local_map_margin", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32)]


class SogmCylinder(C.Structure):
    _fields_ = [("type", C.c_int32), ("_pad", C.c_int32)] + [
        (k, C.c_double) for k in ("x", "y", "z", "w", "h", "vx", "vy", "qw", "qx", "qy", "qz")]


class SogmTrajRecord(C.Structure):
    _fields_ = [("drone_id", C.c_int32), ("n_pieces", C.c_int32), ("time_start", C.c_double),
                ("duration", C.c_double * SOGM_MAX_PIECES),
                ("cpts", C.c_double * (SOGM_MAX_PIECES * 15))]


class SogmAstarParams(C.Structure):
    _fields_ = [("max_tau", C.c_double), ("max_vel", C.c_double), ("max_acc", C.c_double),
                ("w_time", C.c_double), ("horizon", C.c_double), ("lambda_heu", C.c_double),
                ("resolution", C.c_double), ("time_resolution", C.c_double),
                ("allocate_num", C.c_int32), ("check_num", C.c_int32), ("tolerance", C.c_int32),
                ("shot_ignores_time", C.c_int32)]


class SogmPlannerParams(C.Structure):
    _fields_ = [("corridor_tau", C.c_double), ("init_range", C.c_double),
                ("shrink_size", C.c_double), ("opt_max_vel", C.c_double),
                ("opt_max_acc", C.c_double), ("fake_planner", C.c_int32),
                ("firi_iterations", C.c_int32), ("pc_capacity", C.c_int32),
                ("max_faces", C.c_int32)]


class SogmQpSettings(C.Structure):
    _fields_ = [("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double),
                ("eps_abs", C.c_double), ("eps_rel", C.c_double), ("max_iter", C.c_int32),
                ("check_termination", C.c_int32), ("scaling_iters", C.c_int32),
                ("adaptive_rho_interval", C.c_int32), ("residual_fp32", C.c_int32), ("reserved_", C.c_int32)]


class SogmWorld(C.Structure):
    _fields_ = [("cloud_xyz", C.c_void_p), ("block_bounds", C.c_void_p), ("cylinders", C.c_void_p),
                ("n_points", C.c_int32), ("n_blocks", C.c_int32), ("block_points", C.c_int32), ("n_cyl", C.c_int32)]


class SogmPrestamp(C.Structure):
    _fields_ = [("cloud_xyz", C.c_void_p), ("cloud_range", C.c_void_p), ("cylinders", C.c_void_p),
                ("n_cyl", C.c_int32), ("reserved_", C.c_int32), ("next_stamp", C.c_double),
                ("replan_start_offset", C.c_double), ("hover_inout", C.c_void_p), ("out_now", C.c_void_p),
                ("out_t_start", C.c_void_p), ("out_pva", C.c_void_p), ("out_poses", C.c_void_p),
                ("world", C.POINTER(SogmWorld))]


class SogmFlight(C.Structure):
    _fields_ = [("n_ticks", C.c_int32), ("first_tick", C.c_int32), ("t0", C.c_double), ("period", C.c_double),
                ("replan_start_offset", C.c_double), ("worlds", C.POINTER(SogmWorld)), ("goals", C.c_void_p),
                ("drone_ids", C.c_void_p), ("hover_inout", C.c_void_p), ("own_inout", C.c_void_p),
                ("tables", C.c_void_p), ("n_total", C.c_int32), ("agent0", C.c_int32), ("log_records", C.c_void_p),
                ("log_ok", C.c_void_p), ("nccl_comm", C.c_void_p)]


FLIGHT_MAX_TICKS = 64
FLIGHT_HDR_ERR, FLIGHT_HDR_FINISHED, FLIGHT_HDR_LATE_WGS = 4, 5, 15  # sogm_flight_stats out_hdr indices
FLIGHT_STAT_NAMES = ("gate_wait", "map", "search", "corridor", "qp", "finish", "chain", "ticks")
TRAJ_RECORD_BYTES = C.sizeof(SogmTrajRecord)  # 2064
CYLINDER_BYTES = C.sizeof(SogmCylinder)  # 96

_vp = C.c_void_p
_i = C.c_int

# name -> (restype, argtypes); every symbol include/sogm_abi.h declares
PROTOTYPES = {
    "sogm_abi_version": (_i, []),
    "sogm_last_error": (C.c_char_p, []),
    "sogm_device_count": (_i, []),
    "sogm_create": (_i, [C.POINTER(SogmSpec), _i, _i, C.POINTER(_vp)]),
    "sogm_destroy": (None, [_vp]),
    "sogm_grid_bytes": (C.c_int64, [_vp]),
    "sogm_grid_ptr": (_vp, [_vp]),
    "sogm_set_sparse_reset": (_i, [_vp, _i, _i]),
    "sogm_sparse_reset_state": (_i, [_vp, _vp]),
    "sogm_grid_history": (_i, [_vp, _vp]),
    "sogm_map_traffic": (_i, [_vp, _vp, _i]),
    "sogm_set_resample": (_i, [_vp, C.c_float, _i, _vp, _i]),
    "sogm_set_tuning": (_i, [_vp, C.c_char_p, C.c_double]),
    "sogm_get_tuning": (_i, [_vp, C.c_char_p, C.POINTER(C.c_double)]),
    "sogm_tuning_key": (C.c_char_p, [_i]),
    "sogm_set_body_particles": (_i, [_vp, C.POINTER(C.c_double), _i]),
    "sogm_set_overlap_clear": (_i, [_vp, _i]),
    "sogm_set_profiling": (_i, [_vp, _i]),
    "sogm_set_profiling_slots": (_i, [_vp, _i]),
    "sogm_profile_read": (_i, [_vp, C.POINTER(C.c_double)]),
    "sogm_profile_read_all": (_i, [_vp, _i, C.POINTER(C.c_double), _i, C.POINTER(_i)]),
    "sogm_update_gt": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "sogm_project_neighbours": (_i, [_vp, _vp, _i, _vp, _vp]),
    "sogm_update_gt_swarm": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "sogm_device_clock": (_i, [_vp, C.POINTER(C.c_int64), _vp]),
    "sogm_tick_clock": (_i, [_vp, C.POINTER(C.c_int64)]),
    "sogm_cloud_block_bounds": (_i, [_vp, _i, _i, _vp, _vp]),
    "sogm_update_world": (_i, [_vp, C.POINTER(SogmWorld), _vp, _vp, _vp, _i, _vp, _vp]),
    "sogm_set_future_risk": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "sogm_download_reference_layout": (_i, [_vp, _i, _vp]),
    "sogm_tick_inputs": (_i, [_vp, _i, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sogm_merge_latest": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "sogm_map_state": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_float), _vp]),
    "sogm_traj_eval": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "sogm_firi_batched": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, C.c_double, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "sogm_planner_select_agents": (_i, [_vp, _i, _i]),
    "sogm_planner_set_search_mode": (_i, [_vp, _i]),
    "sogm_query_clear": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "sogm_obstacle_points": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "sogm_planner_create": (_i, [_vp, C.POINTER(SogmAstarParams), C.POINTER(SogmPlannerParams),
                                 C.POINTER(SogmQpSettings), C.POINTER(_vp)]),
    "sogm_planner_destroy": (None, [_vp]),
    "sogm_astar_search": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "sogm_corridor_generate": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "sogm_bezier_qp_solve": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sogm_bezier_qp_solve_timed": (_i, [_vp, _vp, _vp, _vp, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sogm_linprog_batched": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "sogm_replan": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sogm_gridmap_create": (_i, [C.POINTER(SogmGridMapParams), _i, _i, C.POINTER(_vp)]),
    "sogm_gridmap_destroy": (None, [_vp]),
    "sogm_gridmap_update": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "sogm_gridmap_query_inflate": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "sogm_gridmap_download": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "sogm_gridmap_force_frame": (_i, [_vp, _i]),
    "sogm_traj_safe": (_i, [_vp, _vp, _vp, C.c_double, _vp, _vp]),
    "sogm_safe_after_opt": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "sogm_planner_flow_error": (_i, [_vp]),
    "sogm_planner_flow_failures": (_i, [_vp, _vp]),
    "sogm_planner_set_publish": (_i, [_vp, _vp, _vp]),
    "sogm_planner_set_prestamp": (_i, [_vp, _vp]),
    "sogm_prestamp_pending": (_i, [_vp]),
    "sogm_prestamp_join": (_i, [_vp, _vp]),
    "sogm_update_prestamped": (_i, [_vp, _vp, _i, _vp, _vp]),
    "sogm_planner_counters": (_i, [_vp, C.POINTER(C.c_int64), _i]),
    "sogm_flight_run": (_i, [_vp, C.POINTER(SogmFlight), _vp]),
    "sogm_flight_prepare": (_i, [_vp, _i]),
    "sogm_flight_stats": (_i, [_vp, _vp, _vp]),
    "sogm_planner_set_swarm": (_i, [_vp, _vp, _i, _vp, _vp]),
    "sogm_traj_allgather": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "sogm_exchange_wait": (_i, [_vp, _vp]),
    "sogm_comm_unique_id": (_i, [C.c_char_p]),
    "sogm_comm_create": (_i, [C.c_char_p, _i, _i, _i, C.POINTER(_vp)]),
    "sogm_comm_destroy": (None, [_vp]),
    "sogm_comm_handle": (_vp, [_vp]),
    "sogm_comm_info": (_i, [_vp, _vp]),
    "sogm_filter_point_cloud": (_i, [_vp, _vp, _vp, C.c_float, _i, _vp, _vp, _vp]),
    "sogm_filter_reserve": (_i, [_vp, _i]),
    "sogm_dsp_create": (_i, [_vp, C.POINTER(SogmDspParams), _vp, _vp, _i, _vp, _i, _i, C.POINTER(_vp)]),
    "sogm_dsp_destroy": (None, [_vp]),
    "sogm_update_dsp": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sogm_dsp_publish": (_i, [_vp, _vp, _vp]),
    "sogm_dsp_download_state": (_i, [_vp, _i, _vp, _vp, _vp]),
    "sogm_dsp_download_born": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "sogm_dsp_download_observations": (_i, [_vp, _i, _vp, _vp, _vp]),
}


class SogmError(RuntimeError):
    pass


def load_library(path=LIB_PATH):
    """Load libsogm_hip.so and bind every ABI symbol.  Raises if the extension is missing."""
    if not os.path.exists(path):
        raise SogmError(
            f"HIP extension not built: {path} is missing. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # The hosts above this binding keep device buffers in torch tensors.  torch ships its own
    # libamdhip64; load it first so that the extension binds to the SAME HIP runtime (two runtimes in one
    # process do not see each other's devices or allocations).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.sogm_abi_version()
    if got != SOGM_ABI_VERSION:  # buffer sizes of existing entry points changed between versions: never mix them
        raise SogmError(f"{path} implements ABI version {got}, this binding was written against {SOGM_ABI_VERSION}: "
                        "rebuild the extension (python -c 'import __graft_entry__ as g; g.build()')")
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load_library()
    return _lib


def check(rc, what):
    if rc != SOGM_OK:
        msg = lib().sogm_last_error()
        raise SogmError(f"{what} failed: status {rc} ({msg.decode() if msg else ''})")
