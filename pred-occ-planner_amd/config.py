"""Parameter sets.  Values are the reference's *effective* fkpcp values (SURVEY.md §5 "Config"):
compile-time macros (plan_env/include/plan_env/map_parameters.h:5-19), in-code defaults
(plan_env/src/map.cpp:15-40, plan_env/src/risk_base.cpp:21-23) and
plan_manager/config/sim_fake.yaml — including the YAML keys the code never reads
(`map/risk_threshold` is ignored; `map/risk_threshold_voxel` keeps its 0.2 default, trap 9).
"""
from ._abi import (SOGM_MAP_FAKE, SOGM_MAP_RISKBASE, SogmAstarParams, SogmPlannerParams,
                   SogmQpSettings, SogmSpec)

# BASELINE.json configs -> (L, W, H, T, n_agents)
GRID_CONFIGS = {
    "parity": (66, 66, 20, 6),       # the reference's compiled grid (map_parameters.h:5-13)
    "cfg0": (40, 40, 20, 10),        # configs[0]: 1 agent, CPU-runnable
    "cfg1": (100, 100, 100, 15),     # configs[1]: 16 agents
    "cfg2": (200, 200, 200, 20),     # configs[2]: 128 agents (the headline metric)
    "cfg4": (300, 300, 300, 30),     # configs[4]
}
AGENTS = {"parity": 4, "cfg0": 1, "cfg1": 16, "cfg2": 128, "cfg4": 128}


def make_spec(grid="parity", map_kind=SOGM_MAP_FAKE, clearance=0.45, time_resolution=0.2, storage=None, tiled=None):
    L, W, H, T = GRID_CONFIGS[grid] if isinstance(grid, str) else grid
    s = SogmSpec()
    s.L, s.W, s.H, s.T = L, W, H, T
    s.resolution = 0.15               # VOXEL_RESOLUTION
    s.time_resolution = time_resolution  # map/time_resolution (sim_fake.yaml:51)
    s.risk_threshold = 0.2            # map/risk_threshold_voxel default (map.cpp:35)
    s.clearance = clearance           # sim_fake.yaml:71  (=> inf_step 2 in fp32, trap 10)
    s.ground_height = -0.01           # sim_fake.yaml:68
    s.ceiling_height = 3.0            # sim_fake.yaml:69
    s.risk_threshold_region = 1.2     # risk_base.cpp:21
    s.risk_thres_reg_decay = 0.2      # risk_base.cpp:22
    s.risk_thres_vox_decay = 0.2      # risk_base.cpp:23
    s.map_kind = map_kind
    # BASELINE configs[4] (300^3 x 30, 128 agents) only fits 288 GB of HBM with fp16 occupancy cells
    s.storage = (1 if grid == "cfg4" else 0) if storage is None else storage
    # cell order of a slice: 2 x 2 x 2 tiles (SOGM_LAYOUT_TILED; the default for fake-perception maps with even sizes: the
    # stamp writes and the reset zeroes about half the sectors — 1.28 against 1.62 ms, 0.58 against 1.04 ms at 128 agents)
    # or x-fastest rows (SOGM_LAYOUT=rows in the environment, A/B runs; odd sizes; an explicit `storage`).
    if tiled is None:
        import os
        tiled = (os.environ.get("SOGM_LAYOUT", "tiled") == "tiled" and map_kind == SOGM_MAP_FAKE
                 and not ((L | W | H) & 1) and storage is None)
    if tiled:
        s.storage |= 16
    return s


def make_astar_params(fake=True):
    """fake: FakeRiskHybridAstar (shot check with time) / RiskHybridAstar (time-less shot check)."""
    p = SogmAstarParams()
    p.shot_ignores_time = 0 if fake else 1
    p.max_tau = 2.0
    p.max_vel = 2.0
    p.max_acc = 6.0
    p.w_time = 5.0
    p.horizon = 5.0
    p.lambda_heu = 5.0
    p.resolution = 0.15
    p.time_resolution = 0.3
    p.allocate_num = 10000
    p.check_num = 1
    p.tolerance = 1
    return p


def make_planner_params(fake=True):
    p = SogmPlannerParams()
    p.corridor_tau = 0.3
    p.init_range = 1.2
    p.shrink_size = 0.2
    p.opt_max_vel = 2.0
    p.opt_max_acc = 6.0
    p.fake_planner = 1 if fake else 0
    p.firi_iterations = 2
    p.pc_capacity = 16384  # obstacle points per corridor box (the reference grows its vector, baseline.cpp:317)
    p.max_faces = 64
    return p


def make_qp_settings():
    """OSQP v0.6 defaults with eps 1e-3 (bezier_optimizer.cpp:269); adaptive rho every 25 iterations."""
    q = SogmQpSettings()
    q.rho = 0.1
    q.sigma = 1e-6
    q.alpha = 1.6
    q.eps_abs = 1e-3
    q.eps_rel = 1e-3
    q.max_iter = 4000
    q.check_termination = 25
    q.scaling_iters = 10
    q.adaptive_rho_interval = 25
    q.residual_fp32 = 0  # 1: BASELINE configs[4]'s "mixed-precision ADMM residuals" (see sogm_abi.h)
    return q


DRONE_SIZE = (0.4, 0.4, 0.45)  # swarm/drone_size_* (sim_fake.yaml:85-87)
