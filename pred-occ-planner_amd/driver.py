"""Batched replan tick driver: the per-tick data flow of the reference swarm, agents sharded over
ranks (one process per GPU).

Per tick and per agent the reference does: map update from the latest cloud + GT state
(FakeParticleRiskVoxel::updateMap incl. the neighbour overlay), BaselinePlanner::replan, then
publishes its BezierTraj on /broadcast_traj (plan_manager/src/plan_manager.cpp:364-399) which every
other agent stores (ParticleATC::trajectoryCallback, traj_coordinator/src/particles.cpp:131-191).
Here: one `sogm_update_gt` + `sogm_project_neighbours` + `sogm_replan` over the rank's agents, then
ONE all-gather of the fixed-size trajectory records (RCCL over xGMI through torch.distributed)
replaces the ROS broadcast; latest-wins per drone_id, a failed replan keeps the previous trajectory.
The replan start state is sampled from the agent's own previous trajectory at t_start
(plan_manager.cpp:169-175); agents without a trajectory hover at their position.
"""
import os

# sogm_replan runs agent groups on separate HIP streams; streams beyond the number of hardware
# queues share a queue and serialise, so ask the runtime for more queues before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("SOGM_GROUPS", "8")

import numpy as np
import torch

from . import _abi, config, scene as scene_mod
from .planner import SogmPlanner, traj_eval
from .sogm import SogmMap, _dev, upload_scene

TICK_PERIOD = 0.1        # fsm/replan_duration (sim_fake.yaml:7)
REPLAN_START_TIME = 0.02  # fsm/replan_start_time (sim_fake.yaml:8)


def shard_bounds(rank, world, agents_per_rank):
    """Contiguous block of agents owned by `rank` (agents are the sharding axis, SURVEY §8 e)."""
    return rank * agents_per_rank, (rank + 1) * agents_per_rank


def exchange_records(own, all_records, dist, world):
    """The trajectory broadcast of the reference (/broadcast_traj, plan_manager.cpp:364-399 ->
    particles.cpp:131-191) as ONE all-gather of fixed-size records.  `own` [A_loc, 2064] uint8,
    `all_records` [A_loc * world, 2064] uint8.  Backend nccl == RCCL on ROCm; gloo on CPU tests."""
    if dist is not None and (world > 1 or dist.is_initialized()):
        dist.all_gather_into_tensor(all_records, own.contiguous())
    else:
        all_records.copy_(own)
    return all_records


def merge_latest(new, old, ok):
    """latest-wins per drone (particles.cpp:179-190); a failed replan keeps the previous trajectory
    (the FSM keeps executing it, plan_manager.cpp:176-196)."""
    return torch.where(ok.bool().unsqueeze(1), new, old)


class SwarmTick:
    def __init__(self, grid="cfg2", agents_per_rank=None, rank=0, world=1, device=0, seed=0x5069,
                 spec=None, scene=None, dist=None, overlap_clear=True, deconflict=True):
        self.rank, self.world, self.dist = rank, world, dist
        self.spec = spec if spec is not None else config.make_spec(grid)
        self.A_loc = agents_per_rank if agents_per_rank is not None else config.AGENTS.get(grid, 4)
        self.A_tot = self.A_loc * world
        half = (self.spec.L // 2) * 0.15
        self.scene = scene if scene is not None else scene_mod.make_scene(self.A_tot, half, seed=seed)
        lo, hi = shard_bounds(rank, world, self.A_loc)
        loc = dict(self.scene)
        loc["n_agents"] = self.A_loc
        for k in ("starts", "goals", "poses", "stamps", "ego_ids"):
            loc[k] = self.scene[k][lo:hi]
        torch.cuda.set_device(device)
        # each agent scans only the cloud around it (its sensing neighbourhood): map half range +
        # the distance it can fly during a run, so per-agent work does not grow with the swarm
        crop, crange = scene_mod.crop_clouds(self.scene, lo, hi, half + 10.0)
        self.dev = upload_scene(loc, cloud=crop, cloud_range=crange)
        self.cloud_points = int(crop.shape[0])
        self.map = SogmMap(self.spec, self.A_loc, device)
        self.map.set_overlap_clear(overlap_clear)
        self.planner = SogmPlanner(self.map, config.make_astar_params(), config.make_planner_params(True),
                                   config.make_qp_settings())
        d = "cuda"
        self.goals = _dev(loc["goals"], np.float64)
        self.hover = _dev(np.concatenate([loc["starts"], np.zeros((self.A_loc, 6))], axis=1), np.float64)
        self.own = torch.zeros((self.A_loc, _abi.TRAJ_RECORD_BYTES), dtype=torch.uint8, device=d)
        self.new = torch.zeros_like(self.own)
        self.ok = torch.zeros((self.A_loc,), dtype=torch.int32, device=d)
        self.all = torch.zeros((self.A_tot, _abi.TRAJ_RECORD_BYTES), dtype=torch.uint8, device=d)
        # replan() ends with ParticleATC::isSafeAfterOpt against the swarm's latest trajectories
        # (baseline_fake.cpp:453-460); "now" of the check = the tick's stamp
        self.now = torch.zeros((self.A_loc,), dtype=torch.float64, device=d)
        if deconflict:
            self.planner.setSwarm(self.all, self.A_tot, self.dev["ego_ids"], self.now)
        self.t0 = float(self.scene["stamps"][0])
        self.tick = 0
        self.n_ok_total = 0

    def close(self):
        self.planner.close()
        self.map.close()

    def step(self):
        """One replan tick for every agent of this rank.  Everything is stream-ordered on the GPU."""
        stamp = self.t0 + self.tick * TICK_PERIOD
        stamps = torch.full((self.A_loc,), stamp, dtype=torch.float64, device="cuda")
        self.now.copy_(stamps)
        t_start = stamps + REPLAN_START_TIME
        pva, valid = traj_eval(self.own, t_start)
        pva = torch.where(valid.bool().unsqueeze(1), pva, self.hover)
        self.hover = torch.cat([pva[:, :3], torch.zeros_like(pva[:, 3:])], dim=1)
        poses = pva[:, :3].to(torch.float32).contiguous()
        self.map.updateMap(self.dev["cloud"], self.dev["cloud_range"], self.dev["cylinders"], self.dev["n_cyl"],
                           poses, stamps)
        self.map.addOtherAgents(self.all, self.A_tot, self.dev["ego_ids"])
        self.planner.replan(pva.contiguous(), self.goals, t_start, self.dev["ego_ids"], self.new, self.ok)
        # latest-wins; a failed replan keeps executing the previous trajectory
        self.own = merge_latest(self.new, self.own, self.ok)
        exchange_records(self.own, self.all, self.dist, self.world)
        self.tick += 1
        return self.ok
