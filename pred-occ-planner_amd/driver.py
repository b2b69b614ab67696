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
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np
import torch

from . import _abi, config, scene as scene_mod
from .planner import SogmPlanner, traj_eval
from .sogm import SogmMap, World, _dev, _stream, upload_scene

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


class RecordExchange:
    """The same all-gather behind the C ABI (sogm_traj_allgather: ncclAllGather on the context's exchange
    stream, consumers wait for it inside the library) — what a C++ host linking libsogm_hip.so uses.  Taken when
    torch.distributed runs on RCCL (backend "nccl"); the communicator is created through sogm_comm_* with the
    128-byte id broadcast from rank 0 (dist.broadcast_object_list, like ncclGetUniqueId + MPI_Bcast in an MPI host).
    Anything else (gloo on CPU tests, no process group) uses exchange_records() above; if the communicator cannot be
    created the fallback is torch.distributed's own all-gather and `self.fallback_reason` says why (bench.py prints
    it: never silent).  `lib` / `backends` / `stream` are injection points for the CPU test, which runs this class
    against a stub library at world size 2 (tests/test_driver_gloo.py)."""

    def __init__(self, ctx, dist, rank, world, device, lib=None, backends=("nccl",), stream=None):
        import ctypes as C
        import sys
        self.ctx, self.comm, self.handle, self.fallback_reason = ctx, C.c_void_p(), None, None
        self._stream = stream if stream is not None else _stream
        self.lib = None
        if not (dist is not None and dist.is_initialized() and dist.get_backend() in backends):
            return
        if os.environ.get("SOGM_EXCHANGE", "abi") != "abi":
            self.fallback_reason = "SOGM_EXCHANGE != abi"
            return
        self.lib = lib if lib is not None else _abi.lib()
        ident = [None]
        if rank == 0:
            buf = C.create_string_buffer(_abi.SOGM_COMM_ID_BYTES)
            rc = self.lib.sogm_comm_unique_id(buf)
            ident = [buf.raw if rc == 0 else rc]
        dist.broadcast_object_list(ident, src=0)   # every rank takes part, whatever rank 0 got
        rc = ident[0] if isinstance(ident[0], int) else self.lib.sogm_comm_create(
            ident[0], rank, world, device, C.byref(self.comm))
        # all ranks must agree on the path: one failed communicator sends everybody to the fallback
        flags = [None] * world
        dist.all_gather_object(flags, int(rc))
        if any(f != 0 for f in flags):
            if rc == 0:
                self.lib.sogm_comm_destroy(self.comm)
            self.comm, self.fallback_reason = C.c_void_p(), f"sogm_comm_create returned {flags} (per rank)"
            print(f"[sogm] rank {rank}: RCCL communicator through the C ABI unavailable: {self.fallback_reason}; "
                  "using torch.distributed's all-gather", file=sys.stderr)
            return
        self.handle = self.lib.sogm_comm_handle(self.comm)

    @property
    def active(self):
        return self.handle is not None

    def info(self):
        """{ranks, rank} as RCCL itself reports them for the communicator (ncclCommCount / ncclCommUserRank)."""
        import ctypes as C
        out = (C.c_int32 * 2)()
        _abi.check(self.lib.sogm_comm_info(self.comm, out), "sogm_comm_info")
        return {"ranks": int(out[0]), "rank": int(out[1])}

    def all_gather(self, own, all_records):
        _abi.check(self.lib.sogm_traj_allgather(self.ctx, self.handle, own.data_ptr(), own.shape[0],
                                                all_records.data_ptr(), self._stream()), "sogm_traj_allgather")

    def wait(self):
        """Make torch's current stream wait for an all-gather in flight (before torch ops read the records)."""
        _abi.check(self.lib.sogm_exchange_wait(self.ctx, self._stream()), "sogm_exchange_wait")

    def close(self):
        if self.comm:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            self.lib.sogm_comm_destroy(self.comm)
            self.comm = None
            self.handle = None


class LoopbackHub:
    """Several ranks of ONE swarm flown by one process on one device, one rank after the other within a tick (bench.py's
    configs[3] rank share; tests/test_rank_share_gpu.py): stands where the collective stands.  Each rank's SwarmTick
    gets hub.exchange(rank); its all_gather() only STAGES the rank's rows, and hub.commit() — called once every
    co-simulated rank has stepped tick k — writes every staged row block into every rank's table.  So the table a rank
    reads during tick k is ver(k - 1) of all co-simulated ranks, exactly what the all-gather leaves behind
    (plan_manager.cpp:364-399 -> particles.cpp:131-191, one tick stale); rows of ranks that are not co-simulated stay
    empty records ("nothing received from that drone yet")."""

    def __init__(self, world, agents_per_rank):
        self.world, self.A_loc, self.staged = world, agents_per_rank, {}

    def exchange(self, rank):
        return _LoopbackExchange(self, rank)

    def commit(self, replayed=None):
        """`replayed`: {rank: rows} of ranks that are not flown live this time — their rows of this tick from an earlier
        pass (tools/bench_rank_share.py: one process cannot hold three planners' streams without sharing hardware
        queues, so one ring neighbour flies live and the other is replayed)."""
        rows = {s: own for s, (own, _) in self.staged.items()}
        rows.update(replayed or {})
        for _, (_, table) in self.staged.items():
            for s, own in rows.items():
                lo, hi = shard_bounds(s, self.world, self.A_loc)
                table[lo:hi].copy_(own)
        self.staged = {}


class _LoopbackExchange:
    active, fallback_reason = True, None

    def __init__(self, hub, rank):
        self.hub, self.rank = hub, rank

    def info(self):
        return {"ranks": self.hub.world, "rank": self.rank}

    def all_gather(self, own, all_records):
        self.hub.staged[self.rank] = (own, all_records)

    def wait(self):
        pass

    def close(self):
        pass


def merge_latest(new, old, ok):
    """latest-wins per drone (particles.cpp:179-190); a failed replan keeps the previous trajectory
    (the FSM keeps executing it, plan_manager.cpp:176-196)."""
    return torch.where(ok.bool().unsqueeze(1), new, old)


# ---- FiniteStateMachine::FSMCallback, per-agent state machines as tensors ----------------
FSM_NEW_PLAN, FSM_EXEC_TRAJ, FSM_REPLAN, FSM_GOAL_REACHED = 0, 1, 2, 3
REPLAN_MAX_FAILURES = 5      # fsm/replan_max_failures (sim_fake.yaml:10)
COLLI_CHECK_DURATION = 0.2   # fsm/colli_check_duration (sim_fake.yaml:9)
GOAL_TOLERANCE = 1.0         # fsm/goal_tolerance (sim_fake.yaml:5)


def hover_records(drone_ids, start_time, pos):
    """publishEmptyTrajectory (plan_manager/src/plan_manager.cpp:404-424): one 0.5 s piece whose five control
    points all sit at the current position.  Returns uint8 [n, 2064] in SogmTrajRecord layout."""
    n, dev = pos.shape[0], pos.device
    dur = torch.zeros((n, _abi.SOGM_MAX_PIECES), dtype=torch.float64, device=dev)
    dur[:, 0] = 0.5
    cpts = torch.zeros((n, _abi.SOGM_MAX_PIECES * 15), dtype=torch.float64, device=dev)
    cpts[:, :15] = pos.to(torch.float64).repeat(1, 5)
    head = torch.stack([drone_ids.to(torch.int32), torch.ones_like(drone_ids, dtype=torch.int32)], dim=1)
    parts = [head.contiguous().view(torch.uint8).view(n, 8),
             start_time.to(torch.float64).contiguous().view(torch.uint8).view(n, 8),
             dur.view(torch.uint8).view(n, -1), cpts.view(torch.uint8).view(n, -1)]
    return torch.cat(parts, dim=1)


def fsm_plan_inputs(status, traj_start, now):
    """Which agents plan in this FSM tick and from which time (plan_manager.cpp:110-135,165-175):
    NEW_PLAN plans from `now` once a second, REPLAN from now + replan_start_time."""
    is_new, is_rep = status == FSM_NEW_PLAN, status == FSM_REPLAN
    due_new = is_new & ((now - traj_start) > 1.0)
    t_start = torch.where(is_rep, now + REPLAN_START_TIME, now)
    return due_new, is_rep, t_start


def fsm_apply(status, fail, traj_start, success, now, due_new, is_rep, ok, safe, reached):
    """State update of one FSMCallback per agent (plan_manager.cpp:92-233) given this tick's replan results.
    Returns (status, fail, traj_start, success, publish_new, publish_hover, hover_start).  `safe` is isTrajSafe,
    `reached` isGoalReached; both are only consulted in EXEC_TRAJ."""
    ok = ok.bool()
    is_new, is_exec = status == FSM_NEW_PLAN, status == FSM_EXEC_TRAJ
    # NEW_PLAN (:110-135)
    traj_start = torch.where(due_new, now, traj_start)
    success = torch.where(due_new, ok, success.bool())
    pub_new = due_new & ok
    pub_hover = due_new & ~ok
    hover_start = traj_start.clone()
    st = torch.where(is_new & success, torch.full_like(status, FSM_EXEC_TRAJ), status)
    # REPLAN (:164-199)
    t_rep = now + REPLAN_START_TIME
    traj_start = torch.where(is_rep, t_rep, traj_start)
    ok_rep = is_rep & ok
    fail = torch.where(ok_rep, torch.zeros_like(fail), torch.where(is_rep, fail + 1, fail))
    pub_new = pub_new | ok_rep
    st = torch.where(ok_rep, torch.full_like(status, FSM_EXEC_TRAJ), st)
    over = is_rep & ~ok & (fail > REPLAN_MAX_FAILURES)
    st = torch.where(over, torch.full_like(status, FSM_NEW_PLAN), st)
    pub_hover = pub_hover | over
    hover_start = torch.where(over, t_rep, hover_start)
    traj_start = torch.where(over, now - 1.0, traj_start)  # force a new plan on the next tick
    # EXEC_TRAJ (:137-162)
    lapse = is_exec & ((now - traj_start) > TICK_PERIOD)   # fsm/replan_duration
    st = torch.where(lapse | (is_exec & ~safe.bool()), torch.full_like(status, FSM_REPLAN), st)
    st = torch.where(is_exec & reached.bool(), torch.full_like(status, FSM_GOAL_REACHED), st)
    return st, fail, traj_start, success, pub_new, pub_hover, hover_start


class HipCompute:
    """The rank's kernels behind the C ABI: the four calls one tick of SwarmTick.step() makes.  The product path.
    (tests/test_driver_gloo.py runs the same SwarmTick.step() over gloo on CPU with the oracle standing in for this
    class — the rank-local bookkeeping around these calls is what that test covers.)"""
    device = "cuda"

    def __init__(self, spec, scene, lo, hi, device, overlap_clear=True, double_buffer=None, grids=None, tuning=None,
                 moving_world=None):
        half = (spec.L // 2) * 0.15
        A_loc = hi - lo
        loc = dict(scene)
        loc["n_agents"] = A_loc
        for k in ("starts", "goals", "poses", "stamps", "ego_ids"):
            loc[k] = scene[k][lo:hi]
        torch.cuda.set_device(device)
        # World frames (moving_world True / False): every tick's update takes the sensor frame OF THAT TICK (the whole
        # cloud + the cylinders, scene.WorldTimeline) and crops it on the device around each agent's current map centre
        # (sogm_update_world).  None: the frozen scene of the earlier rounds with host-side crops cut once around the
        # start (sogm_update_gt_swarm) — kept for the tests that compare single builds.
        self.timeline = scene_mod.WorldTimeline(scene, TICK_PERIOD, moving=bool(moving_world))
        self.use_world = moving_world is not None
        self._frames = {}
        if self.use_world:
            self.dev = upload_scene(loc, cloud=np.zeros((1, 3), np.float32), cloud_range=np.zeros((A_loc, 2), np.int32))
            self.cloud_points = 0   # (per-agent counts are the device-side crop's: sogm_map_traffic / bench)
        else:
            # each agent scans only the cloud around it (its sensing neighbourhood): map half range +
            # the distance it can fly during a run, so per-agent work does not grow with the swarm
            crop, crange = scene_mod.crop_clouds(scene, lo, hi, half + 10.0)
            self.dev = upload_scene(loc, cloud=crop, cloud_range=crange)
            self.cloud_points = int(crop.shape[0])
        self.A_loc = A_loc
        self.map = SogmMap(spec, A_loc, device)
        self.overlap_mode = self.map.set_overlap_clear(overlap_clear, double_buffer=double_buffer, grids=grids)
        if "groups=" not in os.environ.get("SOGM_TUNING", ""):
            self.map.set_tuning("groups", 8)  # the grouped-stream replan (SOGM_FLOW=0): eight agent groups (32 hardware queues)
        for k, v in (tuning or {}).items():
            self.map.set_tuning(k, v)
        self.planner = SogmPlanner(self.map, config.make_astar_params(), config.make_planner_params(True),
                                   config.make_qp_settings())
        self.ego_ids = self.dev["ego_ids"]
        self.fused_update = os.environ.get("SOGM_FUSED_UPDATE", "1") != "0"

    def world(self, k):
        """The device copy of tick k's sensor frame (uploaded on first use; prepare() uploads a range ahead of a timed
        region so that no host-to-device copy falls inside it).  A frozen world has one frame."""
        key = k if self.timeline.moving else 0
        w = self._frames.get(key)
        if w is None:
            f = self.timeline.frame(key)
            w = self._frames[key] = World(f["cloud"], f["cylinders"])
        return w

    def prepare(self, k0, k1):
        for k in range(k0, k1):
            self.world(k)
        torch.cuda.synchronize()

    def forget(self, before):
        """drop the uploaded frames of ticks < before"""
        for k in [k for k in self._frames if k < before and self.timeline.moving]:
            del self._frames[k]

    @property
    def ctx(self):
        return self.map.ctx

    def set_swarm(self, all_records, A_tot, now):
        self.planner.setSwarm(all_records, A_tot, self.ego_ids, now)

    def set_publish(self, own, next_table):
        """replan() merges the new records into `own` (latest wins) and writes every agent's current record into
        `next_table` itself (sogm_planner_set_publish) — no merge launch after the replan"""
        self.planner.setPublish(own, next_table)

    def tick_inputs(self, own, stamp, hover, now, t_start, pva, poses):
        """start states from the executed trajectories, stamps and map centres: one launch (sogm_tick_inputs)"""
        _abi.check(_abi.lib().sogm_tick_inputs(own.data_ptr(), self.A_loc, stamp, REPLAN_START_TIME, hover.data_ptr(),
                                               now.data_ptr(), t_start.data_ptr(), pva.data_ptr(), poses.data_ptr(),
                                               _stream()), "sogm_tick_inputs")

    def update_map(self, poses, now, all_records, A_tot, tick=0):
        """updateMap incl. its closing neighbour overlay in one call, from the sensor frame of `tick`"""
        d = self.dev
        if self.use_world:
            self.map.updateWorld(self.world(tick), poses, now, all_records, A_tot, self.ego_ids)
        elif self.fused_update:
            self.map.updateMapSwarm(d["cloud"], d["cloud_range"], d["cylinders"], d["n_cyl"], poses, now,
                                    all_records, A_tot, self.ego_ids)
        else:  # the two separate calls (SOGM_FUSED_UPDATE=0: A/B aid)
            self.map.updateMap(d["cloud"], d["cloud_range"], d["cylinders"], d["n_cyl"], poses, now)
            self.map.addOtherAgents(all_records, A_tot, self.ego_ids)

    def set_prestamp(self, next_stamp, hover, now, t_start, pva, poses, tick=0):
        """the replan about to run (tick `tick`) also builds the NEXT tick's map and start states
        (sogm_planner_set_prestamp) — from the newest sensor frame that exists while it runs, its own tick's: the next
        tick then plans on a map whose obstacle data is one tick old (staleness 1; a frozen world hides the difference)"""
        d = self.dev
        if self.use_world:
            self.planner.setPrestamp(None, None, None, 0, next_stamp, REPLAN_START_TIME, hover, now, t_start, pva, poses,
                                     world=self.world(tick))
        else:
            self.planner.setPrestamp(d["cloud"], d["cloud_range"], d["cylinders"], d["n_cyl"], next_stamp,
                                     REPLAN_START_TIME, hover, now, t_start, pva, poses)

    def prestamp_pending(self):
        return self.map.prestamp_pending()

    def update_prestamped(self, all_records, A_tot):
        """the update of a pre-stamped tick: grid swap + neighbour overlay"""
        self.map.updatePrestamped(all_records, A_tot, self.ego_ids)

    def replan(self, pva, goals, t_start, new, ok):
        # a dataflow replan whose device-side waits timed out reports ok = 0 for the agents it could not finish and
        # counts the tick in pinned memory: a flight must not go on merging records past such a tick
        code, n = self.planner.flow_failures()
        if n:
            raise RuntimeError(f"sogm_replan: {n} tick(s) failed on the device (flow error code {code})")
        self.planner.replan(pva, goals, t_start, self.ego_ids, new, ok)

    def merge_latest(self, new, ok, own, all_records):
        """latest-wins in one launch; `all_records` (single process only) refreshes the swarm table as well"""
        _abi.check(_abi.lib().sogm_merge_latest(new.data_ptr(), ok.data_ptr(), own.data_ptr(),
                                                all_records.data_ptr() if all_records is not None else None,
                                                self.A_loc, _stream()), "sogm_merge_latest")

    def close(self):
        self.planner.close()
        self.map.close()


class SwarmTick:
    def __init__(self, grid="cfg2", agents_per_rank=None, rank=0, world=1, device=0, seed=0x5069,
                 spec=None, scene=None, dist=None, overlap_clear=True, deconflict=True, fsm=False,
                 double_buffer=None, grids=None, compute=None, exchange=None, prestamp=None, tuning=None,
                 moving_world=None, neighbour_lag=1):
        self.rank, self.world, self.dist = rank, world, dist
        # neighbour_lag = 2: the overlay and isSafeAfterOpt of tick k read table ver(k - 2) instead of ver(k - 1) — the
        # staleness rule of sogm_flight_run, flown here lock-step (one tick after the other) through the per-tick entry
        # points: the reference path the flight's records are held to, bit for bit (single process, no fused publication)
        self.neighbour_lag = int(neighbour_lag)
        assert self.neighbour_lag in (1, 2)
        self.spec = spec if spec is not None else config.make_spec(grid)
        self.A_loc = agents_per_rank if agents_per_rank is not None else config.AGENTS.get(grid, 4)
        self.A_tot = self.A_loc * world
        half = (self.spec.L // 2) * 0.15
        self.scene = scene if scene is not None else scene_mod.make_scene(self.A_tot, half, seed=seed)
        lo, hi = shard_bounds(rank, world, self.A_loc)
        if world > 1 and compute is None and dist is not None and "flight_engines" not in (tuning or {}) \
                and "flight_exchange_units" not in (tuning or {}):
            # several ranks, one GPU each: a flight's four kernels fill every compute unit, and the collective queued behind the
            # call (SogmFlight::nccl_comm) runs kernels of its own — one 16-CU unit is left to no kernel (the map kernel gives it
            # up) so that they find room.  NOT measured on hardware: no multi-GPU box.
            tuning = dict(tuning or {}, flight_exchange_units=1, flight_map_units=3)
        self.compute = compute if compute is not None else HipCompute(
            self.spec, self.scene, lo, hi, device, overlap_clear, double_buffer, grids, tuning, moving_world)
        c = self.compute
        # the HIP objects, for the bench / tools / tests that use the staged entry points beside step()
        self.map, self.planner, self.dev = getattr(c, "map", None), getattr(c, "planner", None), getattr(c, "dev", None)
        self.overlap_mode, self.cloud_points = getattr(c, "overlap_mode", 0), getattr(c, "cloud_points", 0)
        d = c.device
        self.goals = _dev(self.scene["goals"][lo:hi], np.float64, d)
        self.hover = _dev(np.concatenate([self.scene["starts"][lo:hi], np.zeros((self.A_loc, 6))], axis=1), np.float64, d)
        self.pva = torch.zeros((self.A_loc, 9), dtype=torch.float64, device=d)      # replan start states of the tick
        self.poses = torch.zeros((self.A_loc, 3), dtype=torch.float32, device=d)    # map centres of the tick
        self.t_start = torch.zeros((self.A_loc,), dtype=torch.float64, device=d)
        self.own = torch.zeros((self.A_loc, _abi.TRAJ_RECORD_BYTES), dtype=torch.uint8, device=d)
        self.new = torch.zeros_like(self.own)
        self.ok = torch.zeros((self.A_loc,), dtype=torch.int32, device=d)
        self.all = torch.zeros((self.A_tot, _abi.TRAJ_RECORD_BYTES), dtype=torch.uint8, device=d)
        # replan() ends with ParticleATC::isSafeAfterOpt against the swarm's latest trajectories
        # (baseline_fake.cpp:453-460); "now" of the check = the tick's stamp
        self.now = torch.zeros((self.A_loc,), dtype=torch.float64, device=d)
        self.deconflict = deconflict
        if deconflict:
            c.set_swarm(self.all, self.A_tot, self.now)
        # publication inside the replan (default; SOGM_PUBLISH=0: the separate sogm_merge_latest launch): a single
        # process alternates between two swarm tables — the replan reads one (overlay, deconfliction) and its
        # finishing kernel fills the other for the next tick
        self.publish = (os.environ.get("SOGM_PUBLISH", "1") != "0" and hasattr(c, "set_publish") and not fsm
                        and self.neighbour_lag == 1)
        self._tables = [self.all, torch.zeros_like(self.all)] if self.publish else None
        # pre-stamp (default; SOGM_PRESTAMP=0 or prestamp=False: off): every replan also builds the next tick's map and
        # start states, agent by agent as their records are published; the tick's inputs are double-buffered (the replan
        # in flight reads one set)
        # With world frames the default is OFF: a pre-stamp can only be fed the frame of the tick in flight, so the map the
        # next tick plans on is one tick staler than the reference's (map_input_staleness_ticks 1 vs 0) — an opt-in variant.
        self.moving_world = moving_world
        default_on = moving_world is None
        want = (os.environ.get("SOGM_PRESTAMP", "1" if default_on else "0") != "0") if prestamp is None else bool(prestamp)
        self.prestamp = want and self.publish and hasattr(c, "set_prestamp") and self.overlap_mode >= 2
        # ticks between the sensor frame a tick's map is built from and the tick itself
        self.map_input_staleness_ticks = 1 if (self.prestamp and moving_world is not None) else 0
        self._alt = (torch.zeros_like(self.pva), torch.zeros_like(self.t_start), torch.zeros_like(self.now),
                     torch.zeros_like(self.poses)) if self.prestamp else None
        # optional closed-loop mode: every agent runs the reference's FiniteStateMachine (step_fsm)
        self.fsm = fsm
        self.status = torch.full((self.A_loc,), FSM_NEW_PLAN, dtype=torch.int32, device=d)
        self.fail = torch.zeros((self.A_loc,), dtype=torch.int32, device=d)
        self.success = torch.zeros((self.A_loc,), dtype=torch.bool, device=d)
        self.traj_start = torch.full((self.A_loc,), float(self.scene["stamps"][0]) - 2.0, dtype=torch.float64, device=d)
        self.t0 = float(self.scene["stamps"][0])
        self.tick = 0
        self.n_ok_total = 0
        # several ranks (or a process group of one): the swarm table is refreshed by the all-gather, not locally
        self.distributed = dist is not None and (world > 1 or dist.is_initialized())
        self.exchange = exchange if exchange is not None else RecordExchange(c.ctx, dist, rank, world, device)

    def set_prestamp(self, on):
        """switch the pre-stamp on / off between two ticks of a flight (bench.py's labelled variant on the same swarm)"""
        c = self.compute
        on = bool(on) and self.publish and hasattr(c, "set_prestamp") and self.overlap_mode >= 2
        if on and self._alt is None:
            self._alt = (torch.zeros_like(self.pva), torch.zeros_like(self.t_start), torch.zeros_like(self.now),
                         torch.zeros_like(self.poses))
        if not on and self.prestamp:
            torch.cuda.synchronize()
            if c.prestamp_pending():   # a grid stamped for a tick that will now be built at its own start: drop it
                self.map.prestamp_join()
            self.planner.setPrestamp(None, None, None, 0, 0.0, 0.0, None, None, None, None)
        self.prestamp = on
        self.map_input_staleness_ticks = 1 if (on and self.moving_world is not None) else 0
        return on

    def close(self):
        self.exchange.close()
        self.compute.close()

    def _exchange(self):
        """The tick's trajectory broadcast: one RCCL all-gather (behind the C ABI when available)."""
        if self.exchange.active:
            self.own = self.own.contiguous()
            self.exchange.all_gather(self.own, self.all)
        else:
            exchange_records(self.own, self.all, self.dist, self.world)

    def _publish(self):
        """End of a tick: latest-wins merge of the new records into the rank's own table (a failed replan keeps
        executing the previous trajectory, plan_manager.cpp:176-196), then the broadcast (plan_manager.cpp:364-399
        -> particles.cpp:131-191): a single process refreshes the swarm table in the same launch, several ranks
        all-gather it — the records every rank overlays on its NEXT tick are one tick stale, like the ROS topic."""
        local = not self.exchange.active and not self.distributed
        self.compute.merge_latest(self.new, self.ok, self.own, self.all if local else None)
        if not local:
            self._exchange()

    def records_all(self):
        """The swarm's latest records for torch-side readers (waits for an all-gather in flight)."""
        if self.exchange.active:
            self.exchange.wait()
        return self.all

    def step_fsm(self):
        """One FSM tick (plan_manager.cpp:92-233) for every agent: NEW_PLAN / REPLAN agents plan, EXEC_TRAJ
        agents only check time lapse, isTrajSafe and the goal; failures are counted and after
        replan_max_failures the agent publishes a hover record and starts over.  Stream-ordered, no host sync."""
        stamp = self.t0 + self.tick * TICK_PERIOD
        now = torch.full((self.A_loc,), stamp, dtype=torch.float64, device="cuda")
        self.now.copy_(now)
        due_new, is_rep, t_start = fsm_plan_inputs(self.status, self.traj_start, now)
        pva_now, valid_now = traj_eval(self.own, now)
        pva_now = torch.where(valid_now.bool().unsqueeze(1), pva_now, self.hover)
        pva, valid = traj_eval(self.own, t_start)
        pva = torch.where(valid.bool().unsqueeze(1), pva, self.hover).contiguous()
        self.hover = torch.cat([pva_now[:, :3], torch.zeros_like(pva_now[:, 3:])], dim=1)
        if getattr(self.compute, "use_world", False):
            self.map.updateWorld(self.compute.world(self.tick), pva_now[:, :3].to(torch.float32).contiguous(), now)
        else:
            self.map.updateMap(self.dev["cloud"], self.dev["cloud_range"], self.dev["cylinders"], self.dev["n_cyl"],
                               pva_now[:, :3].to(torch.float32).contiguous(), now)
        self.map.addOtherAgents(self.all, self.A_tot, self.dev["ego_ids"])
        safe = self.map.isTrajSafe(self.own, now, COLLI_CHECK_DURATION)
        self.planner.replan(pva, self.goals, t_start, self.dev["ego_ids"], self.new, self.ok)
        reached = (pva_now[:, :3] - self.goals).norm(dim=1) < GOAL_TOLERANCE
        ok = self.ok.bool() & (due_new | is_rep)
        (self.status, self.fail, self.traj_start, self.success, pub_new, pub_hover, hover_start) = fsm_apply(
            self.status, self.fail, self.traj_start, self.success, now, due_new, is_rep, ok, safe, reached)
        hover = hover_records(self.dev["ego_ids"], hover_start, pva_now[:, :3])
        self.own = torch.where(pub_new.unsqueeze(1), self.new, torch.where(pub_hover.unsqueeze(1), hover, self.own))
        # what this FSMCallback saw and did (device tensors; tests compare them with oracle/fsm_oracle.cpp)
        self.last_fsm = {"now": stamp, "ok": ok, "safe": safe, "reached": reached, "pub_new": pub_new,
                         "pub_hover": pub_hover, "hover_start": hover_start, "t_start": t_start, "pos": pva_now[:, :3]}
        self._exchange()
        self.tick += 1
        return ok.to(torch.int32)

    def step(self):
        """One replan tick for every agent of this rank.  Everything is stream-ordered on the GPU."""
        if self.fsm:
            return self.step_fsm()
        stamp = self.t0 + self.tick * TICK_PERIOD
        c = self.compute
        if self.prestamp and c.prestamp_pending():
            # the previous replan built this tick's map and start states (into the alternate buffers)
            (self.pva, self.t_start, self.now, self.poses), self._alt = self._alt, (self.pva, self.t_start, self.now,
                                                                                    self.poses)
            c.update_prestamped(self.all, self.A_tot)
        else:
            c.tick_inputs(self.own, stamp, self.hover, self.now, self.t_start, self.pva, self.poses)
            c.update_map(self.poses, self.now, self.all, self.A_tot, self.tick)
        if self.publish:
            local = not self.exchange.active and not self.distributed
            nxt = self._tables[(self.tick + 1) & 1] if local else None
            c.set_publish(self.own, nxt)
            if self.deconflict:
                c.set_swarm(self.all, self.A_tot, self.now)
            if self.prestamp:
                c.set_prestamp(self.t0 + (self.tick + 1) * TICK_PERIOD, self.hover, self._alt[2], self._alt[1],
                               self._alt[0], self._alt[3], self.tick)
            c.replan(self.pva, self.goals, self.t_start, self.new, self.ok)
            if local:
                self.all = nxt   # what every agent executes after this tick: the next tick's table
            else:
                self._exchange()
        elif self.neighbour_lag == 2:
            c.replan(self.pva, self.goals, self.t_start, self.new, self.ok)
            c.merge_latest(self.new, self.ok, self.own, None)   # own = ver(k)'s rows
            # self.all stays ver(k - 2) for this tick's readers; the next tick reads ver(k - 1) = the table of one tick ago
            self._lag_prev, self.all = self.own.clone(), (self._lag_prev if getattr(self, "_lag_prev", None) is not None
                                                          else torch.zeros_like(self.all))
            if self.deconflict:
                c.set_swarm(self.all, self.A_tot, self.now)
        else:
            c.replan(self.pva, self.goals, self.t_start, self.new, self.ok)
            self._publish()
        self.tick += 1
        return self.ok.clone()  # self.ok is rewritten by the next tick

    # ---- flights: sogm_flight_run, every agent on its own clock ----
    def fly(self, n_ticks):
        """The next n_ticks ticks of every agent in ONE call (sogm_abi.h "Flight"): agent a's tick k starts when its own tick
        k - 1 is finished and every agent has finished tick k - 2; it reads the neighbours' records of tick k - 2.  Needs
        world frames (moving_world True / False) and a flight flown from tick 0 through fly() only.  Returns (ok [n, A]
        int32, records uint8 [n, A, 2064]) device tensors — valid once the stream has run the flight."""
        c = self.compute
        assert getattr(c, "use_world", False), "fly() needs world frames (moving_world=True / False)"
        assert not self.fsm
        if self.world > 1:
            return self._fly_ranks(n_ticks)
        if not hasattr(self, "_fl_tables"):
            assert self.tick == 0, "a flight starts at tick 0 (or continues a flight)"
            self._fl_tables = torch.zeros((4, self.A_tot, _abi.TRAJ_RECORD_BYTES), dtype=torch.uint8, device="cuda")
            self._fl_next = 0
        assert self.tick == self._fl_next, "fly() continues flights only"
        assert 1 <= n_ticks <= _abi.FLIGHT_MAX_TICKS
        log_r = torch.zeros((n_ticks, self.A_loc, _abi.TRAJ_RECORD_BYTES), dtype=torch.uint8, device="cuda")
        log_ok = torch.zeros((n_ticks, self.A_loc), dtype=torch.int32, device="cuda")
        worlds = [c.world(self.tick + i) for i in range(n_ticks)]
        self.planner.flight(worlds, self.tick, self.t0, TICK_PERIOD, REPLAN_START_TIME, self.goals, self.dev["ego_ids"],
                            self.hover, self.own, self._fl_tables, log_r, log_ok)
        self.tick += n_ticks
        self._fl_next = self.tick
        self.all = self._fl_tables[(self.tick - 1) & 3]   # ver(last tick): what every agent executes now
        return log_ok, log_r

    def _fly_ranks(self, n_ticks):
        """fly() over several ranks: tick k reads the OTHER ranks' records of tick k - 2, which only an all-gather between
        two calls can deliver — so the flight goes in calls of two ticks (k, k + 1 read ver(k - 2), ver(k - 1): both
        complete before the call), each followed by the all-gather of the two versions it finished (this rank's rows of
        ver(k), ver(k + 1) to every rank).  Same records as one process flying all agents."""
        c = self.compute
        lo, hi = shard_bounds(self.rank, self.world, self.A_loc)
        if not hasattr(self, "_fl_tables"):
            assert self.tick == 0, "a flight starts at tick 0 (or continues a flight)"
            self._fl_tables = torch.zeros((4, self.A_tot, _abi.TRAJ_RECORD_BYTES), dtype=torch.uint8, device="cuda")
            self._fl_next = 0
        assert self.tick == self._fl_next, "fly() continues flights only"
        log_r = torch.zeros((n_ticks, self.A_loc, _abi.TRAJ_RECORD_BYTES), dtype=torch.uint8, device="cuda")
        log_ok = torch.zeros((n_ticks, self.A_loc), dtype=torch.int32, device="cuda")
        if self.exchange.active and os.environ.get("SOGM_FLIGHT_EXCHANGE", "device") == "device":
            # the exchange behind the call (SogmFlight::nccl_comm): ONE call for all ticks — per tick a device-side wait, the
            # in-place all-gather of this rank's rows and the release of the next-but-one tick's overlays are queued on the
            # context's exchange stream; no host step between ticks, every rank passes the same n_ticks
            assert 1 <= n_ticks <= _abi.FLIGHT_MAX_TICKS
            worlds = [c.world(self.tick + i) for i in range(n_ticks)]
            self.planner.flight(worlds, self.tick, self.t0, TICK_PERIOD, REPLAN_START_TIME, self.goals, self.dev["ego_ids"],
                                self.hover, self.own, self._fl_tables, log_r, log_ok, n_total=self.A_tot, agent0=lo,
                                comm=self.exchange.handle)
            self.tick += n_ticks
            self._fl_next = self.tick
            self.all = self._fl_tables[(self.tick - 1) & 3]
            return log_ok, log_r
        done = 0
        while done < n_ticks:
            n = min(2, n_ticks - done)
            worlds = [c.world(self.tick + i) for i in range(n)]
            # (flight_guard: a test that runs several ranks as threads on ONE device serialises their calls with it — two
            #  flights at once would fight for the same compute-unit partitions, each holding what the other's kernels need)
            guard = getattr(self, "flight_guard", None)
            if guard is not None:
                guard.acquire()
            try:
                self.planner.flight(worlds, self.tick, self.t0, TICK_PERIOD, REPLAN_START_TIME, self.goals, self.dev["ego_ids"],
                                    self.hover, self.own, self._fl_tables, log_r[done:done + n], log_ok[done:done + n],
                                    n_total=self.A_tot, agent0=lo)
                if guard is not None:
                    torch.cuda.current_stream().synchronize()
            finally:
                if guard is not None:
                    guard.release()
            for k in range(self.tick, self.tick + n):
                tab = self._fl_tables[k & 3]
                if self.exchange.active:
                    self.exchange.all_gather(tab[lo:hi], tab)     # in place: this rank's rows are where they belong
                    self.exchange.wait()
                else:
                    exchange_records(tab[lo:hi].clone(), tab, self.dist, self.world)
            self.tick += n
            done += n
        self._fl_next = self.tick
        self.all = self._fl_tables[(self.tick - 1) & 3]
        return log_ok, log_r
