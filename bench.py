#!/usr/bin/env python
"""bench.py — replans/sec of the SOGM replan hot path on MI355X (BASELINE.json metric).

A "step" is one replan tick over the rank's batch of agents: SOGM update (clear + cloud/GT stamps)
+ neighbour overlay + hybrid A* + corridors + Bezier QP, plus the trajectory all-gather (N > 1).
N = 1 workload: BASELINE.json configs[2] — 128 agents on one MI355X, 200^3 x 20 SOGM (the
configuration the metric is quoted on).  N > 1: the same 128 agents per GPU (weak scaling), agents
sharded over ranks, one RCCL all-gather of trajectory records per tick; `--agents 64 --gpus 8` is
BASELINE configs[3] (512 agents over 8 GPUs).

`python bench.py --gpus N` with N > 1 and no RANK in the environment starts the N ranks itself
(re-executes under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`);
under a launcher it checks that WORLD_SIZE == N.  It refuses to run with fewer visible GPUs than ranks.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` and `cpu_baseline`.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# before anything initialises HIP (torch.cuda.device_count() below does): the replan's streams want a hardware queue
# each — ROCm's default is four; the package sets the same default when it is imported first
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", default="cfg2")
    ap.add_argument("--agents", type=int, default=None, help="agents per GPU (default: config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-deconflict", action="store_true", help="skip isSafeAfterOpt at the end of replan()")
    ap.add_argument("--cpu-agents", type=float, default=12.0, dest="cpu_agents",
                    help="cpu_baseline sample: seconds of wall time to spend on the CPU oracle")
    ap.add_argument("--sustained", type=int, default=300,
                    help="ticks of the sustained-flight block after the timed region (0 = skip)")
    return ap.parse_args()


def launch_plan(gpus, env, device_count, argv, port=None):
    """How `bench.py --gpus N` gets its N ranks.  Returns None (run in this process) or the command line to
    re-execute; raises SystemExit with a message when the request cannot be honoured — a bench that silently ran
    one rank while claiming N would print a wrong n_gpus (missing #1).
      * launched by torch.distributed.run / the driver (RANK in env): WORLD_SIZE must equal --gpus;
      * not launched, N == 1: this process is the only rank;
      * not launched, N > 1: re-execute under torch.distributed.run with N local ranks (127.0.0.1 rendezvous)."""
    if gpus < 1:
        raise SystemExit(f"bench.py: --gpus {gpus}: need at least one GPU")
    if "RANK" in env:
        world, local = int(env.get("WORLD_SIZE", "1")), int(env.get("LOCAL_RANK", "0"))
        if world != gpus:
            raise SystemExit(f"bench.py: --gpus {gpus} but launched with WORLD_SIZE={world}: one rank per GPU")
        if device_count <= local and device_count != 1:  # (a launcher may also show each rank its own GPU only)
            raise SystemExit(f"bench.py: local rank {local} but {device_count} GPU(s) visible")
        return None
    if device_count < gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} but only {device_count} GPU(s) visible: refusing to print an "
                         f"n_gpus = {gpus} line from fewer devices")
    if gpus == 1:
        return None
    if port is None:
        import socket
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port)] + list(argv)


def cpu_baseline(pop, spec, scene, n_agents_sample):
    """The CPU oracle ("port") timed on the host cores on a bounded sample of the same workload:
    SOGM update + overlay + full replan for `n_agents_sample` agents of tick 0 (one thread each)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    orc = importlib.import_module("oracle.binding")
    orc.lib()
    ap, pp, qs = pop.config.make_astar_params(), pop.config.make_planner_params(True), pop.config.make_qp_settings()
    cyl = pop.scene.cylinders_to_struct(scene["cylinders"])
    recs = pop.scene.straight_records(scene)
    body = pop.scene.body_particles()
    A = scene["n_agents"]
    # every host core, one agent-replan per thread (SURVEY §8 d); each in-flight agent holds its own SOGM
    # (V*T*4 B = 640 MB at 200^3 x 20) — bounded by the host's RAM, not by a constant
    cores = max(1, min(os.cpu_count() or 1, A))
    try:
        avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
        per_agent = 2.2 * spec.L * spec.W * spec.H * spec.T * 4
        cores = max(1, min(cores, int(0.6 * avail / per_agent)))
    except (ValueError, OSError):
        pass
    budget_s = float(n_agents_sample)

    def one(a):
        g = orc.update_gt(spec, scene["cloud"], cyl, len(scene["cylinders"]), scene["poses"][a])
        orc.project_neighbours(spec, g, recs, A, a, body, scene["poses"][a], scene["stamps"][a])
        pva = np.concatenate([scene["starts"][a], np.zeros(6)])
        ok, rec, stage = orc.replan(spec, ap, pp, qs, g, scene["poses"][a], scene["stamps"][a], pva,
                                    scene["goals"][a], scene["stamps"][a] + 0.02, a)
        if ok:  # isSafeAfterOpt, the last step of replan() (baseline_fake.cpp:453-460)
            ok = orc.safe_after_opt(np.asarray(rec.cpts[:15 * rec.n_pieces]), rec.n_pieces, recs, A, a,
                                    float(scene["stamps"][a]))
        return ok

    # batches of `cores` agents (one thread each) of tick 0 until ~budget_s seconds of wall time
    t0 = time.time()
    oks, nxt = [], 0
    with ThreadPoolExecutor(cores) as ex:
        while time.time() - t0 < budget_s and nxt < 4 * A:
            oks += list(ex.map(one, [(nxt + i) % A for i in range(cores)]))
            nxt += cores
    dt = time.time() - t0
    n = len(oks)
    return {"value": n / dt, "unit": "replans/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"{n} agent-replans of tick 0 (SOGM update + overlay + A* + corridors + QP) of the same "
                      f"{spec.L}x{spec.W}x{spec.H}x{spec.T} workload, {cores} threads, "
                      f"{sum(oks)}/{n} succeeded, {dt:.1f} s wall"}


def main():
    args = parse()
    import torch
    cmd = launch_plan(args.gpus, os.environ, torch.cuda.device_count(), sys.argv)
    if cmd is not None:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.stdout.flush()
        os.execvpe(cmd[0], cmd, env)
    pop = importlib.import_module("pred-occ-planner_amd")
    driver = importlib.import_module("pred-occ-planner_amd.driver")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.device_count() == 1:
        local = 0  # the launcher shows each rank its own GPU only
    dist = None
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # under torch.distributed.run
    if world > 1 or launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    A_loc = args.agents if args.agents is not None else pop.config.AGENTS[args.grid]
    sw = driver.SwarmTick(args.grid, A_loc, rank, world, local, dist=dist, deconflict=not args.no_deconflict,
                          double_buffer={"0": False, "1": True}.get(os.environ.get("SOGM_DOUBLE_BUFFER")),
                          grids=int(os.environ["SOGM_GRIDS"]) if os.environ.get("SOGM_GRIDS") else None)
    spec = sw.spec

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        sw.step()
    barrier()
    sw.planner.counters(reset=True)
    sw.map.sparse_reset_state()  # restart the reset statistics: the timed region's alone are reported
    # only the rated kernel (slot 0, and its two-part form's slot 6) is timed inside the timed region: each timed launch
    # costs two event records on its stream, and the stamp / overlay / planner launches are on the tick's critical path
    sw.map.set_profiling(slots=(0, 6))
    t0 = time.perf_counter()
    oks = []
    for _ in range(args.steps):
        oks.append(sw.step())  # a copy of this tick's ok flags (device tensor, no host sync)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_ok = int(torch.stack(oks).sum().item())  # over ALL timed ticks
    flow_code, flow_failed = sw.planner.flow_failures()  # after the barrier: every timed tick has completed
    if flow_failed:
        raise SystemExit(f"bench.py: {flow_failed} tick(s) of the dataflow replan timed out on the device "
                         f"(code {flow_code}): the timed region is invalid")
    outcomes = sw.planner.counters(reset=True)
    cap_hits = outcomes["corridor_capacity"] + outcomes["pieces_capacity"] + outcomes["deconflict_capacity"]
    if cap_hits:  # a replan cut short by a buffer limit is an outcome the reference (growing vectors) cannot have
        raise SystemExit(f"bench.py: {cap_hits} replans of the timed region hit a capacity limit ({outcomes}): "
                         "the timed region is invalid")
    if dist is not None:
        t = torch.tensor([n_ok] + [outcomes[k] for k in sorted(outcomes)], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        n_ok = int(t[0].item())
        outcomes = dict(zip(sorted(outcomes), [int(v) for v in t[1:].tolist()]))
    # The roofline kernel, launch by launch, INSIDE the timed region: the library records a HIP event pair around
    # every k_clear_slabs launch on the stream it is launched on (the side stream in the pipelined modes) and keeps
    # one pair per launch, so nothing synchronises between ticks (sogm_profile_read_all).
    import numpy as np
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    per_slot = {k: np.array(sw.map.profile_read_all(k)) for k in range(pop._abi.PROF_N)}
    # sparse reset (the default): the log of the grid the last tick built = what the next reset will read and zero
    sparse = sw.map.sparse_reset_state()
    sw.map.set_profiling(True)  # restart the rings for the stage pass below
    # stage pass on the state of the last tick (map must be live: rebuild it without the pre-clear)
    overlap_mode = sw.overlap_mode
    sw.map.set_overlap_clear(False)
    stamps = torch.full((sw.A_loc,), sw.t0 + sw.tick * driver.TICK_PERIOD, dtype=torch.float64, device="cuda")
    t_start = stamps + driver.REPLAN_START_TIME
    pva, valid = planner.traj_eval(sw.own, t_start)
    pva = torch.where(valid.bool().unsqueeze(1), pva, sw.hover).contiguous()
    poses_ = pva[:, :3].to(torch.float32).contiguous()
    sw.map.updateMap(sw.dev["cloud"], sw.dev["cloud_range"], sw.dev["cylinders"], sw.dev["n_cyl"], poses_, stamps)
    stamp_ms = float(sw.map.profile_read()[1])  # the stamp as the tick runs it (with the mark log)
    if sparse["enabled"]:
        sw.map.set_sparse_reset(False)  # the stand-alone figure below is the DENSE clear's (first ticks, dense writers)
    standalone_clear_ms = []
    for _ in range(3):  # the full-width clear with the machine to itself (first launch may still see the page-table
        sw.map.updateMap(sw.dev["cloud"], sw.dev["cloud_range"], sw.dev["cylinders"], sw.dev["n_cyl"], poses_, stamps)
        standalone_clear_ms.append(float(sw.map.profile_read()[0]))  # work of releasing the second grid)
    sw.map.addOtherAgents(sw.all, sw.A_tot, sw.dev["ego_ids"])
    s_ = sw.planner.search(pva, sw.goals, t_start)
    c_ = sw.planner.generateCorridors(pva, t_start, s_["route"], s_["route_len"])
    q_ = sw.planner.optimize(pva, c_["goal"], c_["polys"], c_["nfaces"], c_["npoly"])
    ms_stage = sw.map.profile_read()
    ms_stage[1] = stamp_ms
    avg = np.array([float(per_slot[k].mean()) if len(per_slot[k]) else -1.0 for k in range(pop._abi.PROF_N)])
    n_clear = int(len(per_slot[0]))
    if avg[6] > 0:
        avg[0] += avg[6]  # the clear of one grid = two launches (narrow head + full-width rest): one clear = both
    avg[1:3] = ms_stage[1:3]  # stamp / overlay: the stage pass's launches (not timed inside the timed region)
    avg[3:6] = ms_stage[3:6]  # planner stages: single-stage entry points after the timed region (inside sogm_replan
    #                           they run concurrently on per-group streams and cannot be timed one by one)
    grid_bytes = sw.map.grid_bytes()  # V * T * 4 bytes x agents of this rank = algorithmic bytes / launch
    # HBM traffic of the roofline kernel: PMC counters cannot be read from inside this process; the figure comes
    # from the committed rocprofv3 --pmc passes of THIS command (FETCH_SIZE and WRITE_SIZE in separate runs,
    # tools/make_profile.sh) and is only reported when it was collected on the same workload (same bytes / launch)
    traffic, traffic_source = None, None
    for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                pmc = json.load(f)
            if pmc["algorithmic_bytes_per_launch"] == grid_bytes:
                traffic, traffic_source = pmc["bytes_per_launch"], f"profiles/{name} (committed rocprofv3 --pmc passes, not measured in this run)"
                break
        except (OSError, KeyError, ValueError):
            pass
    standalone = {"kernel": "k_clear_slabs (the dense clear, full width, machine to itself; stage pass after the timed region)",
                  "launch_ms": standalone_clear_ms,
                  "bytes_per_launch": grid_bytes,
                  "achieved": grid_bytes / (min(standalone_clear_ms) * 1e-3) / 1e9,
                  "frac": grid_bytes / (min(standalone_clear_ms) * 1e-3) / 1e9 / 8000.0}
    if sparse["enabled"] and sparse["resets"] > 0:
        # The map is no longer rebuilt by filling V x T cells: the reset zeroes the 32-byte sectors named by the mark
        # log (DESIGN 3.1 "Sparse reset").  The kernel is rated on the bytes it has to move — 4 B read and the
        # 32-byte sector written per log entry (the PMC traffic says what moved) — and SURVEY 8(d)'s dense figure is given beside it for comparison.
        entries = int(sparse["entries_per_reset"])  # mean over the resets of the timed region
        reset_bytes = entries * 36
        achieved = reset_bytes / (avg[0] * 1e-3) / 1e9
        rt = None
        try:
            with open(os.path.join(ROOT, "profiles", "r03_pmc_reset.json")) as f:
                pmc = json.load(f)
            rt = {"bytes_per_entry": pmc["bytes_per_entry"], "bytes_per_launch_scaled": pmc["bytes_per_entry"] * entries,
                  "source": "profiles/r03_pmc_reset.json (committed rocprofv3 --pmc passes of tools/diag_reset_pmc.py, "
                            "scaled by this run's entry count; not measured in this run)"}
        except (OSError, KeyError, ValueError):
            pass
        roofline = {"bound": "hbm",
                    "kernel": "k_reset_sectors: the sparse reset of the SOGM (zeroes the logged 32-byte sectors of the grid "
                              "the update swapped out; side stream, under the replan)",
                    "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                    "traffic": rt["bytes_per_launch_scaled"] if rt else None,
                    "traffic_source": rt["source"] if rt else None,
                    "bytes_per_launch": reset_bytes, "log_entries_per_launch": entries,
                    "avg_launch_ms": float(avg[0]), "launches_timed": n_clear, "sparse_resets": int(sparse["resets"]),
                    "timed_where": "HIP events on the reset's stream around every reset of the timed region",
                    "note": "no kernel of the tick is HBM-bound any more: the tick is bound by the per-agent A* -> "
                            "corridor -> QP chain (stage_ms); this is the largest streaming kernel left",
                    "dense_equivalent": {"bytes": grid_bytes, "rate_GBps": grid_bytes / (avg[0] * 1e-3) / 1e9,
                                         "note": "SURVEY 8(d)'s V*T*4 B per agent-update divided by this launch: above "
                                                 "the HBM peak because the fill is not executed"},
                    "standalone": standalone}
    else:
        achieved = grid_bytes / (avg[0] * 1e-3) / 1e9
        roofline = {"bound": "hbm",
                    "kernel": "the SOGM clear (voxel update): k_clear_chunks narrow + wide launches in the pooled modes, "
                              "k_clear_slabs otherwise", "achieved": achieved,
                    "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                    "traffic_source": traffic_source,
                    "bytes_per_launch": grid_bytes, "avg_launch_ms": float(avg[0]),
                    "launches_timed": n_clear,
                    "timed_where": "HIP events on the clear's stream around the clear (both of its launches), every clear of the timed region",
                    # avg_launch_ms is the launch as it runs inside the tick (a narrow clear sharing the machine
                    # with the planner kernels); the same kernel at full width with the machine to itself:
                    "standalone": standalone}
    out = {
        "metric": "replans/sec (SOGM update + QP: full replan = SOGM update + A* + corridors + QP + deconfliction), "
                  f"{sw.A_loc}-agent batch per GPU, {spec.L}x{spec.W}x{spec.H}x{spec.T} voxel grid; aggregate over all agents",
        "value": sw.A_tot * args.steps / dt,   # replan cycles executed per second (BASELINE.json's metric)
        "value_ok": n_ok / dt,                  # of which replan() returned true
        "unit": "replans/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("f16" if spec.storage == 1 else "f32") + " occupancy / f64 planning",
        "data": "synthetic",
        "config": {"workload": f"{sw.A_loc} agents/GPU x {world} GPU, {spec.L}x{spec.W}x{spec.H}x{spec.T} SOGM, "
                               f"sim_fkpcp-style moving cylinders, batched ADMM QP (BASELINE {'configs[4]' if args.grid == 'cfg4' else 'configs[2]'} per GPU)",
                   "agents_total": sw.A_tot, "grid": [spec.L, spec.W, spec.H, spec.T],
                   "cloud_points": int(sw.scene["cloud"].shape[0]), "cloud_points_scanned": sw.cloud_points, "cylinders": int(len(sw.scene["cylinders"])),
                   "replans_ok_fraction": n_ok / float(sw.A_tot * args.steps),
                   # where the timed replans ended + capacity limits hit (sogm_planner_counters)
                   "outcomes": outcomes,
                   "parallelism": f"agents sharded x{world}, 1 all-gather/tick",
                   # which all-gather ran: "abi" = sogm_traj_allgather (RCCL behind the C ABI), "torch" =
                   # torch.distributed's (also RCCL; the fallback — reason given), "local" = one process
                   "exchange": ("abi" if sw.exchange.active else "torch" if sw.distributed else "local"),
                   "exchange_fallback_reason": sw.exchange.fallback_reason,
                   "sogm_grids_per_agent": overlap_mode if overlap_mode >= 2 else 1,
                   "sogm_reset": "sparse (logged 32-byte sectors)" if sparse["enabled"] else "dense clear",
                   # where the tick's map update runs: inside the previous replan, agent by agent as their records are
                   # published (sogm_planner_set_prestamp), or at the start of the tick
                   "map_update": ("pre-stamped by the previous replan" if sw.prestamp and sparse["enabled"]
                                  else "at the start of the tick")},
        "replans_per_s_per_agent": sw.A_tot * args.steps / dt / sw.A_tot,
        "stage_ms": {"clear": avg[0], "stamp": avg[1], "splat": avg[2], "astar": avg[3], "corridor": avg[4],
                     "qp": avg[5]},
        "roofline": roofline,
    }
    if args.sustained > 0:
        # sustained flight: the 20-step figure covers the first seconds (agents still far apart); keep flying —
        # the swarm converges on the centre, searches get longer — and time every tick (host-synchronised)
        sw.map.set_overlap_clear(overlap_mode != 0, grids=(overlap_mode if overlap_mode >= 2 else 1))
        sw.map.set_profiling(False)
        if sparse["enabled"]:
            sw.map.set_sparse_reset(True)
            for _ in range(3):  # untimed: every grid of the pool is cleared densely once before its log takes over
                sw.step()
        sw.planner.counters(reset=True)
        tick_ms, oks2 = [], []
        barrier()
        for _ in range(args.sustained):
            t1 = time.perf_counter()
            oks2.append(sw.step())
            torch.cuda.synchronize()
            tick_ms.append((time.perf_counter() - t1) * 1e3)
        tm = np.array(tick_ms)
        n_ok2 = int(torch.stack(oks2).sum().item())
        out["sustained"] = {"ticks": args.sustained, "flight_seconds": args.sustained * driver.TICK_PERIOD,
                            "first_tick": sw.tick - args.sustained,
                            "tick_ms_mean": float(tm.mean()), "tick_ms_p50": float(np.percentile(tm, 50)),
                            "tick_ms_p99": float(np.percentile(tm, 99)), "tick_ms_max": float(tm.max()),
                            "value": sw.A_tot * args.sustained / (tm.sum() * 1e-3),
                            "value_ok": n_ok2 * world / (tm.sum() * 1e-3),
                            "replans_ok_fraction": n_ok2 / float(sw.A_loc * args.sustained),
                            "outcomes_rank0": sw.planner.counters(reset=False),
                            "note": "every tick host-synchronised (no overlap between ticks); rank 0's clock"}
    if args.sustained > 0:
        c2 = sw.planner.counters(reset=True)
        if c2["corridor_capacity"] + c2["pieces_capacity"] + c2["deconflict_capacity"]:
            raise SystemExit(f"bench.py: capacity limits hit during the sustained block ({c2})")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pop, spec, sw.scene, args.cpu_agents)
    else:
        out["cpu_baseline"] = None
    sw.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
