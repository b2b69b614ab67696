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
    ap.add_argument("--dense-ticks", type=int, default=8, dest="dense_ticks",
                    help="ticks of the block that runs the DENSE clear inside the tick (kernels[dense clear].in_tick; 0 = skip)")
    ap.add_argument("--sustained", type=int, default=300,
                    help="ticks of the sustained-flight block after the timed region (0 = skip)")
    ap.add_argument("--watchdog", type=float, default=240.0,
                    help="N > 1: seconds the sections after the headline may take before rank 0 prints the line without them (0 = off)")
    ap.add_argument("--no-variants", action="store_true", dest="no_variants",
                    help="skip the labelled variants (pre-stamped lock-step, sogm_flight_run) and the cfg1 / cfg4 blocks")
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
    body = pop.scene.received_body_particles()
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
    # single-thread latency per stage (the reference prints the same split, baseline.cpp:286,380,435):
    # a few agents of tick 0, one after the other, nothing else running on the host
    lat = {"sogm_update": [], "astar": [], "corridor": [], "qp": []}
    for a in range(min(A, 5)):
        t1 = time.perf_counter()
        g = orc.update_gt(spec, scene["cloud"], cyl, len(scene["cylinders"]), scene["poses"][a])
        orc.project_neighbours(spec, g, recs, A, a, body, scene["poses"][a], scene["stamps"][a])
        t2 = time.perf_counter()
        pva = np.concatenate([scene["starts"][a], np.zeros(6)])
        w = orc.astar_search(spec, ap, g, scene["poses"][a], pva, scene["goals"][a], 0.02, pp.corridor_tau)
        t3 = time.perf_counter()
        lat["sogm_update"].append(t2 - t1)
        lat["astar"].append(t3 - t2)
        if w["ret"] == 0:
            continue
        cc = orc.corridor_generate(spec, pp, g, scene["poses"][a], float(scene["stamps"][a]), pva,
                                   float(scene["stamps"][a]) + 0.02, w["route"])
        t4 = time.perf_counter()
        lat["corridor"].append(t4 - t3)
        M = cc["npoly"]
        if M <= 0:
            continue
        goal = np.concatenate([cc["goal"], np.zeros(3)])
        orc.qp_solve(pva, goal, [pp.corridor_tau] * M, cc["polys"], cc["nfaces"], pp.max_faces, pp.opt_max_vel,
                     pp.opt_max_acc, qs)
        lat["qp"].append(time.perf_counter() - t4)
        del g
    stages = {k: (float(np.median(v)) * 1e3 if v else None) for k, v in lat.items()}
    stages["what"] = (f"median single-thread latency in ms over {len(lat['sogm_update'])} agents of tick 0, stage by stage "
                      "(SOGM update = fill + marks + overlay of a 640 MB grid; the GPU's per-agent stage times are chain_ms)")
    return {"value": n / dt, "unit": "replans/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
            "stages_ms": stages,
            "sample": f"{n} agent-replans of tick 0 (SOGM update + overlay + A* + corridors + QP) of the same "
                      f"{spec.L}x{spec.W}x{spec.H}x{spec.T} workload, {cores} threads, "
                      f"{sum(oks)}/{n} succeeded, {dt:.1f} s wall"}


def flow_chain(pop, sw, absolute=False):
    """Per-agent stage times of the LAST dataflow replan (device timestamps, sogm_debug_flow_times): how the tick runs
    the stages — chained per agent — and which agent's chain ended it.  Synchronises."""
    import ctypes as C
    import numpy as np
    A = sw.A_loc
    ts = np.zeros((A, 8), np.int64)
    pop.lib().sogm_debug_flow_times(sw.planner._p, ts.ctypes.data_as(C.c_void_p))
    if not ts[:, 6].any():
        return None  # (the grouped-stream replan keeps no per-agent stamps)
    t0 = ts[:, 7].min() if ts[:, 7].min() > 0 else ts[:, 0].min()
    ms = (ts[:, :7] - t0) / 1e5  # 100 MHz ticks -> ms
    d = lambda a, b: ms[:, b] - ms[:, a]
    crit = int(np.argmax(ms[:, 6]))
    # the slowest QP of that replan, from the solve's own clocks (QpWorkspace::dbg: 100 MHz wall clock + shader clock)
    st = np.zeros((A, 16), np.int64)
    pop.lib().sogm_debug_qp_stats(sw.planner._p, st.ctypes.data_as(C.c_void_p))
    q = int(np.argmax(d(4, 5)))
    its = max(int(st[q, 6]), 1)
    slow_qp = {"agent": q, "ms": float(st[q, 0]) / 1e5, "iterations": int(st[q, 6]), "refactorisations": int(st[q, 3]),
               "checks": int(st[q, 5]), "setup_ms": float(st[q, 1]) / 1e5, "refactor_ms": float(st[q, 2]) / 1e5,
               "checks_ms": float(st[q, 4]) / 1e5,
               "us_per_iteration": float(st[q, 0] - st[q, 1] - st[q, 2] - st[q, 4]) / 100.0 / its,
               "register_resident": bool(st[q, 7] & 1), "rows_in_lds": bool(st[q, 7] & 2),
               "shader_clock_ghz": float(st[q, 11]) / max(float(st[q, 0]) * 10.0, 1.0)}
    extra = {"first_resident_s": float(t0) * 1e-8, "last_finish_s": float(ts[:, 6].max()) * 1e-8} if absolute else {}
    return {**extra, "slowest_qp": slow_qp,"astar_mean": float(d(0, 1).mean()), "astar_max": float(d(0, 1).max()),
            "corridor_mean": float(d(2, 3).mean()), "corridor_max": float(d(2, 3).max()),
            "qp_mean": float(d(4, 5).mean()), "qp_max": float(d(4, 5).max()),
            "finish_mean": float(d(5, 6).mean()), "chain_mean": float(ms[:, 6].mean()), "chain_end": float(ms[:, 6].max()),
            "critical_agent": crit,
            "critical_chain": {"astar": float(d(0, 1)[crit]), "corridor": float(ms[crit, 3] - ms[crit, 1]),
                               "qp": float(ms[crit, 5] - ms[crit, 3]), "finish": float(d(5, 6)[crit])},
            "what": "device timestamps of the dataflow replan's persistent kernels, first search resident = 0"}


def _r(x, nd=2):
    return None if x is None else round(float(x), nd)


class PostHeadlineWatchdog:
    """N > 1 only.  Once the headline of the line is measured, nothing after it may cost the line: if the sections behind it
    (dense ticks, sustained ticks, the multi-rank flight — none of which has run on more than one REAL GPU before the driver's
    first scaling run) do not finish within `seconds`, rank 0 prints the line with what it has, `watchdog` naming the section
    that hung, and every rank leaves with os._exit (a rank stuck inside a collective cannot be joined).  The other ranks
    leave a little later than rank 0 so that its line is out before their end of the communicator goes away."""

    def __init__(self, out, rank, seconds):
        import threading
        self.out, self.rank, self.section, self.done = out, rank, "start", False
        self.t = threading.Timer(seconds + (0 if rank == 0 else 15), self.fire)
        self.t.daemon = True
        self.t.start()

    def fire(self):
        if self.done:
            return
        if self.rank == 0:
            self.out["watchdog"] = f"the sections after the headline did not finish in time; hung in: {self.section}"
            self.out.setdefault("variants", {})
            if self.section == "flight":
                self.out["variants"]["flight"] = {"error": ["watchdog: the multi-rank flight did not return"], "flights_failed": 1}
            self.out.setdefault("cpu_baseline", None)
            try:
                print(json.dumps(compact_line(self.out)), flush=True)
            finally:
                os._exit(0)
        os._exit(0)

    def cancel(self):
        self.done = True
        self.t.cancel()


def compact_line(out):
    """The ONE JSON line bench.py prints.  Everything measured goes to a file (profiles/bench_last_detail.json here, copied to
    gpurun_out/ when that exists); the line keeps the contract's keys, the roofline and cpu_baseline objects without their
    prose, and ENDS with `summary` (< 1200 characters): a driver that keeps only the last 2000 characters of the line still
    holds every number DESIGN.md section 6 quotes."""
    detail_paths = []
    for d in ("profiles", "gpurun_out"):
        dd = os.path.join(ROOT, d)
        if os.path.isdir(dd):
            try:
                with open(os.path.join(dd, "bench_last_detail.json"), "w") as f:
                    json.dump(out, f, indent=1)
                detail_paths.append(os.path.join(d, "bench_last_detail.json"))
            except OSError:
                pass
    g = lambda o, *ks: (g(o.get(ks[0]), *ks[1:]) if len(ks) > 1 else o.get(ks[0])) if isinstance(o, dict) else None
    ro, cb, cfg = out.get("roofline") or {}, out.get("cpu_baseline"), dict(out.get("config") or {})
    for k in ("world", "tick_overlap"):  # prose: in the detail file
        cfg.pop(k, None)
    line = {k: out[k] for k in ("metric", "value", "value_ok", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in out}
    line["config"] = cfg
    if "multi_gpu" in out:
        line["multi_gpu"] = out["multi_gpu"]
    if "watchdog" in out:
        line["watchdog"] = out["watchdog"]
    line["detail_file"] = detail_paths
    line["roofline"] = {k: ro.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_frac",
                                               "achieved_distinct", "frac_distinct", "bytes_per_launch", "avg_launch_ms",
                                               "launches_timed", "bytes_counted", "timed_where") if k in ro}
    line["cpu_baseline"] = ({k: cb.get(k) for k in ("value", "unit", "cores", "host_cores", "kind", "sample")} if cb else None)
    v, cf, su, ch = out.get("variants") or {}, out.get("configs") or {}, out.get("sustained") or {}, out.get("chain_ms") or {}
    fl, ks = v.get("flight") or {}, {k.get("kernel", "")[:8]: k for k in ro.get("kernels", []) if isinstance(k, dict)}
    stamp = next((k for n, k in ks.items() if n.startswith("k_cull")), {})
    line["summary"] = {
        "headline": {"v": _r(out.get("value"), 0), "ms": _r(out.get("ms_per_step")), "ok": _r(g(out, "config", "replans_ok_fraction"), 4)},
        "lockstep_sustained": {"v": _r(su.get("value"), 0), "p50": _r(su.get("tick_ms_p50")), "p99": _r(su.get("tick_ms_p99")),
                               "max": _r(su.get("tick_ms_max")), "ok": _r(su.get("replans_ok_fraction"), 4),
                               "ticks": su.get("ticks")} if su else None,
        "prestamped": {"v": _r(g(v, "prestamped_lockstep", "value"), 0), "ms": _r(g(v, "prestamped_lockstep", "ms_per_step"))},
        "flight": {"v": _r(fl.get("value"), 0), "ms": _r(fl.get("ms_per_step")), "ok": _r(fl.get("replans_ok_fraction"), 4),
                   "sust_v": _r(g(fl, "sustained", "value"), 0), "sust_ok": _r(g(fl, "sustained", "replans_ok_fraction"), 7),
                   "sust_worst_ms": _r(g(fl, "sustained", "ms_per_tick_worst_flight")),
                   "sust_best_ms": _r(g(fl, "sustained", "ms_per_tick_best_flight")),
                   "map_gate_ms": _r((g(fl, "per_agent_tick_ms", "map") or 0) + (g(fl, "per_agent_tick_ms", "gate_wait") or 0))
                   if fl.get("per_agent_tick_ms") else None,
                   "flights": fl.get("flights"), "flights_failed": fl.get("flights_failed"),
                   **({"ranks": fl.get("ranks")} if fl.get("ranks") else {}),
                   **({"err": json.dumps(fl["error"])[:100]} if fl.get("error") else {})} if fl else None,
        "chain_ms": {"astar": _r(ch.get("astar_mean")), "corr": _r(ch.get("corridor_mean")), "corr_max": _r(ch.get("corridor_max")),
                     "qp": _r(ch.get("qp_mean")), "qp_max": _r(ch.get("qp_max")), "mean": _r(ch.get("chain_mean")),
                     "end": _r(ch.get("chain_end")), "qp_us_it": _r(g(ch, "slowest_qp", "us_per_iteration"), 3)} if ch else None,
        "reset": {"ms": _r(ro.get("avg_launch_ms"), 3), "frac": _r(ro.get("frac"), 3), "traffic_frac": _r(ro.get("traffic_frac"), 3),
                  "frac_distinct": _r(ro.get("frac_distinct"), 3)},
        "stamp_ms": _r(stamp.get("launch_ms"), 3),
        "cfg4": {"v": _r(g(cf, "cfg4", "replans_per_s"), 0), "flight_v": _r(g(cf, "cfg4", "flight", "replans_per_s"), 0),
                 "flights_failed": g(cf, "cfg4", "flight", "flights_failed"), "err": (g(cf, "cfg4", "error") or "")[:60] or None},
        "cfg1_ms_frame": _r(g(cf, "cfg1", "ms_per_frame")),
        "cfg3_share": {"ms": _r(g(cf, "cfg3_rank_share", "ms_per_step")),
                       "proj_v": _r(g(cf, "cfg3_rank_share", "projected_cfg3_replans_per_s"), 0)},
        "cpu": {"v": _r(cb.get("value"), 1), "cores": cb.get("cores")} if cb else None,
    }
    return line


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
    # configs[3]'s rank share (tools/bench_rank_share.py): the ring neighbours of the rank this GPU will play fly in child
    # processes NOW, before this process holds a hardware queue (they exit before the first tick here); their per-tick rows
    # are replayed in the configs block at the end
    share_rows, share_err = None, None
    if world == 1 and not launched and not args.no_variants and args.grid == "cfg2" and args.agents is None:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        rank_share = importlib.import_module("bench_rank_share")
        try:
            import tempfile
            share_rows = rank_share.neighbour_rows(tempfile.mkdtemp(prefix="sogm_rows_"))
        except Exception as e:  # noqa: BLE001  (the headline does not depend on it: say so in the block instead)
            share_err = str(e)[-400:]
    A_loc = args.agents if args.agents is not None else pop.config.AGENTS[args.grid]
    # The world moves: every tick's update takes that tick's sensor frame (scene.WorldTimeline: cylinders advanced by v dt,
    # their cloud points with them) and crops it on the device around the agent's current map centre.  All frames of the
    # run are generated on the host and uploaded BEFORE the first timed tick (inputs resident in HBM, as the contract
    # says; 7.5 MB per frame at 128 agents) — SOGM_WORLD=static flies the frozen world of the earlier rounds.
    moving = os.environ.get("SOGM_WORLD", "moving") != "static"
    sw = driver.SwarmTick(args.grid, A_loc, rank, world, local, dist=dist, deconflict=not args.no_deconflict,
                          double_buffer={"0": False, "1": True}.get(os.environ.get("SOGM_DOUBLE_BUFFER")),
                          grids=int(os.environ["SOGM_GRIDS"]) if os.environ.get("SOGM_GRIDS") else None,
                          moving_world=moving)
    spec = sw.spec
    n_frames = args.warmup + args.steps + (3 + args.dense_ticks if args.dense_ticks > 0 else 0) + \
        (3 + args.sustained if args.sustained > 0 else 0) + (3 + args.steps) + 8
    t_up = time.perf_counter()
    sw.compute.prepare(0, n_frames)
    world_upload_s = time.perf_counter() - t_up

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
    sw.map.set_profiling(slots=(0, 6, pop._abi.PROF_EXCHANGE))
    sw.map.map_traffic(reset=True)
    t0 = time.perf_counter()
    oks = []
    for _ in range(args.steps):
        oks.append(sw.step())  # a copy of this tick's ok flags (device tensor, no host sync)
    barrier()
    dt = time.perf_counter() - t0
    dt_local = dt
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_ok = int(torch.stack(oks).sum().item())  # over ALL timed ticks
    flow_code, flow_failed = sw.planner.flow_failures()  # after the barrier: every timed tick has completed
    if dist is not None:  # on every rank: one failed rank fails the line
        t = torch.tensor([flow_failed], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        flow_failed = int(t.item())
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
    # The rated kernel, launch by launch, INSIDE the timed region: the library records a HIP event pair around every
    # reset (and every all-gather) on the stream it is launched on and keeps one pair per launch, so nothing
    # synchronises between ticks (sogm_profile_read_all).
    import numpy as np
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    per_slot = {k: np.array(sw.map.profile_read_all(k)) for k in range(pop._abi.PROF_N)}
    moved = sw.map.map_traffic(reset=True)      # device-side counts of the timed region: what the resets and stamps moved
    sparse = sw.map.sparse_reset_state()        # (after it: this call restarts the reset's own counters)
    chain = flow_chain(pop, sw)                 # the last timed tick's per-agent stage times (dataflow replan)
    # distinct 32-byte sectors among the entries of the CURRENT grid's mark log (the last timed tick's map): duplicates are
    # zeroed twice by the reset and merge in the L2, so the honest HBM denominator is 4 B x entries + 32 B x DISTINCT sectors
    log_distinct = None
    if sparse["enabled"]:
        import ctypes as _C
        d3 = (_C.c_uint64 * 6)()
        fn = pop.lib().sogm_debug_log_distinct
        fn.restype, fn.argtypes = _C.c_int, [_C.c_void_p, _C.c_void_p]
        if fn(sw.map.ctx, d3) == 0 and d3[0] > 0:
            log_distinct = {"entries": int(d3[0]), "distinct_sectors": int(d3[1]), "ratio": d3[1] / d3[0],
                            "agents_overflowed": int(d3[2]),
                            "duplicates_within_log_positions": {"8": int(d3[3]), "63": int(d3[4]), "1023": int(d3[5])},
                            "what": "the last timed tick's map: valid entries of its mark logs and the distinct sectors "
                                    "among them (sogm_debug_log_distinct: test-and-set over a throw-away bitmap, untimed)"}
    sw.map.set_profiling(True)  # restart the rings for the stage pass below
    # stage pass on the state of the last tick (map must be live: rebuild it without the pre-clear)
    overlap_mode = sw.overlap_mode
    sw.map.set_overlap_clear(False)
    stamps = torch.full((sw.A_loc,), sw.t0 + sw.tick * driver.TICK_PERIOD, dtype=torch.float64, device="cuda")
    t_start = stamps + driver.REPLAN_START_TIME
    pva, valid = planner.traj_eval(sw.own, t_start)
    pva = torch.where(valid.bool().unsqueeze(1), pva, sw.hover).contiguous()
    poses_ = pva[:, :3].to(torch.float32).contiguous()
    frame = sw.compute.world(sw.tick)
    sw.map.updateWorld(frame, poses_, stamps)
    stamp_ms = float(sw.map.profile_read()[1])  # the stamp as the tick runs it (with the mark log), machine to itself
    stamp_moved = sw.map.map_traffic(reset=True)
    reset_alone = None
    if sparse["enabled"]:
        # the sparse reset with the machine to itself: a second update of the (now log-covered) single grid resets it in
        # the update's own stream — the kernel's wide variant (4 lanes = 64-byte lines, 8 entries per trip)
        sw.map.updateWorld(frame, poses_, stamps)
        r_ms = float(sw.map.profile_read()[0])
        mv = sw.map.map_traffic(reset=True)
        if mv["resets"] == 1 and r_ms > 0:
            b_ = 4.0 * mv["reset_entries"] + mv["reset_bytes_zeroed"]
            reset_alone = {"where": "machine to itself, in the update's own stream (4 lanes = 64-byte lines, 8 entries per trip), "
                                    "stage pass after the timed region",
                           "launch_ms": r_ms, "log_entries": int(mv["reset_entries"]), "bytes_zeroed": int(mv["reset_bytes_zeroed"]),
                           "bytes_per_launch": b_, "achieved": b_ / (r_ms * 1e-3) / 1e9, "frac": b_ / (r_ms * 1e-3) / 1e9 / 8000.0}
        sw.map.set_sparse_reset(False)  # the stand-alone figure below is the DENSE clear's (first ticks, dense writers)
    standalone_clear_ms = []
    for _ in range(3):  # the full-width clear with the machine to itself (first launch may still see the page-table
        sw.map.updateWorld(frame, poses_, stamps)
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
    #                           they run per agent in persistent kernels and cannot be timed one by one: chain_ms)
    grid_bytes = sw.map.grid_bytes()  # V * T * 4 bytes x agents of this rank = SURVEY 8(d)'s bytes per SOGM build
    PEAK = 8000.0  # GB/s, MI355X HBM3E (MI355X_MICROARCH guide)

    def rate(nbytes, ms):
        gbps = nbytes / (ms * 1e-3) / 1e9 if ms and ms > 0 else None
        return gbps, (gbps / PEAK if gbps is not None else None)

    def committed(name):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return json.load(f)
        except (OSError, ValueError):
            return None

    # ---- the dense clear: the kernel that does SURVEY 8(d)'s work (first use of a grid, dense writers, SOGM_SPARSE_RESET=0)
    sa_gbps, sa_frac = rate(grid_bytes, min(standalone_clear_ms))
    pmc_dense = committed("r06_pmc_traffic.json") or committed("r05_pmc_traffic.json")
    k_dense = {"kernel": "k_clear_slabs / k_clear_chunks (dense clear: V*T*4 B per agent-update, SURVEY 8(d))",
               "bytes_per_launch": grid_bytes,
               "standalone": {"where": "full width, machine to itself, stage pass after the timed region",
                              "launch_ms": standalone_clear_ms, "achieved": sa_gbps, "frac": sa_frac},
               "in_tick": None,  # filled below (a block of ticks with the sparse reset off)
               "traffic": pmc_dense["bytes_per_launch"] if pmc_dense and pmc_dense.get("algorithmic_bytes_per_launch") == grid_bytes else None,
               "traffic_source": "committed rocprofv3 --pmc passes (profiles/), not measured in this run"}
    # ---- the stamp (k_cull_cylinders + k_stamp_bits + k_stamp_marks, stage pass: machine to itself)
    marks, s_entries = int(stamp_moved["stamp_marks"]), int(stamp_moved["stamp_entries"])
    stamp_alg = 4 * marks + 4 * s_entries
    st_gbps, st_frac = rate(stamp_alg, stamp_ms)
    pmc_stamp = committed("r06_pmc_stamp.json") or committed("r05_pmc_stamp.json")
    k_stamp = {"kernel": "k_cull_cylinders + k_stamp_bits + k_stamp_marks (the stamp; stage pass after the timed region, "
                         "machine to itself — inside the headline tick the same kernels run at its start, on the critical path; in the "
                         "pre-stamped variant as k_prestamp_flow under the replan)",
               "launch_ms": stamp_ms, "marks": marks, "log_entries": s_entries,
               "algorithmic_bytes": stamp_alg, "achieved": st_gbps, "frac": st_frac,
               "note": "algorithmic bytes = 4 B per marked cell + 4 B per log entry; a mark costs HBM a 32-byte sector "
                       "unless the cells of its 2 x 2 x 2 tile (rows: its x-neighbours) share it: traffic / algorithmic is the write "
                       "amplification; neither pass is HBM-bound (bits: the count of its memory-side atomic ORs, marks: "
                       "instruction issue — DESIGN.md 3.1)",
               "traffic": (pmc_stamp or {}).get("bytes_per_launch"),
               "write_amplification": ((pmc_stamp["bytes_per_launch"] / pmc_stamp["algorithmic_bytes_per_launch"])
                                       if pmc_stamp and pmc_stamp.get("algorithmic_bytes_per_launch") else None),
               "traffic_source": "profiles/r06_pmc_stamp.json (r05 until it exists; committed rocprofv3 --pmc passes of the same cell order, not measured in this run)"
                                 if pmc_stamp else None}
    if sparse["enabled"] and moved["resets"] > 0:
        # The map is no longer rebuilt by filling V x T cells: the reset zeroes the lines named by the mark log (DESIGN
        # 3.1 "Sparse reset").  Rated on what it MOVED in this run, counted on the device: 4 B read per log entry + the
        # bytes of the stores it issued (repeated and out-of-grid entries store nothing).
        n_res = int(moved["resets"])
        entries = moved["reset_entries"] / n_res
        zeroed = moved["reset_bytes_zeroed"] / n_res
        reset_bytes = 4.0 * entries + zeroed
        achieved, frac = rate(reset_bytes, avg[0])
        pmc = committed("r06_pmc_reset.json") or committed("r05_pmc_reset.json")
        traffic = pmc["bytes_per_entry"] * entries if pmc and "bytes_per_entry" in pmc else None
        # rated on DISTINCT sectors: 4 B per entry read + 32 B per distinct sector written
        distinct_bytes = (4.0 * entries + 32.0 * entries * log_distinct["ratio"]) if log_distinct else None
        achieved_distinct, frac_distinct = rate(distinct_bytes, avg[0]) if distinct_bytes else (None, None)
        k_reset = {"kernel": "k_reset_sectors (sparse reset of the grid the update swapped out; side stream, under the replan's QP stage)",
                   "avg_launch_ms": float(avg[0]), "launches_timed": n_clear, "log_entries_per_launch": entries,
                   "bytes_zeroed_per_launch": zeroed, "bytes_per_launch": reset_bytes, "achieved": achieved, "frac": frac,
                   "traffic": traffic, "standalone": reset_alone, "log_distinct": log_distinct,
                   "bytes_distinct_per_launch": distinct_bytes, "achieved_distinct": achieved_distinct,
                   "frac_distinct": frac_distinct,
                   "counted_over_traffic": (reset_bytes / traffic) if traffic else None,
                   "distinct_over_traffic": (distinct_bytes / traffic) if traffic and distinct_bytes else None,
                   "traffic_frac": (traffic / (avg[0] * 1e-3) / 1e9 / PEAK) if traffic else None,
                   "traffic_source": "profiles/r06_pmc_reset.json (r05 until it exists): rocprofv3 --pmc FETCH_SIZE (doubled, gfx950 streaming-read "
                                     "correction) + WRITE_SIZE per log entry, scaled by this run's entry count" if traffic else None}
        roofline = {"bound": "hbm", "kernel": k_reset["kernel"],
                    "achieved": achieved, "peak": PEAK, "unit": "GB/s", "frac": frac,
                    "traffic": traffic, "traffic_source": k_reset["traffic_source"],
                    "traffic_frac": k_reset["traffic_frac"],
                    # the same launch rated on 4 B x entries + 32 B x DISTINCT sectors (the log holds duplicates the stamp's
                    # wave-local lookback cannot see; they are zeroed twice and merge in the L2)
                    "achieved_distinct": achieved_distinct, "frac_distinct": frac_distinct,
                    "bytes_distinct_per_launch": distinct_bytes, "log_distinct": log_distinct,
                    "bytes_per_launch": reset_bytes, "log_entries_per_launch": entries,
                    "bytes_zeroed_per_launch": zeroed,
                    "avg_launch_ms": float(avg[0]), "launches_timed": n_clear, "sparse_resets": n_res,
                    "bytes_counted": "on the device in this run (sogm_map_traffic): 4 B x log entries read + 16 B x stores issued",
                    "timed_where": "HIP events on the reset's stream around every reset of the timed region",
                    "note": "no kernel of the tick is HBM-bound: the tick is bound by the per-agent A* -> corridor -> QP "
                            "chain (chain_ms); the reset is the largest streaming kernel left, latency-bound scattered "
                            "32-byte stores; SURVEY 8(d)'s V*T*4 B fill is not executed (dense_equivalent), the kernel "
                            "that does execute it is kernels[dense clear]",
                    "dense_equivalent": {"bytes": grid_bytes, "rate_GBps": grid_bytes / (avg[0] * 1e-3) / 1e9},
                    "standalone": dict(k_dense["standalone"], kernel=k_dense["kernel"], bytes_per_launch=grid_bytes),
                    "kernels": [k_reset, k_stamp, k_dense]}
    else:
        achieved, frac = rate(grid_bytes, avg[0])
        k_dense["in_tick"] = {"avg_launch_ms": float(avg[0]), "launches_timed": n_clear, "achieved": achieved, "frac": frac}
        roofline = {"bound": "hbm",
                    "kernel": "the SOGM clear (voxel update): k_clear_chunks narrow + wide launches in the pooled modes, "
                              "k_clear_slabs otherwise", "achieved": achieved,
                    "peak": PEAK, "unit": "GB/s", "frac": frac, "traffic": k_dense["traffic"],
                    "traffic_source": k_dense["traffic_source"],
                    "bytes_per_launch": grid_bytes, "avg_launch_ms": float(avg[0]),
                    "launches_timed": n_clear,
                    "timed_where": "HIP events on the clear's stream around the clear (both of its launches), every clear of the timed region",
                    "standalone": dict(k_dense["standalone"], kernel=k_dense["kernel"], bytes_per_launch=grid_bytes),
                    "kernels": [k_stamp, k_dense]}
    exchange_kind = "abi" if sw.exchange.active else "torch" if sw.distributed else "local"
    if world > 1 and exchange_kind != "abi":
        raise SystemExit(f"bench.py: {world} ranks but the trajectory exchange is '{exchange_kind}' "
                         f"({sw.exchange.fallback_reason}): the N > 1 line must measure the all-gather behind the C ABI")
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
        "dtype": ("f16" if spec.storage & 1 else "f32") + " occupancy / f64 planning",
        "data": "synthetic",
        "config": {"workload": f"{sw.A_loc} agents/GPU x {world} GPU, {spec.L}x{spec.W}x{spec.H}x{spec.T} SOGM, "
                               f"sim_fkpcp-style moving cylinders, batched ADMM QP (BASELINE {'configs[4]' if args.grid == 'cfg4' else 'configs[2]'} per GPU)",
                   "agents_total": sw.A_tot, "grid": [spec.L, spec.W, spec.H, spec.T],
                   "cloud_points": int(sw.scene["cloud"].shape[0]), "cylinders": int(len(sw.scene["cylinders"])),
                   "world": ("moving: per-tick sensor frames (cylinders advance by v*dt, up to 1 m/s, their cloud points with "
                             "them), cropped on the device around each agent's current map centre through 256-point blocks; "
                             f"{n_frames} frames uploaded before the first tick in {world_upload_s:.1f} s") if moving
                            else "static (SOGM_WORLD=static): one frame for every tick",
                   # ticks between the sensor frame a tick's map is built from and the tick itself: 0 = the reference's
                   # updateMap-from-the-latest-cloud; 1 = built by the previous replan's pre-stamp from ITS tick's frame
                   "map_input_staleness_ticks": sw.map_input_staleness_ticks,
                   "replans_ok_fraction": n_ok / float(sw.A_tot * args.steps),
                   # where the timed replans ended + capacity limits hit (sogm_planner_counters)
                   "outcomes": outcomes,
                   "parallelism": f"agents sharded x{world}, 1 all-gather/tick",
                   "tick_overlap": "lock-step (every agent's tick k ends before any agent's tick k + 1 starts; overlay and "
                                   "isSafeAfterOpt read the neighbours' records of tick k - 1) — variants.flight: per-agent overlap "
                                   "under the staleness rule (sogm_flight_run)",
                   # which all-gather ran: "abi" = sogm_traj_allgather (RCCL behind the C ABI), "local" = one process
                   "exchange": exchange_kind,
                   "exchange_fallback_reason": sw.exchange.fallback_reason,
                   "sogm_grids_per_agent": overlap_mode if overlap_mode >= 2 else 1,
                   "sogm_reset": "sparse (logged 32-byte sectors)" if sparse["enabled"] else "dense clear",
                   "sogm_cell_order": "2x2x2 tiles" if spec.storage & 16 else "x-fastest rows",
                   # where the tick's map update runs: inside the previous replan, agent by agent as their records are
                   # published (sogm_planner_set_prestamp), or at the start of the tick
                   "map_update": ("pre-stamped by the previous replan" if sw.prestamp and sparse["enabled"]
                                  else "at the start of the tick")},
        "replans_per_s_per_agent": sw.A_tot * args.steps / dt / sw.A_tot,
        "stage_ms": {"clear": avg[0], "stamp": avg[1], "splat": avg[2], "astar": avg[3], "corridor": avg[4],
                     "qp": avg[5],
                     "what": "clear: the reset inside the timed region; the rest: ONE grouped launch per stage over all "
                             "agents after the timed region (sogm_astar_search / sogm_corridor_generate / "
                             "sogm_bezier_qp_solve), i.e. each stage's slowest agent — NOT how the tick runs them "
                             "(per agent, chained: chain_ms)"},
        "chain_ms": chain,
        "roofline": roofline,
    }
    if world > 1:  # the first N > 1 run verifies itself: what RCCL says, every rank's clock, the collective's own duration
        info = sw.exchange.info()
        mine = torch.tensor([dt_local / args.steps * 1e3, float(info["ranks"]),
                             float(per_slot[pop._abi.PROF_EXCHANGE].mean()) if len(per_slot[pop._abi.PROF_EXCHANGE]) else -1.0],
                            dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        out["multi_gpu"] = {"rccl_ranks": info["ranks"], "rccl_ranks_per_rank": [int(t[1].item()) for t in allr],
                            "rank_ms_per_step": [float(t[0].item()) for t in allr],
                            "allgather_ms_per_rank": [float(t[2].item()) for t in allr],
                            "allgather_launches_timed": int(len(per_slot[pop._abi.PROF_EXCHANGE])),
                            "allgather_bytes_per_rank": sw.A_loc * pop._abi.TRAJ_RECORD_BYTES,
                            "exchange": exchange_kind}
        if any(int(t[1].item()) != world for t in allr):
            raise SystemExit(f"bench.py: ncclCommCount != {world} on some rank: {out['multi_gpu']}")
    dog = PostHeadlineWatchdog(out, rank, args.watchdog) if world > 1 and args.watchdog > 0 else None
    if dog:
        dog.section = "dense ticks"
    # ---- the dense clear INSIDE the tick (sparse reset off): the fill of SURVEY 8(d) as the tick used to run it
    sw.map.set_overlap_clear(overlap_mode != 0, grids=(overlap_mode if overlap_mode >= 2 else 1))
    if sparse["enabled"] and args.dense_ticks > 0:
        sw.map.set_profiling(False)
        for _ in range(3):  # every grid of the pool once through the dense path
            sw.step()
        barrier()
        sw.map.set_profiling(slots=(0, 6))
        t1 = time.perf_counter()
        for _ in range(args.dense_ticks):
            sw.step()
        barrier()
        dense_tick_ms = (time.perf_counter() - t1) / args.dense_ticks * 1e3
        d0, d6 = np.array(sw.map.profile_read_all(0)), np.array(sw.map.profile_read_all(6))
        ms = float(d0.mean()) + (float(d6.mean()) if len(d6) else 0.0) if len(d0) else None
        g_, f_ = rate(grid_bytes, ms)
        k_dense["in_tick"] = {"avg_launch_ms": ms, "launches_timed": int(len(d0)), "achieved": g_, "frac": f_,
                              "tick_ms": dense_tick_ms,
                              "where": f"{args.dense_ticks} ticks with the sparse reset off (sogm_set_sparse_reset 0), after the timed region"}
        sw.map.set_profiling(False)
    variants = {}
    sparse_on, pool_mode = sparse["enabled"], sw.overlap_mode
    if dog:
        dog.section = "sustained lock-step ticks"
    if args.sustained > 0:
        # sustained flight: the 20-step figure covers the first seconds (agents still far apart); keep flying —
        # the swarm converges on the centre, searches get longer — and time every tick (host-synchronised)
        sw.map.set_profiling(False)
        if sparse["enabled"]:
            sw.map.set_sparse_reset(True)
            for _ in range(3):  # untimed: every grid of the pool is cleared densely once before its log takes over
                sw.step()
        sw.planner.counters(reset=True)
        # the device's wall clock against the host's (sogm_device_clock): the slowest tick is split with it
        barrier()
        offs = []
        for _ in range(5):
            h_a = time.perf_counter()
            d_s, h_b = sw.map.device_clock()
            offs.append((h_b - h_a, h_b - d_s))   # host = device + offset, measured at the call's return
        clock_offset = min(offs)[1]
        clock_err_ms = min(offs)[0] * 1e3
        tick_ms, oks2, marks = [], [], []
        slowest = None
        sw.map.map_traffic(reset=True)
        barrier()
        for _ in range(args.sustained):
            t1 = time.perf_counter()
            oks2.append(sw.step())
            t_l = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            tick_ms.append((t2 - t1) * 1e3)
            if slowest is None or tick_ms[-1] > slowest["tick_ms"]:  # (between two timed ticks: not in either)
                d_upd, d_rep = sw.map.tick_clock()
                ch = flow_chain(pop, sw, absolute=True) or {}
                split = None
                if ch and d_upd > 0 and d_rep > 0:
                    first_res, last_fin = ch.pop("first_resident_s"), ch.pop("last_finish_s")
                    to_h = lambda d: d + clock_offset
                    split = {"host_launch_until_first_kernel": (to_h(d_upd) - t1) * 1e3,
                             "map_update_until_first_search_resident": (first_res - d_upd) * 1e3,
                             "chain_first_search_to_last_finish": (last_fin - first_res) * 1e3,
                             "last_finish_to_closing_kernel": (d_rep - last_fin) * 1e3,
                             "closing_kernel_to_host_sync_return": (t2 - to_h(d_rep)) * 1e3,
                             "host_launch_calls_ms": (t_l - t1) * 1e3,
                             "clock_alignment_error_ms": clock_err_ms}
                    split["sum"] = sum(v for k, v in split.items() if k not in ("host_launch_calls_ms", "clock_alignment_error_ms"))
                slowest = dict(ch, tick=sw.tick - 1, tick_ms=tick_ms[-1], split_ms=split)
            mv = sw.map.map_traffic(reset=True)   # (synchronises; between two timed ticks)
            marks.append(int(mv["stamp_marks"]))
        tm = np.array(tick_ms)
        mk = np.array(marks, np.float64)
        n_ok2 = int(torch.stack(oks2).sum().item())
        out["sustained"] = {"ticks": args.sustained, "flight_seconds": args.sustained * driver.TICK_PERIOD,
                            "first_tick": sw.tick - args.sustained,
                            "tick_ms_mean": float(tm.mean()), "tick_ms_p50": float(np.percentile(tm, 50)),
                            "tick_ms_p99": float(np.percentile(tm, 99)), "tick_ms_max": float(tm.max()),
                            "value": sw.A_tot * args.sustained / (tm.sum() * 1e-3),
                            "value_ok": n_ok2 * world / (tm.sum() * 1e-3),
                            "replans_ok_fraction": n_ok2 / float(sw.A_loc * args.sustained),
                            "outcomes_rank0": sw.planner.counters(reset=False),
                            # the world keeps being stamped: cells marked per tick over the block (per-tick device crops
                            # around the agents' current map centres; the frozen crops of round 4 decayed as agents left them)
                            "stamp_marks_per_tick": {"first_tenth_mean": float(mk[:max(1, len(mk) // 10)].mean()),
                                                     "last_tenth_mean": float(mk[-max(1, len(mk) // 10):].mean()),
                                                     "min": float(mk.min()), "max": float(mk.max()), "mean": float(mk.mean())},
                            # the slowest tick accounted for: host launch -> first kernel -> first search resident -> last
                            # finish -> closing kernel -> host sync return (device stamps mapped onto the host clock)
                            "slowest_tick": slowest,
                            "note": "every tick host-synchronised (no overlap between ticks); rank 0's clock"}
        flow_code, flow_failed = sw.planner.flow_failures()
        if flow_failed:
            raise SystemExit(f"bench.py: {flow_failed} tick(s) of the sustained block timed out on the device (code {flow_code})")
        c2 = sw.planner.counters(reset=True)
        if c2["corridor_capacity"] + c2["pieces_capacity"] + c2["deconflict_capacity"]:
            raise SystemExit(f"bench.py: capacity limits hit during the sustained block ({c2})")
    scene_kept, seed_kept = sw.scene, None
    sw.close()
    sw = None
    if not args.no_variants and world == 1 and moving and sparse_on and pool_mode >= 2:
        # ---- labelled variant: the pre-stamp (round 4's headline path) on the moving world, a fresh swarm flying the SAME
        # ticks as the headline.  The replan of tick k builds tick k + 1's map from frame k, the newest frame that exists while
        # it runs: the update leaves the critical path, the map a tick plans on is one tick staler than the reference's.
        torch.cuda.empty_cache()
        pw = driver.SwarmTick(args.grid, A_loc, 0, 1, local, deconflict=not args.no_deconflict, moving_world=True, prestamp=True,
                              scene=scene_kept)
        if pw.prestamp:
            pw.compute.prepare(0, args.warmup + args.steps + 1)
            for _ in range(args.warmup):
                pw.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            okv = [pw.step() for _ in range(args.steps)]
            torch.cuda.synchronize()
            dtv = time.perf_counter() - t1
            if pw.planner.flow_failures()[1]:
                raise SystemExit("bench.py: a tick of the pre-stamped variant failed on the device")
            variants["prestamped_lockstep"] = {
                "value": pw.A_tot * args.steps / dtv, "unit": "replans/s", "ms_per_step": dtv / args.steps * 1e3,
                "replans_ok_fraction": int(torch.stack(okv).sum().item()) / float(pw.A_loc * args.steps),
                "map_input_staleness_ticks": pw.map_input_staleness_ticks, "neighbour_record_staleness_ticks": 1,
                "what": "lock-step tick whose replan also builds the NEXT tick's map (sogm_planner_set_prestamp) from the frame "
                        "of the tick in flight: the update leaves the critical path, the map is one tick stale; same scene and "
                        "ticks as the headline"}
        pw.close()
        pw = None
    if not args.no_variants and world == 1 and moving:
        # ---- labelled variant: sogm_flight_run — every agent on its own clock under the staleness rule (own record of tick
        # k - 1, neighbours' of tick k - 2), a fresh swarm flying the SAME ticks as the headline (3 warm-up + 20 timed), then
        # the same 300 ticks as the sustained block in flights of 60.  Same maps as the headline (staleness 0).
        torch.cuda.empty_cache()
        fw = driver.SwarmTick(args.grid, A_loc, 0, 1, local, deconflict=True, moving_world=True, prestamp=False, grids=1,
                              scene=scene_kept)
        n_fl = args.warmup + args.steps + (args.sustained if args.sustained > 0 else 0)
        fw.compute.prepare(0, n_fl + 1)
        # EVERY flight is checked when it ends: the device's error word (k_flight_reset zeroes it at the start of the next
        # call), the finished count (every agent-tick of the call) and the planner's cumulative failure word.  A flight that
        # timed out on the device fails the variant: no value is reported for it, only what happened.
        fl_log = {"flights": 0, "flights_failed": 0, "failed_word": fw.planner.flow_failures()[1], "errors": []}

        def fly_checked(n, label):
            t_ = time.perf_counter()
            ok_, _ = fw.fly(n)
            torch.cuda.synchronize()
            secs = time.perf_counter() - t_
            _, h_ = fw.planner.flight_stats()
            code_, failed_ = fw.planner.flow_failures()
            fin_ = int(h_[pop._abi.FLIGHT_HDR_FINISHED])
            fl_log["flights"] += 1
            bad = h_[pop._abi.FLIGHT_HDR_ERR] != 0 or fin_ != fw.A_loc * n or failed_ != fl_log["failed_word"]
            fl_log["failed_word"] = failed_
            if bad:
                fl_log["flights_failed"] += 1
                fl_log["errors"].append({"flight": label, "ticks": n, "device_error": int(h_[pop._abi.FLIGHT_HDR_ERR]),
                                         "agent_ticks_finished": fin_, "agent_ticks": fw.A_loc * n,
                                         "ms_per_tick": secs / n * 1e3})
            return ok_, secs, bad

        fly_checked(args.warmup, "warm-up")
        okf, dtf, bad_timed = fly_checked(args.steps, "timed") if args.steps <= pop._abi.FLIGHT_MAX_TICKS else (None, None, True)
        msf, hdr = fw.planner.flight_stats()
        per = (msf[:, :7].sum(axis=0) / max(msf[:, 7].sum(), 1.0)).tolist()
        fl = {"unit": "replans/s",
              "map_input_staleness_ticks": 0, "neighbour_record_staleness_ticks": 2,
              "tick_overlap": "per agent: tick k starts when the agent's own tick k - 1 is finished and every agent has finished "
                              "tick k - 2; it reads its own record of tick k - 1 and the neighbours' records of tick k - 2",
              "what": "sogm_flight_run: ONE call for all timed ticks, four persistent kernels on four CU-masked streams; "
                      "records bit-identical to the same rule flown lock-step (tests/test_flight_gpu.py)"}
        if not bad_timed and not fl_log["flights_failed"]:
            fl.update({"value": fw.A_tot * args.steps / dtf, "ms_per_step": dtf / args.steps * 1e3,
                       "replans_ok_fraction": int(okf.sum().item()) / float(fw.A_loc * args.steps),
                       "per_agent_tick_ms": dict(zip(pop._abi.FLIGHT_STAT_NAMES[:7], per))})
        if args.sustained > 0 and not fl_log["flights_failed"]:
            # (flights of 60 ticks: every call starts with the whole swarm in step and ends with a drain behind its last
            #  straggler, which a longer flight spreads over more ticks; sogm_flight_run takes up to 64)
            FL_N = min(60, pop._abi.FLIGHT_MAX_TICKS)
            per_flight, n_flight, oks3, left = [], [], [], args.sustained
            while left > 0:
                n = min(FL_N, left)
                ok_, secs, bad = fly_checked(n, f"sustained {len(per_flight)}")
                per_flight.append(secs * 1e3 / n)
                n_flight.append(n)
                oks3.append(ok_)
                left -= n
                if bad:
                    break  # (the records behind an aborted flight are not the flight's)
            pf = np.array(per_flight)
            total_ms = float((pf * np.array(n_flight)).sum())
            sus = {"ticks": args.sustained, "flights_of": FL_N, "first_tick": args.warmup + args.steps,
                   "ms_per_tick_worst_flight": float(pf.max()), "ms_per_tick_best_flight": float(pf.min())}
            if not fl_log["flights_failed"]:
                sus.update({"ms_per_tick_mean": total_ms / args.sustained,
                            "value": fw.A_tot * args.sustained / (total_ms * 1e-3),
                            "replans_ok_fraction": int(torch.cat(oks3).sum().item()) / float(fw.A_loc * args.sustained)})
            fl["sustained"] = sus
        fl["flights"], fl["flights_failed"] = fl_log["flights"], fl_log["flights_failed"]
        if fl_log["flights_failed"]:
            fl["error"] = {"what": "a flight timed out on the device (3 s bounded waits): the variant reports no value",
                           "failed": fl_log["errors"]}
        cf = fw.planner.counters(reset=True)
        if cf["corridor_capacity"] + cf["pieces_capacity"] + cf["deconflict_capacity"]:
            raise SystemExit(f"bench.py: capacity limits hit during the flights ({cf})")
        variants["flight"] = fl
        fw.close()
        torch.cuda.empty_cache()
    if dog:
        dog.section = "flight"
    if not args.no_variants and world > 1 and moving and args.steps <= pop._abi.FLIGHT_MAX_TICKS:
        # ---- labelled variant on N > 1 ranks: the flight with the exchange BEHIND the call (SogmFlight::nccl_comm): every rank
        # flies warm-up + timed ticks in ONE sogm_flight_run each, the per-tick all-gathers of the table versions are queued
        # on the exchange stream behind device-side waits — no host step between ticks.  Verified like the lock-step line:
        # every rank's own clock, error word, finished count, late workgroups; one failed rank fails the variant.  Every rank
        # runs the same sequence of collectives whatever happens to it (a rank that raised still joins the all_gather below).
        err_txt, dtf, okf, hdr = None, 0.0, None, None
        fw = None
        try:
            torch.cuda.empty_cache()
            fw = driver.SwarmTick(args.grid, A_loc, rank, world, local, dist=dist, deconflict=True, moving_world=True,
                                  prestamp=False, grids=1, scene=scene_kept)
            if not fw.exchange.active:
                raise RuntimeError(f"the exchange is not the ABI's ({fw.exchange.fallback_reason})")
            fw.compute.prepare(0, args.warmup + args.steps + 1)
            fw.planner.flight_prepare(len(scene_kept["cloud"]))
            barrier()
            fw.fly(args.warmup)
            barrier()
            t1 = time.perf_counter()
            okf, _ = fw.fly(args.steps)
            barrier()
            dtf = time.perf_counter() - t1
            _, hdr = fw.planner.flight_stats()
            if (hdr[pop._abi.FLIGHT_HDR_ERR] != 0 or int(hdr[pop._abi.FLIGHT_HDR_FINISHED]) != fw.A_loc * args.steps
                    or hdr[pop._abi.FLIGHT_HDR_LATE_WGS] != 0 or fw.planner.flow_failures()[1]):
                err_txt = (f"rank {rank}: device error {int(hdr[pop._abi.FLIGHT_HDR_ERR])}, "
                           f"{int(hdr[pop._abi.FLIGHT_HDR_FINISHED])} of {fw.A_loc * args.steps} agent-ticks, "
                           f"{int(hdr[pop._abi.FLIGHT_HDR_LATE_WGS])} late workgroups")
        except Exception as e:  # noqa: BLE001
            err_txt = f"rank {rank}: {str(e)[-300:]}"
        mine = [dtf, float(int(okf.sum().item())) if okf is not None and err_txt is None else -1.0, 0.0 if err_txt is None else 1.0]
        allr = [None] * world
        dist.all_gather_object(allr, (mine, err_txt))
        failed = [e for _, e in allr if e]
        fl = {"unit": "replans/s", "ranks": world, "neighbour_record_staleness_ticks": 2,
              "what": "sogm_flight_run on every rank, ONE call for all timed ticks, the per-tick all-gather of the table versions "
                      "queued behind the call on the exchange stream (SogmFlight::nccl_comm): no host step between ticks",
              "rank_ms_per_step": [m[0] / args.steps * 1e3 for m, _ in allr], "flights_failed": len(failed)}
        if failed:
            fl["error"] = failed
        else:
            dmax = max(m[0] for m, _ in allr)
            fl.update({"value": A_loc * world * args.steps / dmax, "ms_per_step": dmax / args.steps * 1e3,
                       "replans_ok_fraction": sum(m[1] for m, _ in allr) / float(A_loc * world * args.steps)})
        variants["flight"] = fl
        if fw is not None:
            try:
                fw.close()
            except Exception:  # noqa: BLE001
                pass
    out["variants"] = variants
    if not args.no_variants and world == 1 and args.grid == "cfg2":
        # ---- the other BASELINE configurations that fit one GPU, driver-timed in the same run
        cfgs = {}
        try:
            torch.cuda.empty_cache()
            s4 = driver.SwarmTick("cfg4", pop.config.AGENTS["cfg4"], 0, 1, local, moving_world=True)
            s4.compute.prepare(0, 16)
            for _ in range(3):
                s4.step()
            torch.cuda.synchronize()
            s4.map.set_profiling(slots=(0, 6))
            s4.map.map_traffic(reset=True)
            t1 = time.perf_counter()
            ok4 = [s4.step() for _ in range(10)]
            torch.cuda.synchronize()
            dt4 = time.perf_counter() - t1
            r_ms = np.array(s4.map.profile_read_all(0))
            mv4 = s4.map.map_traffic(reset=True)
            b4 = (4.0 * mv4["reset_entries"] + mv4["reset_bytes_zeroed"]) / max(int(mv4["resets"]), 1)
            cfgs["cfg4"] = {"workload": f"BASELINE configs[4]: {s4.A_loc} agents, {s4.spec.L}x{s4.spec.W}x{s4.spec.H}x{s4.spec.T} SOGM, "
                                        "fp16 occupancy cells (a single grid per agent: 207 GB), moving world, lock-step",
                            "replans_per_s": s4.A_tot * 10 / dt4, "ms_per_step": dt4 / 10 * 1e3, "steps": 10,
                            "replans_ok_fraction": int(torch.stack(ok4).sum().item()) / float(s4.A_loc * 10),
                            "sogm_grids_per_agent": s4.overlap_mode if s4.overlap_mode >= 2 else 1,
                            "roofline": {"bound": "hbm", "kernel": "k_reset_sectors (fp16 sectors of 16 cells)",
                                         "bytes_per_launch": b4, "avg_launch_ms": float(r_ms.mean()) if len(r_ms) else None,
                                         "achieved": (b4 / (r_ms.mean() * 1e-3) / 1e9) if len(r_ms) else None, "peak": PEAK,
                                         "unit": "GB/s", "frac": (b4 / (r_ms.mean() * 1e-3) / 1e9 / PEAK) if len(r_ms) else None,
                                         "launches_timed": int(len(r_ms))}}
            if s4.planner.flow_failures()[1]:
                raise SystemExit("bench.py: a cfg4 tick failed on the device")
            s4.close()
            torch.cuda.empty_cache()
            # ... and as a flight (one grid per agent is all a flight needs: configs[4] cannot hold a second)
            f4 = driver.SwarmTick("cfg4", pop.config.AGENTS["cfg4"], 0, 1, local, moving_world=True, prestamp=False, grids=1,
                                  tuning={"flight_qp_units": 3, "flight_search_units": 1, "flight_map_units": 6})
            f4.compute.prepare(0, 16)
            f4.fly(3)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            okf4, _ = f4.fly(10)
            torch.cuda.synchronize()
            dtf4 = time.perf_counter() - t1
            _, h4 = f4.planner.flight_stats()
            if (h4[pop._abi.FLIGHT_HDR_ERR] != 0 or int(h4[pop._abi.FLIGHT_HDR_FINISHED]) != f4.A_loc * 10
                    or f4.planner.flow_failures()[1]):
                cfgs["cfg4"]["flight"] = {"error": f"the flight timed out on the device (code {int(h4[pop._abi.FLIGHT_HDR_ERR])}, "
                                                   f"{int(h4[pop._abi.FLIGHT_HDR_FINISHED])} of {f4.A_loc * 10} agent-ticks finished)",
                                          "flights_failed": 1}
            else:
                cfgs["cfg4"]["flight"] = {"replans_per_s": f4.A_tot * 10 / dtf4, "ms_per_step": dtf4 / 10 * 1e3,
                                          "replans_ok_fraction": int(okf4.sum().item()) / float(f4.A_loc * 10), "flights_failed": 0,
                                          "what": "sogm_flight_run, same scene and ticks (map kernel on 96 CUs, QP 48, search 16)"}
            f4.close()
            torch.cuda.empty_cache()
        except pop.SogmError as e:  # (e.g. HBM already held by another process: say so, do not invent a figure)
            cfgs["cfg4"] = {"error": str(e)[:300]}
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        bench_dsp = importlib.import_module("bench_dsp")
        cfgs["cfg1"] = bench_dsp.run("cfg1", None, 30, 5)
        if share_rows is not None:
            try:
                cfgs["cfg3_rank_share"] = rank_share.run(rows_file=share_rows)
            except Exception as e:  # noqa: BLE001
                cfgs["cfg3_rank_share"] = {"error": str(e)[-400:]}
        elif share_err is not None:
            cfgs["cfg3_rank_share"] = {"error": share_err}
        out["configs"] = cfgs
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pop, spec, scene_kept, args.cpu_agents)
    else:
        out["cpu_baseline"] = None
    if dog:
        dog.cancel()
    if rank == 0:
        print(json.dumps(compact_line(out)))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
