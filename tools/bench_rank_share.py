"""One rank's share of BASELINE configs[3] (512 agents over 8 GPUs, 64 per GPU, 200^3 x 20) flown on ONE GPU.

A rank of configs[3] owns a contiguous block of 64 agents of a 512-agent swarm and reads a 512-row table of trajectory
records that the all-gather refreshes once per tick.  Its neighbours on the start circle are the blocks of the ranks
before and after it.  This tool flies the timed rank beside them in one process — each rank a driver.SwarmTick(rank,
world=8) of its own over the ONE 512-agent scene, one rank after the other within a tick, driver.LoopbackHub standing
where the collective stands (every rank reads the others' records of the tick before, as the all-gather leaves them):
  pass 1 (child process): the timed rank and its ring neighbour BEFORE it fly live; that neighbour's rows are logged;
  pass 2 (child process): the timed rank and its ring neighbour AFTER it fly live, the first one's rows replayed; logged;
  pass 3 (this process, timed): the timed rank alone with both neighbours' logged rows replayed tick by tick.
(Passes because one process cannot hold three planners: their streams outnumber the device's hardware queue slots and the
firmware time-slices them — measured: ticks of 15-19 ms, two of five map updates delayed by 9-13 ms, also with two live
planners after two closed ones, ROCm pools the queues of destroyed streams; one or two planners in a fresh process fly
clean.  An artefact of co-simulation, not of configs[3]: profiles/EXPERIMENTS.md.)  bench.py runs passes 1-2
(neighbour_rows) before it touches the GPU itself and pass 3 among its configs blocks.
The replayed neighbours planned beside a timed rank whose own records differed slightly from pass 3's (it had not
met both of them yet): a second-order difference.  The ranks farther away stay "nothing received yet" rows: their agents start >= 96 m of arc
away and cannot reach a timed agent's +-15 m map within the few seconds flown here.

What is timed: the timed rank's step() alone (host-synchronised around it) — its 64 agents' map update from the tick's
sensor frame, overlay of the 512-row table, A*, corridors, QP, deconfliction.  What is NOT in it: the RCCL all-gather
itself (512 x 2064 B = 1 MB per tick) and whatever a real 8-process node adds.  The figure x 8 is a PROJECTION of
configs[3], labelled so; the measured curve is the driver's (bench.py --gpus 8 --agents 64).

run() returns the dict bench.py prints as configs.cfg3_rank_share; the command line prints it as one JSON line."""
import argparse, importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def run(grid="cfg2", agents_per_rank=64, world=8, timed_rank=0, steps=20, warmup=3, device=0, neighbours=2, rows_file=None,
        log=None, live_with=None, out_file=None):
    pop = importlib.import_module("pred-occ-planner_amd")
    driver = importlib.import_module("pred-occ-planner_amd.driver")
    bench = importlib.import_module("bench")     # flow_chain: the per-agent stage stamps of the last dataflow replan
    nxt, prv = (timed_rank + 1) % world, (timed_rank - 1) % world
    spec = pop.config.make_spec(grid)
    scene = pop.scene.make_scene(agents_per_rank * world, (spec.L // 2) * 0.15, seed=0x5069)   # ONE scene: the whole swarm's
    n_ticks = steps + warmup

    def fly(live, replay, timed):
        """ranks `live` fly n_ticks ticks (the timed rank first within a tick: nothing of the other's work is queued in
        front of it); `replay` = {rank: [rows per tick]}; returns the per-tick rows of every live rank + the timings"""
        hub = driver.LoopbackHub(world, agents_per_rank)
        sws = {}
        for r in live:
            # the timed rank as a rank of the real job would run (three grids per agent when HBM has room: the sparse
            # reset under the replan); a neighbour with a single grid — only its records matter
            sws[r] = driver.SwarmTick(grid, agents_per_rank, r, world, device, scene=scene, moving_world=True, prestamp=False,
                                      exchange=hub.exchange(r), grids=None if (r == timed_rank and timed) else 1)
            sws[r].compute.prepare(0, n_ticks)
        tw = sws[timed_rank]
        assert tw.A_tot == agents_per_rank * world and tw.exchange.active and tw.publish
        offs = []
        for _ in range(5):
            h_a = time.perf_counter()
            d_s, h_b = tw.map.device_clock()
            offs.append((h_b - h_a, h_b - d_s))   # host = device + offset (as bench.py's slowest-tick split)
        clock_offset = min(offs)[1]
        lo, hi = driver.shard_bounds(timed_rank, world, agents_per_rank)
        rows = {r: [] for r in live}
        tm, oks, seen, slowest, splits = [], [], [], None, []
        for k in range(n_ticks):
            on = timed and k >= warmup
            for r in live:
                if r == timed_rank and on:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                ok_r = sws[r].step()
                if r == timed_rank:
                    ok = ok_r
                    if on:
                        torch.cuda.synchronize()
                        t2 = time.perf_counter()
            for r in live:
                rows[r].append(sws[r].own.clone())
            hub.commit({r: v[k] for r, v in replay.items()})
            if not on:
                continue
            ms = (t2 - t0) * 1e3
            tm.append(ms)
            d_upd, d_rep = tw.map.tick_clock()
            ch = bench.flow_chain(pop, tw, absolute=True) or {}
            if ch and d_upd > 0 and d_rep > 0:
                first_res, last_fin = ch.pop("first_resident_s"), ch.pop("last_finish_s")
                splits.append([round(v * 1e3, 3) for v in (d_upd + clock_offset - t0, first_res - d_upd, last_fin - first_res,
                                                            d_rep - last_fin, t2 - d_rep - clock_offset)])
            if slowest is None or ms > slowest["tick_ms"]:
                slowest = dict(ch, tick=tw.tick - 1, tick_ms=ms)
            oks.append(int(ok.sum().item()))
            # rows of OTHER ranks that hold a record when the next tick reads the table: the overlay's cross-rank input
            filled = (tw.records_all() != 0).any(dim=1)
            filled[lo:hi] = False
            seen.append(int(filled.sum().item()))
        torch.cuda.synchronize()
        for r in live:
            if sws[r].planner.flow_failures()[1]:
                raise RuntimeError(f"rank {r}: a tick failed on the device")
        grids = tw.overlap_mode if tw.overlap_mode >= 2 else 1
        for r in live:
            sws[r].close()
        torch.cuda.empty_cache()
        return rows, (np.array(tm), oks, seen, slowest, splits, grids)

    def load(path):
        z = np.load(path)
        assert int(z["n_ticks"]) == n_ticks and int(z["world"]) == world and int(z["agents"]) == agents_per_rank, "rows of another run"
        return {int(k[1:]): [torch.from_numpy(x).cuda() for x in z[k]] for k in z.files if k[0] == "r"}

    replay = load(rows_file) if rows_file else {}
    if log is not None:
        # a logging pass (child process): the timed rank + `live_with` live, `rows_file` replayed; rows of `log` written
        rows, _ = fly([timed_rank, live_with], replay, False)
        keep = dict({f"r{r}": np.stack([x.cpu().numpy() for x in v]) for r, v in replay.items()},
                    **{f"r{log}": np.stack([x.cpu().numpy() for x in rows[log]])})
        np.savez(out_file, n_ticks=n_ticks, world=world, agents=agents_per_rank, **keep)
        return {"logged": log, "ticks": n_ticks}
    if rows_file is None and neighbours >= 1 and world > 1:
        import tempfile
        replay = load(neighbour_rows(tempfile.mkdtemp(prefix="sogm_rows_"), grid, agents_per_rank, world, timed_rank, steps, warmup,
                                     neighbours))
    live = [timed_rank]
    _, (tm, oks, seen, slowest, splits, grids) = fly(live, replay, True)
    return {"workload": f"one rank's share of BASELINE configs[3]: rank {timed_rank} of {world}, {agents_per_rank} of "
                        f"{agents_per_rank * world} agents, {spec.L}x{spec.W}x{spec.H}x{spec.T} SOGM, {agents_per_rank * world}-row "
                        f"record table, moving world, lock-step; flown alone on this GPU with the rows of ranks {sorted(replay)} replayed tick by "
                        "tick from passes in which they flew live beside it (LoopbackHub in place of the all-gather); "
                        "the other ranks silent",
            "ms_per_step": float(tm.mean()), "ms_p50": float(np.median(tm)), "ms_max": float(tm.max()), "steps": steps,
            "rank_replans_per_s": agents_per_rank * steps / (tm.sum() * 1e-3),
            "replans_ok_fraction": sum(oks) / float(agents_per_rank * steps),
            "other_ranks_rows_in_table": {"first": seen[0], "last": seen[-1]},
            "sogm_grids_per_agent": grids,
            "projected_cfg3_replans_per_s": agents_per_rank * world * steps / (tm.sum() * 1e-3),
            "projection": f"x{world}: every rank's tick is the same shape and the ranks only meet in one 1 MB all-gather per "
                          "tick, which is NOT in this figure — a projection, not a measurement of 8 GPUs",
            "tick_ms": [round(float(x), 3) for x in tm], "slowest_tick": slowest,
            "split_ms": {"what": "per tick: host launch -> first kernel, map update -> first search resident, chain, last "
                                 "finish -> closing kernel, closing kernel -> sync return (one clock: sogm_device_clock)",
                         "ticks": splits}}


def neighbour_rows(tmpdir, grid="cfg2", agents_per_rank=64, world=8, timed_rank=0, steps=20, warmup=3, neighbours=2):
    """Passes 1-2 in child processes (each exits before the next starts: its hardware queues are gone with it).  Returns the
    .npz with the per-tick rows of the ring neighbours."""
    import subprocess
    nxt, prv = (timed_rank + 1) % world, (timed_rank - 1) % world
    base = [sys.executable, os.path.abspath(__file__), "--grid", str(grid), "--agents", str(agents_per_rank), "--world", str(world),
            "--rank", str(timed_rank), "--steps", str(steps), "--warmup", str(warmup)]
    f1, f2 = os.path.join(tmpdir, "rows1.npz"), os.path.join(tmpdir, "rows2.npz")
    todo = [(prv, None, f1), (nxt, f1, f2)] if neighbours >= 2 and world > 2 else [(nxt, None, f2)]
    for log, rep, out in todo:
        cmd = base + ["--log", str(log), "--out", out] + (["--rows", rep] if rep else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        if r.returncode != 0:
            raise RuntimeError(f"bench_rank_share: logging pass failed: {r.stderr[-1500:]}")
    return f2


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="cfg2")
    ap.add_argument("--agents", type=int, default=64)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--neighbours", type=int, default=2, help="ring neighbours: 2 (default: one live, one replayed), 1, 0")
    ap.add_argument("--rows", default=None, help="replay these logged rows (.npz of an earlier pass)")
    ap.add_argument("--log", type=int, default=None, help="logging pass: fly this rank live beside the timed one, write its rows")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    print(json.dumps(run(a.grid, a.agents, a.world, a.rank, a.steps, a.warmup, neighbours=a.neighbours, rows_file=a.rows,
                         log=a.log, live_with=a.log, out_file=a.out)))
