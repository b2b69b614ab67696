#!/usr/bin/env python
"""Soak of sogm_flight_run in the sequence bench.py runs it (the round-5 driver box lost one sustained flight of five to the
3 s device time-out): one process, a lock-step headline swarm (three grids, nine streams) flown and closed, a pre-stamped
swarm flown and closed, then REPS fresh flight swarms, each flying 3 + 20 + 5 x 60 ticks at 128 x 200^3 x 20.  After every
flight: the error word, the finished count, the planner's cumulative failure word; a failed flight dumps the control block
(sogm_debug_flight_dump: header counters, tick_done, parked lists, every agent's tick / stage / segment counters and the stamps
of its current tick) and the soak goes on.

    python tools/soak_flight.py [reps] [lockstep_ticks]       env: GPU_MAX_HW_QUEUES, SOGM_TUNING, SOAK_FLIGHTS=60,60,...
Prints one SOAK json line at the end (flights, failed, ms per tick best / worst, ok fraction of every full repetition)."""
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
_abi = pop._abi
lib = pop.lib()
lib.sogm_debug_flight_dump.restype = C.c_int
lib.sogm_debug_flight_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
lib.sogm_debug_corridor_stats.restype = C.c_int
lib.sogm_debug_corridor_stats.argtypes = [C.c_void_p, C.c_void_p]
HDR_NAMES = ("S_READY", "S_TICKET", "Q_READY", "Q_TICKET", "ERR", "FINISHED", "MW_TAIL", "MW_HEAD", "LW_TAIL", "LW_HEAD",
             "M_READY", "M_TICKET", "MAPS_DONE", "ADMITTED", "PACE_CLOCK", "U_READY", "U_TICKET", "UW_TAIL", "UW_HEAD", "END")
ERR_NAMES = {2: "flow_wait_slot", 12: "fl_wait_item (search / QP ring)", 15: "wq_take (work queue position never published)",
             16: "admission order", 17: "fl_wait_item_end (map head ring)"}


def dump(sw, first_tick, n_ticks):
    A = sw.A_loc
    words = np.zeros((20 + 128 + 4 * A,), np.int32)
    ts = np.zeros((A, 16), np.int64)
    rc = lib.sogm_debug_flight_dump(sw.planner._p, words.ctypes.data_as(C.c_void_p), len(words), ts.ctypes.data_as(C.c_void_p))
    if rc != 0:
        return {"dump_rc": rc}
    hdr = dict(zip(HDR_NAMES, words[:20].tolist()))
    td, pn = words[20:84], words[84:148]
    tick_of, seg_done, stage, urgent = (words[148 + i * A:148 + (i + 1) * A] for i in range(4))
    kmin = int(tick_of.min())
    lag = np.nonzero(tick_of == kmin)[0]
    seg = np.zeros((A, 16, 16), np.int64)   # per (agent, segment slot): phase durations 0-11, the flight's stamps 12-15
    lib.sogm_debug_corridor_stats(sw.planner._p, seg.ctypes.data_as(C.c_void_p))
    t_last = max(ts.max(), seg[:, :, 12:15].max())

    def seg_rows(a):
        rows = []
        for j in range(16):
            r = seg[a, j]
            hw, xcc = int(r[15]) & 0xFFFFFFFF, int(r[15]) >> 32
            rows.append({"seg": j, "N": int(r[0]), "taken": round(float(t_last - r[12]) / 1e5, 2) if r[12] else None,
                         "points_done": round(float(t_last - r[13]) / 1e5, 2) if r[13] else None,
                         "done": round(float(t_last - r[14]) / 1e5, 2) if r[14] else None,
                         "phase_ms": {"setup": r[5] / 1e5, "firi0": r[6] / 1e5, "mvie": r[7] / 1e5, "firi1": r[8] / 1e5,
                                      "lbfgs": r[9] / 1e5, "total": r[10] / 1e5},
                         "lbfgs_it": int(r[3]), "shader_ghz": round(float(r[11]) / max(float(r[9]) * 10.0, 1.0), 2),
                         "xcc": xcc & 0xF, "se": (hw >> 13) & 7, "sh": (hw >> 12) & 1, "cu": (hw >> 8) & 15,
                         "simd": (hw >> 4) & 3, "wave": hw & 15})
        return rows
    out = {"hdr": hdr, "err_wait": ERR_NAMES.get(hdr["ERR"], "?"), "first_tick": first_tick, "n_ticks": n_ticks,
           "tick_done": td[:n_ticks].tolist(), "parked_n": pn[:n_ticks].tolist(),
           "tick_of_hist": {int(k): int((tick_of == k).sum()) for k in np.unique(tick_of)},
           "laggards": [{"agent": int(a), "tick": int(tick_of[a]), "stage": int(stage[a]), "seg_done": int(seg_done[a]),
                         "urgent": int(urgent[a]),
                         # stamps of the agent's current tick relative to the newest stamp anywhere, ms (0 = never written this tick)
                         "ts_ms_before_last": [round(float(t_last - v) / 1e5, 2) if v else None for v in ts[a]],
                         "segments": seg_rows(a)}
                        for a in lag[:8]]}
    return out


def fly_checked(sw, n, log, label):
    t1 = time.perf_counter()
    first = sw.tick
    ok, _ = sw.fly(n)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t1) * 1e3 / n
    _, hdr = sw.planner.flight_stats()
    code, failed = sw.planner.flow_failures()
    fin = int(hdr[_abi.FLIGHT_HDR_FINISHED])
    bad = hdr[_abi.FLIGHT_HDR_ERR] != 0 or fin != sw.A_loc * n or failed != log["failed_word"]
    log["failed_word"] = failed
    log["flights"] += 1
    log["ms"].append(ms)
    if bad:
        log["failed"] += 1
        d = dump(sw, first, n)
        print(f"SOAK-FAIL {label} first_tick {first} n {n} ms/tick {ms:.2f} err {int(hdr[_abi.FLIGHT_HDR_ERR])} finished {fin}/{sw.A_loc * n} "
              f"flow_failures {(code, failed)}\nSOAK-DUMP " + json.dumps(d), flush=True)
    return ok, bad


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    lock = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    flights = [int(x) for x in os.environ.get("SOAK_FLIGHTS", "3,20,60,60,60,60,60").split(",")]
    grid = os.environ.get("GRID", "cfg2")
    A = int(os.environ.get("AGENTS", pop.config.AGENTS[grid]))
    t_all = time.perf_counter()
    scene = None
    if lock > 0:
        # the headline swarm of bench.py: three grids, lock-step, then the pre-stamped one — flown and closed (their
        # streams' hardware queues go back to ROCm's pool, as in the bench)
        sw = driver.SwarmTick(grid, A, moving_world=True)
        sw.compute.prepare(0, lock + 1)
        for _ in range(lock):
            sw.step()
        torch.cuda.synchronize()
        assert not sw.planner.flow_failures()[1], "a lock-step tick failed"
        scene = sw.scene
        sw.close()
        torch.cuda.empty_cache()
        pw = driver.SwarmTick(grid, A, moving_world=True, prestamp=True, scene=scene)
        pw.compute.prepare(0, 24)
        for _ in range(23):
            pw.step()
        torch.cuda.synchronize()
        pw.close()
        torch.cuda.empty_cache()
    log = {"flights": 0, "failed": 0, "ms": [], "failed_word": 0}
    frames, okfrac = None, []
    for r in range(reps):
        fw = driver.SwarmTick(grid, A, moving_world=True, prestamp=False, grids=1, scene=scene)
        scene = fw.scene
        if frames is not None:
            fw.compute._frames = frames   # (sensor frames are context-free device tensors: uploaded once)
        fw.compute.prepare(0, sum(flights) + 1)
        frames = fw.compute._frames
        log["failed_word"] = 0
        oks, bad_any = [], False
        for i, n in enumerate(flights):
            ok, bad = fly_checked(fw, n, log, f"rep {r} flight {i}")
            oks.append(ok)
            bad_any |= bad
            if bad:
                break   # (the records after an aborted flight are not the flight's: start over with a fresh swarm)
        if not bad_any:
            okfrac.append(int(torch.cat(oks).sum().item()) / float(A * sum(flights)))
        fw.close()
        torch.cuda.empty_cache()
        print(f"rep {r}: flights {log['flights']} failed {log['failed']} last ms/tick {log['ms'][-1]:.2f} "
              f"elapsed {time.perf_counter() - t_all:.0f} s", flush=True)
    ms = np.array(log["ms"])
    full = ms[[i for i in range(len(ms))]]
    print("SOAK " + json.dumps({"reps": reps, "flights_per_rep": flights, "flights": log["flights"], "failed": log["failed"],
                                "ms_per_tick_best": float(full.min()), "ms_per_tick_worst": float(full.max()),
                                "ms_per_tick_median": float(np.median(full)),
                                "ok_fraction_distinct": sorted(set(round(x, 7) for x in okfrac)),
                                "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                                "SOGM_TUNING": os.environ.get("SOGM_TUNING"), "wall_s": time.perf_counter() - t_all}), flush=True)


if __name__ == "__main__":
    main()
