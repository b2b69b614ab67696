#!/usr/bin/env python
"""Flies the bench swarm lock-step with the update flow on, host-synchronised every tick, until a tick fails or N ticks
have passed; on a failure prints the flow's control words (ticket, error, per-agent progress, map_ready vs epoch).
    python tools/diag_update_flow.py [ticks] [agents]"""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
A = int(sys.argv[2]) if len(sys.argv) > 2 else 128
FLOW = int(sys.argv[3]) if len(sys.argv) > 3 else 1
TUNE = dict(kv.split("=") for kv in sys.argv[4].split(",")) if len(sys.argv) > 4 and sys.argv[4] else {}
sw = driver.SwarmTick("cfg2", A, moving_world=True, prestamp=False, tuning={"update_flow": FLOW, **{k: float(v) for k, v in TUNE.items()}})
sw.compute.prepare(0, N + 2)
sw.map.device_clock()   # (allocates the tick clock words)
lib = pop.lib()
ms = []
try:
    for k in range(N):
        t0 = time.perf_counter()
        sw.step()
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
        if k % 25 == 5:
            ts = np.zeros((A, 8), np.int64)
            lib.sogm_debug_flow_times(sw.planner._p, ts.ctypes.data_as(C.c_void_p))
            c0 = round(sw.map.tick_clock()[0] * 1e8)   # the update's first kernel (100 MHz ticks)
            st, fin = (ts[:, 0] - c0) / 1e5, (ts[:, 6] - c0) / 1e5
            crit = int(np.argmax(fin))
            if FLOW:
                uts, uo = np.zeros((A, 4), np.int64), np.zeros(A, np.int32)
                # (read BEFORE the next update: the order array is re-ranked at the end of the replan — the order the
                #  flow took the agents in this tick is the one of the previous replan; good enough for a timeline)
                lib.sogm_debug_update_flow_times(sw.map.ctx, uts.ctypes.data_as(C.c_void_p), uo.ctypes.data_as(C.c_void_p))
                u = (uts - c0) / 1e5
                by_ready = np.argsort(u[:, 3])
                pick = [by_ready[0], by_ready[A // 4], by_ready[A // 2], by_ready[-1]]
                print("   update flow, ms after the update's first kernel {first ticket, bits done, marks done, ready}: " +
                      " | ".join(f"#{int(np.flatnonzero(by_ready == a)[0])} agent {a}: " + " ".join(f"{x:.2f}" for x in u[a]) for a in pick))
            print(f"tick {k}: {ms[-1]:.2f} ms | search starts (ms after the update's first kernel) min {st.min():.2f} mean {st.mean():.2f} "
                  f"max {st.max():.2f} | chains mean {(fin - st).mean():.2f} max {(fin - st).max():.2f} | end {fin.max():.2f} by agent {crit}: "
                  f"start {st[crit]:.2f} chain {fin[crit] - st[crit]:.2f}; rank of its start {int((st < st[crit]).sum())}")
    print(f"flow {FLOW} {TUNE}: {N} ticks, no failure; tick ms mean {np.mean(ms):.2f} p50 {np.median(ms):.2f} max {np.max(ms):.2f}")
except RuntimeError as e:
    print("FAILED at tick", len(ms), e)
    out = (C.c_int32 * (10 + 2 * A))()
    rc = lib.sogm_debug_update_flow(sw.map.ctx, out, len(out))
    o = np.array(out[:])
    print("rc", rc, "epoch", o[0], "pending", o[1], "ticket", o[2], "err", o[3])
    stage, ready = o[10:10 + A], o[10 + A:10 + 2 * A]
    per = int(sw.map.get_tuning("update_bits") + sw.map.get_tuning("update_marks") + sw.map.get_tuning("update_splat"))
    print("tickets per agent", per, "total", per * A)
    print("stage != per:", {int(a): int(stage[a]) for a in np.flatnonzero(stage != per)})
    print("map_ready != epoch:", {int(a): int(ready[a]) for a in np.flatnonzero(ready != o[0])})
    print("flow failures", sw.planner.flow_failures(), "flow error", sw.planner.flow_error())
