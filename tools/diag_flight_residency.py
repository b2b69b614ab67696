#!/usr/bin/env python
"""Which workgroups of a flight's four kernels are resident from the start?  sogm_flight_run's liveness argument needs every
workgroup of every kernel to be running before the first hand-over (a workgroup that is still waiting in the dispatcher when
the hardware scheduler saves and restores the queues takes a restored wave's place).  Flies one flight and prints, per kernel,
how many workgroups were launched, how many started within 1 ms of the first one, and when the others started.
    python tools/diag_flight_residency.py [ticks]       env: SOGM_TUNING"""
import ctypes as C, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
lib = pop.lib()
lib.sogm_debug_flight_wg_starts.restype = C.c_int
lib.sogm_debug_flight_wg_starts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]


def residency(sw):
    st = np.zeros((8, 4096), np.int64)
    wgs = np.zeros((4,), np.int32)
    assert lib.sogm_debug_flight_wg_starts(sw.planner._p, st.ctypes.data_as(C.c_void_p), wgs.ctypes.data_as(C.c_void_p)) == 0
    t0 = st[:4][st[:4] > 0].min()
    out = {}
    for k, name in enumerate(("qp", "search", "light", "map")):
        s = st[k, :wgs[k]]
        ms = (s[s > 0] - t0) / 1e5
        out[name] = {"launched": int(wgs[k]), "ran": int((s > 0).sum()), "within_1ms": int((ms < 1.0).sum()),
                     "late_start_ms": [round(float(x), 2) for x in np.sort(ms[ms >= 1.0])[:6]] + (["..."] if (ms >= 1.0).sum() > 6 else []),
                     "last_start_ms": round(float(ms.max()), 2) if len(ms) else None}
        # workgroups per compute unit among the early starters: {workgroups on a CU: how many CUs}, and per (XCC, SE)
        early = (s > 0) & ((s - t0) / 1e5 < 1.0)
        hw = st[4 + k, :wgs[k]][early]
        cu_key = ((hw >> 32) & 0xF) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 8) & 15)
        per_cu = np.unique(cu_key, return_counts=True)[1]
        out[name]["cus_used"] = int(len(per_cu))
        out[name]["wgs_per_cu_hist"] = {int(a): int(b) for a, b in zip(*np.unique(per_cu, return_counts=True))}
        se_key = ((hw >> 32) & 0xF) * 10 + ((hw >> 13) & 7)
        if os.environ.get("VERBOSE"):
            out[name]["per_xcc_se"] = {int(a): int(b) for a, b in zip(*np.unique(se_key, return_counts=True))}
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    grid = os.environ.get("GRID", "cfg2")
    sw = driver.SwarmTick(grid, pop.config.AGENTS[grid], moving_world=True, prestamp=False, grids=1)
    sw.compute.prepare(0, n + 4)
    sw.fly(3)
    torch.cuda.synchronize()
    sw.fly(n)
    torch.cuda.synchronize()
    print("RESIDENCY " + json.dumps({"tuning": os.environ.get("SOGM_TUNING"), "ticks": n, **residency(sw)}), flush=True)
    sw.close()
