#!/usr/bin/env python
"""Timeline of a flight (sogm_flight_run): per agent-tick stamps -> where the chain spends its time, how far the swarm is
spread, admission rate.  python tools/diag_flight.py [ticks] ["k=v,..."]"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np
import torch

pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tuning = {k: float(x) for k, x in (kv.split("=") for kv in (sys.argv[2] if len(sys.argv) > 2 else "").split(",") if kv)}
A = 128
sw = driver.SwarmTick("cfg2", A, moving_world=True, prestamp=False, grids=1, tuning=tuning)
sw.compute.prepare(0, ticks + 4)
sw.fly(3)
torch.cuda.synchronize()
sw.fly(ticks)
torch.cuda.synchronize()
ts = np.zeros((ticks, A, 12), np.int64)
lib = pop.lib()
lib.sogm_debug_flight_times.restype = C.c_int
lib.sogm_debug_flight_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert lib.sogm_debug_flight_times(sw.planner._p, ts.ctypes.data_as(C.c_void_p), ticks) == 0
t0 = ts[:, :, 7].min()
ms = (ts - t0) / 1e5
names = ["A*start", "A*done", "corr1st", "corrFinal", "QPstart", "QPdone", "finished", "published", "headStart", "admitted",
         "marksDone", "mapReady"]
order = [7, 8, 9, 10, 11, 0, 1, 2, 3, 4, 5, 6]
print("flight of", ticks, "ticks:", round(ms[:, :, 6].max(), 2), "ms =", round(ms[:, :, 6].max() / ticks, 2), "ms per tick")
print("mean interval between consecutive stamps (ms):")
for a, b in zip(order[:-1], order[1:]):
    d = ms[:, :, b] - ms[:, :, a]
    print(f"  {names[a]:>10} -> {names[b]:<10} mean {d.mean():7.3f}  p50 {np.percentile(d, 50):7.3f}  p90 {np.percentile(d, 90):7.3f}  max {d.max():7.3f}")
for k in (0, 1, 2, 5, 10, ticks - 1):
    if k >= ticks:
        continue
    print(f"tick {k}: published {ms[k, :, 7].min():7.2f}..{ms[k, :, 7].max():7.2f}  admitted {ms[k, :, 9].min():7.2f}..{ms[k, :, 9].max():7.2f}  "
          f"mapReady {ms[k, :, 11].min():7.2f}..{ms[k, :, 11].max():7.2f}  finished {ms[k, :, 6].min():7.2f}..{ms[k, :, 6].max():7.2f}")
adm = np.sort(ms[:, :, 9].ravel())
gaps = np.diff(adm)
print("admissions: mean gap %.1f us, p50 %.1f, p90 %.1f, max %.1f" % (gaps.mean() * 1e3, np.percentile(gaps, 50) * 1e3, np.percentile(gaps, 90) * 1e3, gaps.max() * 1e3))
# agents in each stage over time (sampled)
T = np.linspace(ms[:, :, 7].min(), ms[:, :, 6].max(), 60)
print("time   waitAdm  map  search  corridor  qp  finish")
for t in T[::3]:
    w = ((ms[:, :, 7] <= t) & (t < ms[:, :, 9])).sum()
    m = ((ms[:, :, 9] <= t) & (t < ms[:, :, 11])).sum()
    s = ((ms[:, :, 11] <= t) & (t < ms[:, :, 1])).sum()
    c = ((ms[:, :, 1] <= t) & (t < ms[:, :, 3])).sum()
    q = ((ms[:, :, 3] <= t) & (t < ms[:, :, 5])).sum()
    f = ((ms[:, :, 5] <= t) & (t < ms[:, :, 6])).sum()
    print(f"{t:6.1f} {w:7d} {m:5d} {s:6d} {c:8d} {q:5d} {f:6d}")
sw.close()
