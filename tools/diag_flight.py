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
ts = np.zeros((ticks, A, 16), np.int64)
lib = pop.lib()
lib.sogm_debug_flight_times.restype = C.c_int
lib.sogm_debug_flight_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert lib.sogm_debug_flight_times(sw.planner._p, ts.ctypes.data_as(C.c_void_p), ticks) == 0
t0 = ts[:, :, 7].min()
ms = (ts - t0) / 1e5
names = ["A*start", "A*done", "corr1st", "corrFinal", "QPstart", "QPdone", "finished", "published", "headStart", "admitted",
         "marksDone", "mapReady", "headDone", "bitsDone", "splatQueued"]
order = [7, 8, 9, 10, 14, 11, 0, 1, 2, 3, 4, 5, 6]
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
# The critical path of the flight: walk back from the last finish.  An agent-tick starts either at the agent's own previous
# finish (edge "own") or when tick k - 2 is complete (edge "gate": its last finisher); what follows is the agent-tick's
# chain, split into waiting for admission, map, search, corridors, QP, finish.
QP_ITERS = sw.planner.last_qp_iterations() if hasattr(sw.planner, "last_qp_iterations") else None
k, a = ticks - 1, int(np.argmax(ms[ticks - 1, :, 6]))
path = []
while k >= 0:
    pub, hs, adm, mk, sp, mr = ms[k, a, 7], ms[k, a, 8], ms[k, a, 9], ms[k, a, 10], ms[k, a, 14], ms[k, a, 11]
    seg = {"tick": k, "agent": a, "parked": hs - pub, "admission": adm - hs, "map": (mk - adm) + (mr - sp), "gate": 0.0,
           "search": ms[k, a, 1] - mr, "corridor": ms[k, a, 3] - ms[k, a, 1], "qp": ms[k, a, 5] - ms[k, a, 3],
           "finish": ms[k, a, 6] - ms[k, a, 5]}
    gate_open = ms[k - 2, :, 6].max() if k >= 2 else -1.0
    if k >= 2 and gate_open > mk + 1e-3:   # the overlay was parked until the gate opened: go to the last finisher of k - 2
        seg["edge"] = "gate"
        seg["parked"] = seg["admission"] = 0.0
        seg["gate"] = sp - gate_open           # hand-over from the finish that opened the gate
        seg["map"] = mr - sp                   # (only the overlay is on the path)
        path.append(seg)
        a = int(np.argmax(ms[k - 2, :, 6]))
        k -= 2
    else:
        seg["edge"] = "own"
        seg["gate"] = sp - mk
        path.append(seg)
        k -= 1
path.reverse()
print("critical path (edge = how the agent-tick was reached; ms):")
tot = {}
for s_ in path:
    print("  tick %2d agent %3d via %-4s parked %.2f adm %.2f map %.2f gate %.2f A* %.2f corr %.2f qp %.2f fin %.2f" % (
        s_["tick"], s_["agent"], s_["edge"], s_["parked"], s_["admission"], s_["map"], s_["gate"], s_["search"], s_["corridor"], s_["qp"], s_["finish"]))
    for key in ("parked", "admission", "map", "gate", "search", "corridor", "qp", "finish"):
        tot[key] = tot.get(key, 0.0) + s_[key]
print("critical path totals (ms):", {k_: round(v, 2) for k_, v in tot.items()}, "sum", round(sum(tot.values()), 2),
      "own edges", sum(1 for s_ in path if s_["edge"] == "own"), "gate edges", sum(1 for s_ in path if s_["edge"] == "gate"))
# agent-ticks that went through the urgent lane (no admission wait, head at once) against the others: the map's phases
adm_w = ms[:, :, 9] - ms[:, :, 8]
urg = (adm_w < 0.004) & (ms[:, :, 8] - ms[:, :, 7] < 0.06)
urg[0] = False
for name, sel in (("urgent lane", urg), ("plain lane", ~urg)):
    if sel.sum() == 0:
        continue
    h_d = (ms[:, :, 12] - ms[:, :, 9])[sel]
    r_b = (ms[:, :, 13] - ms[:, :, 12])[sel]
    mk = (ms[:, :, 10] - ms[:, :, 13])[sel]
    print(f"{name}: head {h_d.mean():.3f} (p90 {np.percentile(h_d, 90):.3f})  reset + bits {r_b.mean():.3f} (p90 {np.percentile(r_b, 90):.3f})  marks {mk.mean():.3f} (p90 {np.percentile(mk, 90):.3f})")
    a_m = (ms[:, :, 10] - ms[:, :, 9])[sel]
    m_r = (ms[:, :, 11] - ms[:, :, 10])[sel]
    print(f"{name}: {int(sel.sum())} agent-ticks; admitted -> marksDone mean {a_m.mean():.3f} p50 {np.percentile(a_m, 50):.3f} p90 {np.percentile(a_m, 90):.3f} max {a_m.max():.3f};"
          f" marksDone -> mapReady mean {m_r.mean():.3f} p90 {np.percentile(m_r, 90):.3f} max {m_r.max():.3f}")
# how many urgent maps are under construction at once, sampled at every urgent admission
ua, ur = np.sort(ms[:, :, 9][urg]), np.sort(ms[:, :, 11][urg])
if len(ua):
    conc = [int((ua <= t).sum() - (ur <= t).sum()) for t in ua]
    print("urgent maps under construction at an urgent admission: mean %.1f p90 %d max %d" % (np.mean(conc), np.percentile(conc, 90), max(conc)))
last = [int(np.argmax(ms[k_, :, 6])) for k_ in range(ticks)]
print("last finisher per tick:", last)
sw.close()
