"""Which capacity limit do long flights hit?  Flies the bench scene (SwarmTick.step, fused path) for N ticks and
keeps, over the whole flight, the largest obstacle-point count of a corridor box, the largest number of planes FIRI
selected in either iteration, and the ticks / agents that reported a capacity hit (sogm_planner_counters)."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
A = 128
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 323
sw = driver.SwarmTick("cfg2", A)
lib = pop.lib()
lib.sogm_debug_corridor_stats.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros((A * 16, 16), np.int64)
mx = np.zeros(3, np.int64)
hist_n, hist_h = [], []
cap_ticks = []
prev = 0
for k in range(ticks):
    sw.step()
    lib.sogm_debug_corridor_stats(sw.planner._p, out.ctypes.data)  # synchronises
    d = out[out[:, 10] > 0]
    if len(d):
        mx = np.maximum(mx, d[:, :3].max(axis=0))
        hist_n.append(int(d[:, 0].max()))
        hist_h.append(int(max(d[:, 1].max(), d[:, 2].max())))
    c = sw.planner.counters()["corridor_capacity"]
    if c != prev:
        cap_ticks.append((k, c - prev, int(d[:, 0].max()), int(d[:, 1].max()), int(d[:, 2].max())))
        prev = c
print("flight of", ticks, "ticks: max points in a box", int(mx[0]), "| max planes FIRI iteration 0 / 1:", int(mx[1]), int(mx[2]))
print("points per box, max per tick: p50 / p99", np.percentile(hist_n, 50), np.percentile(hist_n, 99))
print("planes, max per tick: p50 / p99", np.percentile(hist_h, 50), np.percentile(hist_h, 99))
print("capacity hits (tick, hits, max N, max nH0, max nH1):", cap_ticks)
print("counters", sw.planner.counters())
sw.close()
