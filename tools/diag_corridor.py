"""Diagnostic: per-segment corridor phase timings from the in-kernel counters."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
planner = importlib.import_module("pred-occ-planner_amd.planner")
A = 128
sw = driver.SwarmTick("cfg2", A, overlap_clear=False)
for _ in range(3):
    sw.step()
sw.map.set_profiling(True)
P = sw.planner
stamp = sw.t0 + sw.tick * driver.TICK_PERIOD
stamps = torch.full((A,), stamp, dtype=torch.float64, device="cuda")
t_start = stamps + driver.REPLAN_START_TIME
pva, valid = planner.traj_eval(sw.own, t_start)
pva = torch.where(valid.bool().unsqueeze(1), pva, sw.hover).contiguous()
poses = pva[:, :3].to(torch.float32).contiguous()
sw.map.updateMap(sw.dev["cloud"], sw.dev["cloud_range"], sw.dev["cylinders"], sw.dev["n_cyl"], poses, stamps)
sw.map.addOtherAgents(sw.all, A, sw.dev["ego_ids"])
s = P.search(pva, sw.goals, t_start)
if os.environ.get("SOGM_CONTEND") == "2":
    # contention source: the real SOGM clear (+ stamp) of a second map on a side stream
    sogm_mod = importlib.import_module("pred-occ-planner_amd.sogm")
    m2 = sogm_mod.SogmMap(sw.spec, A)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        ev0.record(side)
        for _ in range(2):
            m2.updateMap(sw.dev["cloud"], sw.dev["cloud_range"], sw.dev["cylinders"], sw.dev["n_cyl"], poses, stamps)
        ev1.record(side)
elif os.environ.get("SOGM_CONTEND"):
    # contention source: a streaming zero-fill of a 16 GiB tensor on a side stream while the corridors run
    big = torch.empty(int(os.environ.get("SOGM_CONTEND_GB", "16")) << 28, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(12):
            big.zero_()
c = P.generateCorridors(pva, t_start, s["route"], s["route_len"])
ms = sw.map.profile_read()
lib = pop.lib()
lib.sogm_debug_corridor_stats.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros((A * 16, 16), np.int64)
lib.sogm_debug_corridor_stats(P._p, out.ctypes.data)
d = out[out[:, 10] > 0]
us = lambda t: t / 100.0  # 100 MHz
print("corridor kernel ms", round(ms[4], 2), "segments", len(d))
if os.environ.get("SOGM_CONTEND") == "2":
    torch.cuda.synchronize()
    print("SUMMARY tuning", os.environ.get("SOGM_TUNING"),
          "| 2 x (clear+stamp) ms", round(ev0.elapsed_time(ev1), 2), "| corridor ms", round(ms[4], 2),
          "| points us max/mean", round(d[:, 5].max() / 100.0, 1), round(d[:, 5].mean() / 100.0, 1))
print("N pts: max/mean", d[:, 0].max(), d[:, 0].mean(), " nH0 max/mean", d[:, 1].max(), d[:, 1].mean(), " nH1 max", d[:, 2].max())
print("lbfgs iters max/mean", d[:, 3].max(), d[:, 3].mean(), " evals max/mean", d[:, 4].max(), d[:, 4].mean())
for name, col in (("points", 5), ("firi0 done", 6), ("mvie total", 7), ("  lbfgs", 9), ("firi1 done", 8), ("segment total", 10)):
    print(f"{name:14s} us: max {us(d[:, col].max()):9.1f}  mean {us(d[:, col].mean()):9.1f}  p90 {us(np.percentile(d[:, col], 90)):9.1f}")
if d[:, 11].max() > 0:
    f = d[:, 11] / np.maximum(d[:, 9], 1) * 100.0
    print(f"shader clock during the L-BFGS (clock64 / wall_clock64): mean {f.mean():.0f} MHz, min {f.min():.0f}, max {f.max():.0f}")
i = np.argmax(d[:, 10])
print("slowest segment:", d[i])

if hasattr(lib, "sogm_debug_mvie_prof"):
    pr = np.zeros(7, np.uint64)
    lib.sogm_debug_mvie_prof.argtypes = [C.c_void_p]
    lib.sogm_debug_mvie_prof(pr.ctypes.data)
    if pr[1]:
        print(f"costMVIE (line search): {int(pr[1])} calls, mean {pr[0] / pr[1] / 100.0:.2f} us")
        it = max(int(pr[4]), 1)
        print(f"L-BFGS per iteration: line search {pr[2] / it / 100.0:.2f} us, direction update (two-loop) "
              f"{pr[3] / it / 100.0:.2f} us (its two loops alone {pr[6] / it / 100.0:.2f}), mean history {pr[5] / it:.1f}, iterations {it}")
