"""Sustained lock-step ticks (bench.py's scene), every tick host-synchronised: which QP solves have slow termination checks?
Per tick the solves' own clocks (sogm_debug_qp_stats) are read; a solve whose mean check exceeds `--us` microseconds (a
check is ~4.7 us) is printed with its columns: solve ms, iterations, checks, check ms split into spill of the row state |
residual pass | the rest (tests + infeasibility certificate), shader clocks / wall time (GHz) — and the tick's chain.
    python tools/diag_qp_checks.py [--ticks 300] [--us 10]"""
import argparse, ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch

ap = argparse.ArgumentParser()
ap.add_argument("--ticks", type=int, default=300)
ap.add_argument("--skip", type=int, default=23)
ap.add_argument("--us", type=float, default=10.0)
a = ap.parse_args()
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
bench = importlib.import_module("bench")
sw = driver.SwarmTick("cfg2", 128, 0, 1, 0, moving_world=True, prestamp=False)
sw.compute.prepare(0, a.skip + a.ticks)
for _ in range(a.skip):
    sw.step()
torch.cuda.synchronize()
A = sw.A_loc
slow, tm = 0, []
for k in range(a.ticks):
    t0 = time.perf_counter()
    sw.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    tm.append(ms)
    st = np.zeros((A, 16), np.int64)
    pop.lib().sogm_debug_qp_stats(sw.planner._p, st.ctypes.data_as(C.c_void_p))
    n = np.maximum(st[:, 5], 1)
    per = st[:, 4] / n / 100.0          # us per check
    for q in np.flatnonzero((st[:, 5] > 0) & (per > a.us)):
        slow += 1
        ch = bench.flow_chain(pop, sw) or {}
        print(f"tick {sw.tick - 1} ({ms:.2f} ms) agent {q}: solve {st[q, 0] / 1e5:.2f} ms, {st[q, 6]} iterations, {st[q, 5]} checks "
              f"{st[q, 4] / 1e5:.2f} ms = spill {st[q, 12] / 1e5:.2f} | residuals {st[q, 13] / 1e5:.2f} | rest "
              f"{(st[q, 4] - st[q, 12] - st[q, 13]) / 1e5:.2f}; refactorisations {st[q, 3]} {st[q, 2] / 1e5:.2f} ms; "
              f"{st[q, 11] / max(st[q, 0] * 10.0, 1.0):.2f} GHz; flags {st[q, 7]}; tick astar_max {ch.get('astar_max', 0):.2f} "
              f"qp_max {ch.get('qp_max', 0):.2f} chain_end {ch.get('chain_end', 0):.2f}", flush=True)
tm = np.array(tm)
print(f"{a.ticks} ticks: {slow} slow-check solves; tick mean {tm.mean():.2f} p50 {np.median(tm):.2f} p99 {np.percentile(tm, 99):.2f} max {tm.max():.2f} ms")
sw.close()
