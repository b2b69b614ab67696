"""Diagnostic: per-agent stage timestamps of the dataflow replan (sogm_debug_flow_times) over a few ticks of the
bench workload: where each agent's chain spends its time, and which agent ends the tick."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
A = 128
sw = driver.SwarmTick("cfg2", A, grids=int(os.environ.get("SOGM_GRIDS", "3")),
                      overlap_clear=os.environ.get("SOGM_OVERLAP", "1") != "0")
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 12
every = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # print every n-th tick (long flights)
lib = pop.lib()
names = ["A* start", "A* done", "corr first", "corr final", "QP start", "QP done", "finished"]
for k in range(ticks):
    sw.step()
    if k % every:
        continue
    ts = np.zeros((A, 8), np.int64)
    lib.sogm_debug_flow_times(sw.planner._p, ts.ctypes.data_as(C.c_void_p))
    t0 = ts[:, 7].min() if ts[:, 7].min() > 0 else ts[:, 0].min()  # first A* workgroup resident
    us = (ts[:, :7] - t0) / 100.0
    end = us[:, 6]
    crit = int(np.argmax(end))
    d = lambda a, b: us[:, b] - us[:, a]
    print(f"tick {k}: end {end.max()/1000:.2f} ms (agent {crit}); mean/max ms: A* {d(0,1).mean()/1000:.2f}/{d(0,1).max()/1000:.2f}  wait->corr {d(1,2).mean()/1000:.2f}/{d(1,2).max()/1000:.2f}  corr {d(2,3).mean()/1000:.2f}/{d(2,3).max()/1000:.2f}  wait->QP {d(3,4).mean()/1000:.2f}/{d(3,4).max()/1000:.2f}  QP {d(4,5).mean()/1000:.2f}/{d(4,5).max()/1000:.2f}  fin {d(5,6).mean()/1000:.2f}/{d(5,6).max()/1000:.2f}")
    print("   critical agent:", " | ".join(f"{n} {us[crit, i]/1000:.2f}" for i, n in enumerate(names)))
    if hasattr(lib, "sogm_debug_prestamp_times"):
        pt = np.zeros((A, 4), np.int64)
        lib.sogm_debug_prestamp_times(sw.planner._p, pt.ctypes.data_as(C.c_void_p))
        pu = (pt - t0) / 100.0
        fin = us[:, 6]
        q = lambda v: f"{v.mean()/1000:.2f}/{np.percentile(v, 90)/1000:.2f}/{v.max()/1000:.2f}"
        print(f"   pre-stamp, mean/p90/max ms over agents: finished -> record seen {q(pu[:, 0] - fin)} | -> culled {q(pu[:, 1] - pu[:, 0])} | "
              f"-> bits {q(pu[:, 2] - pu[:, 1])} | -> marks {q(pu[:, 3] - pu[:, 2])} | whole {q(pu[:, 3] - fin)};  first record seen at "
              f"{pu[:, 0].min()/1000:.2f} ms, agents finished before that: {int((fin < pu[:, 0].min()).sum())}, last marks {pu[:, 3].max()/1000:.2f}")
        last = np.argsort(-pu[:, 3])[:4]  # the agents whose pre-stamp ended last
        for a in last:
            print(f"   pre-stamp of agent {a}: finished {us[a, 6]/1000:.2f} | record seen {pu[a, 0]/1000:.2f} | culled {pu[a, 1]/1000:.2f} | bits {pu[a, 2]/1000:.2f} | marks {pu[a, 3]/1000:.2f}")
