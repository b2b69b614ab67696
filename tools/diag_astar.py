"""Diagnostic: A* per-phase time for the slowest agents after N ticks."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
planner = importlib.import_module("pred-occ-planner_amd.planner")
A = 128
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sw = driver.SwarmTick("cfg2", A, overlap_clear=False)
sw.map.set_profiling(True)
lib = pop.lib()
lib.sogm_debug_astar_stats.argtypes = [C.c_void_p, C.c_void_p]
worst = None
for k in range(nt):
    sw.step()
    ms = sw.map.profile_read()
    out = np.zeros((A, 8), np.int64)
    lib.sogm_debug_astar_stats(sw.planner._p, out.ctypes.data)
    i = np.argmax(out[:, :5].sum(axis=1))
    tot = out[i, :5].sum() / 100.0
    print(f"tick {k}: astar {ms[3]:.2f} ms; slowest agent {i}: {tot:.0f} us, {out[i,5]} expansions -> us/exp "
          f"pop {out[i,0]/100/max(out[i,5],1):.1f} eval {out[i,1]/100/max(out[i,5],1):.1f} dup {out[i,2]/100/max(out[i,5],1):.1f} "
          f"merge {out[i,3]/100/max(out[i,5],1):.1f} write {out[i,4]/100/max(out[i,5],1):.1f}")
