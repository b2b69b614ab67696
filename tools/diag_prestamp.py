"""Diagnostic (profiling build: make -C pred-occ-planner_amd/csrc EXTRA=-DSOGM_PROFILE_PRESTAMP, or SOGM_LIB_PATH to such a
build): where the waves of the dataflow replan's pre-stamp (k_prestamp_flow) spend their time over a few ticks of the
bench workload — waiting for a published agent, waiting for the agent's earlier passes, cull, bits, marks (candidate
walk / slice loops).

    python tools/diag_prestamp.py [ticks=10]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
lib = pop.lib()
if not hasattr(lib, "sogm_debug_prestamp_prof"):
    raise SystemExit("not a profiling build (make EXTRA=-DSOGM_PROFILE_PRESTAMP)")
lib.sogm_debug_prestamp_prof.argtypes = [C.c_void_p, C.c_int]
A = 128
sw = driver.SwarmTick("cfg2", A)
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(4):
    sw.step()
torch.cuda.synchronize()
out = np.zeros(12, np.uint64)
lib.sogm_debug_prestamp_prof(out.ctypes.data_as(C.c_void_p), 1)
for _ in range(ticks):
    sw.step()
torch.cuda.synchronize()
lib.sogm_debug_prestamp_prof(out.ctypes.data_as(C.c_void_p), 1)
o = out.astype(np.float64)
ms = o[:7] / 1e5 / ticks  # wave-milliseconds per tick
print(f"per tick, wave-milliseconds: waiting for a published agent {ms[0]:.1f} | waiting for cull / bits of the agent {ms[1]:.1f} | "
      f"cull {ms[2]:.2f} | bits {ms[3]:.1f} | marks {ms[4]:.1f} (candidate walk {ms[5]:.1f}, slice loops {ms[6]:.1f})")
print(f"tickets per tick {o[7] / ticks:.0f}; chunks of 64 voxels per tick {o[8] / ticks:.0f}: {o[5] / max(o[8], 1) / 100:.2f} us walk + "
      f"{o[6] / max(o[8], 1) / 100:.2f} us slices per chunk; marks ticket {o[4] / 100 / (ticks * A * 64):.1f} us (if 64 per agent), bits ticket {o[3] / 100 / (ticks * A * 32):.1f} us")
sw.close()
