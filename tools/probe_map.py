"""Early GPU probe: SOGM update timing at a BASELINE config (per-kernel HIP-event times)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

pop = importlib.import_module("pred-occ-planner_amd")
sogm = importlib.import_module("pred-occ-planner_amd.sogm")

grid = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
A = int(sys.argv[2]) if len(sys.argv) > 2 else pop.config.AGENTS[grid]
spec = pop.config.make_spec(grid)
sc = pop.scene.make_scene(A, (spec.L // 2) * 0.15, seed=0x5069)
print("agents", A, "cloud", sc["cloud"].shape, "cyl", len(sc["cylinders"]), flush=True)
dev = sogm.upload_scene(sc)
m = sogm.SogmMap(spec, A)
recs = sogm._dev(pop.scene.straight_records(sc))
print("grid GB", m.grid_bytes() / 1e9, flush=True)
m.set_profiling(True)
for it in range(5):
    torch.cuda.synchronize()
    t0 = time.time()
    m.updateMap(dev["cloud"], dev["cloud_range"], dev["cylinders"], dev["n_cyl"], dev["poses"], dev["stamps"])
    m.addOtherAgents(recs, A, dev["ego_ids"])
    torch.cuda.synchronize()
    wall = (time.time() - t0) * 1e3
    ms = m.profile_read()
    print(f"it{it} wall {wall:.2f} ms  clear {ms[0]:.3f} ms ({m.grid_bytes()/ms[0]/1e9:.0f} GB/s)  stamp {ms[1]:.3f}  splat {ms[2]:.3f}", flush=True)
m.close()
