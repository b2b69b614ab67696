"""Micro: duration of k_clear_slabs alone on an idle GPU for the width given by SOGM_TUNING="clear_wgs=..,clear_throttle=.."
(128 agents x 200^3 x 20 = 81.92 GB), via the in-stream clear of an un-pipelined update."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
sogm = importlib.import_module("pred-occ-planner_amd.sogm")
A = 128
spec = pop.config.make_spec("cfg2")
sc = pop.scene.make_scene(A, 15.0, seed=3, n_cyl=4)
dev = sogm.upload_scene(sc)
m = sogm.SogmMap(spec, A)
m.set_profiling(True)
ts = []
for _ in range(4):
    m.updateMap(dev["cloud"], dev["cloud_range"], dev["cylinders"], dev["n_cyl"], dev["poses"], dev["stamps"])
    torch.cuda.synchronize()
    ts.append(m.profile_read()[0])
print("tuning", os.environ.get("SOGM_TUNING", "defaults"),
      "clear ms", [round(t, 2) for t in ts], "TB/s", round(81.92 / min(ts[1:]), 2))
