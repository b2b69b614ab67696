import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
sw = driver.SwarmTick("cfg2", 128, overlap_clear=False)
sw.map.set_profiling(True)
for _ in range(3):
    sw.compute.tick_inputs(sw.own, sw.t0, sw.hover, sw.now, sw.t_start, sw.pva, sw.poses)
    sw.map.updateMap(sw.dev["cloud"], sw.dev["cloud_range"], sw.dev["cylinders"], sw.dev["n_cyl"], sw.poses, sw.now)
    ms = sw.map.profile_read()
print("stamp_wgs", os.environ.get("SOGM_TUNING"), "clear", round(ms[0], 3), "stamp (cull+bits+marks) ms", round(ms[1], 3), "n_cyl", sw.dev["n_cyl"])
