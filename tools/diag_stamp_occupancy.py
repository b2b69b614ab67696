import importlib, os, sys, json, time
sys.path.insert(0, os.environ.get("SOGM_REPO", "/root/repo"))
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
sogm = importlib.import_module("pred-occ-planner_amd.sogm")
for kb in (0, 10, 20, 26, 40):
    sw = driver.SwarmTick("cfg2", 128, moving_world=True, prestamp=False, grids=1, tuning={"stamp_lds_kb": kb})
    sw.map.set_overlap_clear(False)
    sw.map.set_profiling(True)
    poses = sogm._dev(sw.scene["poses"], np.float32); stamps = sogm._dev(sw.scene["stamps"], np.float64)
    ms = []
    for k in range(6):
        sw.map.updateWorld(sw.compute.world(k), poses, stamps)
        ms.append(round(sw.map.profile_read()[1], 3))
    print("stamp_lds_kb", kb, "layout", "tiled" if sw.spec.storage & 16 else "rows", "stamp ms", ms, flush=True)
    sw.close()
