"""The chunked, width-adaptive SOGM clear (k_clear_chunks) by itself, for counter collection: rocprofv3 --pmc
serialises kernels, under which the dataflow replan cannot run (its persistent kernels wait for each other), so the
traffic of the in-tick clear kernels is measured here — the pooled update queues the clear of the swapped-out grid
(tuning key clear_early = 1 makes the update itself queue it) and nothing else runs.  Serialised, the narrow launch clears the
whole grid (the wide launch finds the cursor exhausted): bytes per launch = the grid.

    SOGM_TUNING=clear_early=1 rocprofv3 --pmc WRITE_SIZE -d /tmp/cw -- python tools/diag_clear_pmc.py
"""
import importlib, os, sys
os.environ.setdefault("SOGM_TUNING", "clear_early=1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
driver = importlib.import_module("pred-occ-planner_amd.driver")
sw = driver.SwarmTick("cfg2", 128)
sw.map.set_profiling(True)
for _ in range(5):  # every update queues the clear of the grid it swaps out (the first one: of the dirty spares); nothing else runs
    sw.compute.tick_inputs(sw.own, sw.t0, sw.hover, sw.now, sw.t_start, sw.pva, sw.poses)
    sw.map.updateMap(sw.dev["cloud"], sw.dev["cloud_range"], sw.dev["cylinders"], sw.dev["n_cyl"], sw.poses, sw.now)
    torch.cuda.synchronize()
print("clear launches timed:", len(sw.map.profile_read_all(0)), "ms:", [round(x, 2) for x in sw.map.profile_read_all(0)])
sw.close()
