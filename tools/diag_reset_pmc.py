"""The sparse reset of the SOGM (k_reset_sectors) and the logging stamp by themselves, for counter collection
(rocprofv3 --pmc serialises kernels, under which the dataflow replan cannot run): single-grid mode, every update =
reset of the logged sectors + stamp + overlay of the same scene, nothing else runs.  Prints the entry count the
resets read, which tools/make_profile_md.py divides the PMC bytes by.

    rocprofv3 --pmc WRITE_SIZE -d /tmp/rw -- python tools/diag_reset_pmc.py
"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
driver = importlib.import_module("pred-occ-planner_amd.driver")
sw = driver.SwarmTick("cfg2", 128, overlap_clear=False, moving_world=True, prestamp=False)  # (SOGM_LAYOUT=rows|tiled)
print("cell order:", "2x2x2 tiles" if sw.spec.storage & 16 else "x-fastest rows")
sw.map.set_profiling(True)
entries, moved = [], []
sw.map.map_traffic(reset=True)
for k in range(6):  # update 0 clears densely (the grid is untracked), updates 1.. reset the logged sectors
    sw.compute.tick_inputs(sw.own, sw.t0, sw.hover, sw.now, sw.t_start, sw.pva, sw.poses)
    sw.compute.update_map(sw.poses, sw.now, sw.all, sw.A_tot, k)   # sogm_update_world: frame k, cropped on the device
    torch.cuda.synchronize()
    moved.append(sw.map.map_traffic(reset=True))  # (before the state query: that one restarts the reset's counters)
    st = sw.map.sparse_reset_state()
    entries.append(st["total_entries"])
ms = sw.map.profile_read_all(0)
print("reset launches (first = dense clear) ms:", [round(x, 3) for x in ms])
print("log entries after each update:", entries, "max per agent", st["max_entries"], "capacity", st["log_capacity"])
import json
print("device-side counts per update (sogm_map_traffic): " + json.dumps(moved))
print("stamp (cull + bits + marks) launches ms:", [round(x, 3) for x in sw.map.profile_read_all(1)])
sw.close()
