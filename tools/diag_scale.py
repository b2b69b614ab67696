"""Diagnostic: one rank's tick at multi-GPU scene scale on a single GPU (the other ranks' records are a
stub: straight-line trajectories, refreshed never) — checks that per-rank work does not grow with N."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
sogm = importlib.import_module("pred-occ-planner_amd.sogm")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12


class StubDist:
    """all_gather stand-in: own shard goes to its slice, the other shards keep their initial records."""
    def __init__(self, rank, a_loc):
        self.rank, self.a_loc = rank, a_loc
    def is_initialized(self):
        return True
    def get_backend(self):
        return "stub"   # not "nccl": the driver uses all_gather_into_tensor below
    def all_gather_into_tensor(self, out, own):
        out[self.rank * self.a_loc:(self.rank + 1) * self.a_loc].copy_(own)


t0 = time.time()
sw = driver.SwarmTick("cfg2", 128, rank, world, 0, dist=StubDist(rank, 128))
print("setup s", round(time.time() - t0, 1), "agents total", sw.A_tot, "cylinders", len(sw.scene["cylinders"]))
# other shards: constant-velocity records so that overlay / deconfliction have real work
recs = pop.scene.straight_records(sw.scene)
sw.all.copy_(sogm._dev(recs).view(sw.A_tot, -1))
for k in range(3):
    sw.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
oks = [sw.step() for _ in range(steps)]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"world {world} rank {rank}: {dt*1e3:.2f} ms/tick -> {128/dt:.0f} replans/s per GPU, ok {torch.stack(oks).float().mean().item():.3f}")
# per-agent stage times of the last tick (in-kernel stamps of the dataflow replan)
import ctypes as C
ts = np.zeros((128, 8), np.int64)
pop.lib().sogm_debug_flow_times(sw.planner._p, ts.ctypes.data_as(C.c_void_p))
t0 = ts[:, 7].min() if ts[:, 7].min() > 0 else ts[:, 0].min()
us = (ts[:, :7] - t0) / 100.0
d = lambda a, b: us[:, b] - us[:, a]
print(f"last tick: end {us[:, 6].max()/1000:.2f} ms; mean/max ms: A* {d(0,1).mean()/1000:.2f}/{d(0,1).max()/1000:.2f}  corr {d(2,3).mean()/1000:.2f}/{d(2,3).max()/1000:.2f}  "
      f"wait->QP {d(3,4).mean()/1000:.2f}/{d(3,4).max()/1000:.2f}  QP {d(4,5).mean()/1000:.2f}/{d(4,5).max()/1000:.2f}  fin {d(5,6).mean()/1000:.2f}/{d(5,6).max()/1000:.2f}")
