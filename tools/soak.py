"""Soak: many ticks of the bench workload (and of the FSM mode); reports tick-time distribution and replans_ok."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
driver = importlib.import_module("pred-occ-planner_amd.driver")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
fsm = len(sys.argv) > 2 and sys.argv[2] == "fsm"
sw = driver.SwarmTick("cfg2", 128, fsm=fsm)
ts, oks = [], []
for k in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ok = sw.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    oks.append(float(ok.float().mean().item()))
ts = np.array(ts)
print(f"{'fsm' if fsm else 'tick'} x{n}: ms mean {ts.mean():.2f} p50 {np.median(ts):.2f} p99 {np.percentile(ts, 99):.2f} max {ts.max():.2f}; ok mean {np.mean(oks):.3f} min {np.min(oks):.3f}")
if fsm:
    st = sw.status.cpu().numpy()
    print("final FSM states", np.bincount(st, minlength=4), "fail max", int(sw.fail.max().item()))
pos = driver.traj_eval(sw.own, sw.now)[0][:, :3]
d = (pos - sw.goals).norm(dim=1)
print("distance to goal: mean %.2f min %.2f; finite %s" % (d.mean().item(), d.min().item(), bool(torch.isfinite(pos).all())))
