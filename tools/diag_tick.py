"""Diagnostic: run a few ticks of the bench scenario and print per-agent stage statistics."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
sogm = importlib.import_module("pred-occ-planner_amd.sogm")
planner = importlib.import_module("pred-occ-planner_amd.planner")
grid = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
A = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 6
sw = driver.SwarmTick(grid, A)
sw.map.set_profiling(True)
P = sw.planner
for k in range(ticks):
    stamp = sw.t0 + sw.tick * driver.TICK_PERIOD
    stamps = torch.full((A,), stamp, dtype=torch.float64, device="cuda")
    t_start = stamps + driver.REPLAN_START_TIME
    pva, valid = planner.traj_eval(sw.own, t_start)
    pva = torch.where(valid.bool().unsqueeze(1), pva, sw.hover).contiguous()
    sw.hover = torch.cat([pva[:, :3], torch.zeros_like(pva[:, 3:])], dim=1)
    poses = pva[:, :3].to(torch.float32).contiguous()
    sw.map.updateMap(sw.dev["cloud"], sw.dev["cloud_range"], sw.dev["cylinders"], sw.dev["n_cyl"], poses, stamps)
    sw.map.addOtherAgents(sw.all, A, sw.dev["ego_ids"])
    s = P.search(pva, sw.goals, t_start)
    c = P.generateCorridors(pva, t_start, s["route"], s["route_len"])
    q = P.optimize(pva, c["goal"], c["polys"], c["nfaces"], c["npoly"])
    ms = sw.map.profile_read()
    st = s["stats"].cpu().numpy(); ret = s["ret"].cpu().numpy()
    nf = c["nfaces"].cpu().numpy(); npoly = c["npoly"].cpu().numpy()
    it = q["iters"].cpu().numpy(); status = q["status"].cpu().numpy()
    m_rows = 9 * (npoly + 1) + 21 * npoly + 5 * nf.sum(axis=1)
    print(f"tick {k}: ms astar {ms[3]:.2f} corridor {ms[4]:.2f} qp {ms[5]:.2f}")
    print("  astar ret", np.bincount(ret, minlength=6), "iters max/mean", st[:, 1].max(), st[:, 1].mean(), "nodes max", st[:, 0].max(), "searches", np.bincount(st[:, 3]))
    print("  npoly", np.bincount(npoly, minlength=9), "faces max", nf.max(), "m max/mean", m_rows.max(), m_rows.mean())
    print("  qp status", {int(k2): int((status == k2).sum()) for k2 in np.unique(status)}, "iters max/mean/median", it.max(), it.mean(), np.median(it))
    # emulate driver bookkeeping
    rec, ok = P.replan(pva, sw.goals, t_start, sw.dev["ego_ids"], sw.new, sw.ok)
    sw.own = torch.where(sw.ok.bool().unsqueeze(1), sw.new, sw.own)
    sw.all.copy_(sw.own)
    sw.tick += 1
