"""BASELINE configs[4] (128 agents, 300^3 x 30, fp16 occupancy cells): the QP stage with fp64 and with fp32
residual checks (SogmQpSettings.residual_fp32) on the same corridors of a few ticks of the flight — stage time,
statuses, iteration counts, coefficient differences."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
planner = importlib.import_module("pred-occ-planner_amd.planner")
grid = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
A = 128
sw = driver.SwarmTick(grid, A, overlap_clear=False)
qs32 = pop.config.make_qp_settings()
qs32.residual_fp32 = 1
P, P32 = sw.planner, planner.SogmPlanner(sw.map, pop.config.make_astar_params(), pop.config.make_planner_params(True), qs32)
sw.map.set_profiling(True)
for tick in range(6):
    stamp = sw.t0 + sw.tick * driver.TICK_PERIOD
    stamps = torch.full((A,), stamp, dtype=torch.float64, device="cuda")
    sw.now.copy_(stamps)
    t_start = stamps + driver.REPLAN_START_TIME
    pva, valid = planner.traj_eval(sw.own, t_start)
    pva = torch.where(valid.bool().unsqueeze(1), pva, sw.hover).contiguous()
    sw.hover = torch.cat([pva[:, :3], torch.zeros_like(pva[:, 3:])], dim=1)
    poses = pva[:, :3].to(torch.float32).contiguous()
    sw.map.updateMap(sw.dev["cloud"], sw.dev["cloud_range"], sw.dev["cylinders"], sw.dev["n_cyl"], poses, stamps)
    sw.map.addOtherAgents(sw.all, sw.A_tot, sw.dev["ego_ids"])
    s = P.search(pva, sw.goals, t_start)
    c = P.generateCorridors(pva, t_start, s["route"], s["route_len"])
    res = {}
    for name, pl in (("fp64", P), ("fp32", P32)):
        for _ in range(2):
            q = pl.optimize(pva, c["goal"], c["polys"], c["nfaces"], c["npoly"])
            ms = sw.map.profile_read()
        res[name] = ({k: v.cpu().numpy() for k, v in q.items()}, float(ms[5]))
    a, b = res["fp64"][0], res["fp32"][0]
    ok = np.isin(a["status"], (1, 2)) & (a["status"] == b["status"])
    print(f"tick {sw.tick}: qp stage ms fp64 {res['fp64'][1]:.3f} fp32 {res['fp32'][1]:.3f} | same status "
          f"{int((a['status'] == b['status']).sum())}/{A} | iterations sum {int(a['iters'].sum())} vs {int(b['iters'].sum())} "
          f"| max |dx| on solved {np.abs(a['cpts'][ok] - b['cpts'][ok]).max():.2e}")
    P.replan(pva, sw.goals, t_start, sw.dev["ego_ids"], sw.new, sw.ok)
    sw.own = driver.merge_latest(sw.new, sw.own, sw.ok)
    driver.exchange_records(sw.own, sw.all, sw.dist, sw.world)
    sw.tick += 1
