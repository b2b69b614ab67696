#!/usr/bin/env python
"""Are the QPs that run to max_iter infeasible?  ("Settle the 4000-iteration QPs".)

  dump     (GPU box)  fly the bench scene for N ticks with the staged entry points and save every QP's inputs
                      (start state, local goal, polytopes) with the status / iteration count the HIP solver gave
                      -> gpurun_out/qp_dump.npz
  analyze  (CPU)      for every dumped QP: a HiGHS feasibility LP on {l <= A x <= u} (scipy), and the CPU oracle's
                      OSQP restatement on the same QP -> a table (feasible / infeasible x status x iterations)

    python tools/diag_qp_infeasible.py dump 23
    python tools/diag_qp_infeasible.py analyze gpurun_out/qp_dump.npz
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dump(ticks, out):
    import torch
    pop = importlib.import_module("pred-occ-planner_amd")
    driver = importlib.import_module("pred-occ-planner_amd.driver")
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    sw = driver.SwarmTick("cfg2", 128, overlap_clear=False)
    P, A = sw.planner, sw.A_loc
    rows = {k: [] for k in ("tick", "agent", "pva", "goal", "polys", "nfaces", "npoly", "status", "iters")}
    for _ in range(ticks):
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
        q = P.optimize(pva, c["goal"], c["polys"], c["nfaces"], c["npoly"])
        cn = {k: v.cpu().numpy() for k, v in c.items()}
        qn = {k: v.cpu().numpy() for k, v in q.items()}
        pv = pva.cpu().numpy()
        for a in range(A):
            if cn["npoly"][a] <= 0:
                continue
            rows["tick"].append(sw.tick)
            rows["agent"].append(a)
            rows["pva"].append(pv[a])
            rows["goal"].append(cn["goal"][a])
            rows["polys"].append(cn["polys"][a])
            rows["nfaces"].append(cn["nfaces"][a])
            rows["npoly"].append(cn["npoly"][a])
            rows["status"].append(qn["status"][a])
            rows["iters"].append(qn["iters"][a])
        P.replan(pva, sw.goals, t_start, sw.dev["ego_ids"], sw.new, sw.ok)
        sw.own = driver.merge_latest(sw.new, sw.own, sw.ok)
        driver.exchange_records(sw.own, sw.all, sw.dist, sw.world)
        sw.tick += 1
    np.savez_compressed(out, **{k: np.asarray(v) for k, v in rows.items()},
                        max_faces=P.pp.max_faces, tau=P.pp.corridor_tau, vmax=P.pp.opt_max_vel, amax=P.pp.opt_max_acc)
    st = np.asarray(rows["status"])
    print(json.dumps({"qps": int(len(st)), "status": {int(k): int((st == k).sum()) for k in np.unique(st)}}))
    sw.close()


def feasible(A, l, u):
    """HiGHS feasibility LP on {l <= A x <= u} (bounds beyond 1e20 = none).  Returns (feasible?, max violation of
    the point found)."""
    from scipy.optimize import linprog
    from scipy.sparse import csr_matrix, vstack
    fin_u, fin_l = u < 1e20, l > -1e20
    eq = fin_u & fin_l & (u - l < 1e-12)
    iu, il = fin_u & ~eq, fin_l & ~eq
    As = csr_matrix(A)
    Aub = vstack([As[iu], -As[il]])
    bub = np.concatenate([u[iu], -l[il]])
    r = linprog(np.zeros(A.shape[1]), A_ub=Aub, b_ub=bub, A_eq=As[eq], b_eq=u[eq], bounds=(None, None),
                method="highs", options={"primal_feasibility_tolerance": 1e-9})
    if r.status == 0:
        ax = A @ r.x
        return True, float(max((ax - u)[fin_u].max(), (l - ax)[fin_l].max(), 0.0))
    return (False if r.status == 2 else None), float("nan")


def analyze(path, limit=None, which="fail"):
    pop = importlib.import_module("pred-occ-planner_amd")
    orc = importlib.import_module("oracle.binding")
    orc.lib()
    d = np.load(path)
    qs = pop.config.make_qp_settings()
    MF, tau, vmax, amax = int(d["max_faces"]), float(d["tau"]), float(d["vmax"]), float(d["amax"])
    st = d["status"]
    sel = np.nonzero(~np.isin(st, (1, 2)))[0] if which == "fail" else np.arange(len(st))
    if limit:
        sel = sel[:limit]
    table = {}
    for i in sel:
        M = int(d["npoly"][i])
        goal = np.concatenate([d["goal"][i], np.zeros(3)])
        Q, A, l, u = orc.qp_assemble(d["pva"][i], goal, [tau] * M, d["polys"][i], d["nfaces"][i], MF, vmax, amax,
                                     m_cap=16384)
        feas, viol = feasible(A, l, u)
        s2, x, it = orc.qp_solve(d["pva"][i], goal, [tau] * M, d["polys"][i], d["nfaces"][i], MF, vmax, amax, qs)
        key = (("feasible" if feas else "infeasible") if feas is not None else "highs?", int(st[i]), int(s2))
        e = table.setdefault(key, {"n": 0, "gpu_iters": [], "oracle_iters": []})
        e["n"] += 1
        e["gpu_iters"].append(int(d["iters"][i]))
        e["oracle_iters"].append(int(it))
    print("HiGHS | status (dumped) | status (oracle now) | count | iterations dumped (median/max) | oracle now (median/max)")
    for k in sorted(table):
        e = table[k]
        print(f"{k[0]:10s} | {k[1]:3d} | {k[2]:3d} | {e['n']:4d} | {int(np.median(e['gpu_iters']))}/{max(e['gpu_iters'])}"
              f" | {int(np.median(e['oracle_iters']))}/{max(e['oracle_iters'])}")
    return table


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        dump(int(sys.argv[2]) if len(sys.argv) > 2 else 23, os.path.join(ROOT, "gpurun_out", "qp_dump.npz"))
    else:
        analyze(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else None, sys.argv[4] if len(sys.argv) > 4 else "fail")
