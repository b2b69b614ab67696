"""Full-chain parity AT THE BASELINE GRIDS: the replan chain (SOGM build + overlay, A* pop order, corridor polytopes,
QP status / iterations / coefficients) of the HIP path against the CPU oracle on the bench scenes themselves, over
several ticks of the closed loop:
  cfg0  40 x 40 x 20 x 10, 1 agent, static pillars (BASELINE configs[0], the CPU-runnable anchor)
  cfg1  100^3 x 15, 16 agents, perception pipeline with the RiskVoxel rules (configs[1])
  cfg2  200^3 x 20, the 128-agent bench scene, 8 of its agents checked (configs[2], the headline workload)
  cfg4  300^3 x 30 with fp16 occupancy cells, 2 agents (configs[4])
Voxel values, A* expansions, polytopes: bit-exact.  Bezier coefficients: 1e-4 (north_star), same status/iterations.
The oracle sums costMVIE in the reference's order and solves LPs with the sdlp restatement (no kernel-shaped modes)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


class _Inputs:
    """What the oracle's updateMap gets for agent `a` at tick `tick`: the sensor frame of that tick (world-frame flights:
    the whole cloud — the oracle crops it itself, fake_particle_risk_voxel.cpp:88-104) or the frozen scene's host-side
    crop (legacy flights)."""

    def __init__(self, pop, sw, tick):
        c = sw.compute
        self.world = getattr(c, "use_world", False)
        if self.world:
            f = c.timeline.frame(tick)
            self.cloud, self.cyl, self.n_cyl = f["cloud"], pop.scene.cylinders_to_struct(f["cylinders"]), len(f["cylinders"])
        else:
            self.cloud, self.crange = sw.dev["cloud"].cpu().numpy(), sw.dev["cloud_range"].cpu().numpy()
            self.cyl, self.n_cyl = pop.scene.cylinders_to_struct(sw.scene["cylinders"]), len(sw.scene["cylinders"])

    def cloud_of(self, a):
        return self.cloud if self.world else self.cloud[self.crange[a, 0]:self.crange[a, 1]]


def _tick_with_parity(pop, orc, sw, agents, check_grid_cells=True):
    """One tick of SwarmTick.step() spelled out, with every stage of `agents` compared against the oracle.
    Returns a dict of counts."""
    import torch
    driver = importlib.import_module("pred-occ-planner_amd.driver")
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    spec, P, A = sw.spec, sw.planner, sw.A_loc
    ap, pp, qs = P.ap, P.pp, P.qs
    stamp = sw.t0 + sw.tick * driver.TICK_PERIOD
    stamps = torch.full((A,), stamp, dtype=torch.float64, device="cuda")
    sw.now.copy_(stamps)
    t_start = stamps + driver.REPLAN_START_TIME
    pva, valid = planner.traj_eval(sw.own, t_start)
    pva = torch.where(valid.bool().unsqueeze(1), pva, sw.hover).contiguous()
    sw.hover = torch.cat([pva[:, :3], torch.zeros_like(pva[:, 3:])], dim=1)
    poses = pva[:, :3].to(torch.float32).contiguous()
    if getattr(sw.compute, "use_world", False):  # this tick's sensor frame, cropped on the device
        sw.map.updateWorld(sw.compute.world(sw.tick), poses, stamps)
    else:
        sw.map.updateMap(sw.dev["cloud"], sw.dev["cloud_range"], sw.dev["cylinders"], sw.dev["n_cyl"], poses, stamps)
    sw.map.addOtherAgents(sw.all, sw.A_tot, sw.dev["ego_ids"])
    s = P.search(pva, sw.goals, t_start, route_cap=64, trace_cap=12000)
    c = P.generateCorridors(pva, t_start, s["route"], s["route_len"])
    q = P.optimize(pva, c["goal"], c["polys"], c["nfaces"], c["npoly"])
    sn = {k: v.cpu().numpy() for k, v in s.items()}
    cn = {k: v.cpu().numpy() for k, v in c.items()}
    qn = {k: v.cpu().numpy() for k, v in q.items()}
    pv, ps = pva.cpu().numpy(), poses.cpu().numpy()
    ts = t_start.cpu().numpy()
    goals = sw.goals.cpu().numpy()
    recs = planner.records_from_bytes(sw.all.cpu().numpy())
    inp = _Inputs(pop, sw, sw.tick)
    ego = sw.dev["ego_ids"].cpu().numpy()
    stats = {"agents": 0, "expansions": 0, "polys": 0, "qp_ok": 0, "worst_dx": 0.0}
    for a in agents:
        g = orc.update_gt(spec, inp.cloud_of(a), inp.cyl, inp.n_cyl, ps[a])
        orc.project_neighbours(spec, g, recs, sw.A_tot, int(ego[a]), sw.map.body, ps[a], stamp)
        if check_grid_cells:
            got = sw.map.download(a)
            assert np.array_equal(got, g), f"tick {sw.tick} agent {a}: SOGM differs in {(got != g).sum()} cells"
            del got
        w = orc.astar_search(spec, ap, g, ps[a], pv[a], goals[a], float(ts[a] - stamp), pp.corridor_tau,
                             trace_cap=12000)
        assert sn["ret"][a] == w["ret"], (sw.tick, a, sn["ret"][a], w["ret"])
        assert list(sn["stats"][a]) == w["stats"], (sw.tick, a, sn["stats"][a], w["stats"])
        k = min(w["trace_len"], 12000)
        assert np.array_equal(sn["trace"][a, :k], w["trace"][:k]), f"tick {sw.tick} agent {a}: A* pop order differs"
        n = len(w["route"])
        assert sn["route_len"][a] == n and np.array_equal(sn["route"][a, :n], w["route"])
        stats["expansions"] += w["stats"][1]
        cc = orc.corridor_generate(spec, pp, g, ps[a], stamp, pv[a], float(ts[a]), w["route"])
        del g
        M = cc["npoly"]
        assert cn["npoly"][a] == M, (sw.tick, a, cn["npoly"][a], M)
        assert np.array_equal(cn["nfaces"][a][:max(M, 0)], cc["nfaces"][:max(M, 0)])
        for i in range(max(M, 0)):
            nf = cc["nfaces"][i]
            assert np.array_equal(cn["polys"][a, i, :nf], cc["polys"][i, :nf]), \
                f"tick {sw.tick} agent {a} polytope {i}: max |d| {np.abs(cn['polys'][a, i, :nf] - cc['polys'][i, :nf]).max()}"
            stats["polys"] += 1
        if M > 0:
            assert np.array_equal(cn["goal"][a], cc["goal"])
            goal = np.concatenate([cc["goal"], np.zeros(3)])
            st, x, it = orc.qp_solve(pv[a], goal, [pp.corridor_tau] * M, cc["polys"], cc["nfaces"], pp.max_faces,
                                     pp.opt_max_vel, pp.opt_max_acc, qs)
            assert qn["status"][a] == st and qn["iters"][a] == it, (sw.tick, a, qn["status"][a], st, qn["iters"][a], it)
            if st in (1, 2):
                d = float(np.abs(qn["cpts"][a, :15 * M] - x).max())
                assert d <= TOL, (sw.tick, a, d)
                stats["worst_dx"] = max(stats["worst_dx"], d)
                stats["qp_ok"] += 1
        stats["agents"] += 1
    # advance the closed loop exactly like SwarmTick.step() — and hold the fused sogm_replan (dataflow kernels) to
    # the staged entry points just checked against the oracle: same ok flag, same record, bit for bit
    safe = P.isSafeAfterOpt(q["cpts"], c["npoly"], sw.all, sw.A_tot, sw.dev["ego_ids"], sw.now).cpu().numpy()
    P.replan(pva, sw.goals, t_start, sw.dev["ego_ids"], sw.new, sw.ok)
    new = planner.records_from_bytes(sw.new.cpu().numpy())
    okf = sw.ok.cpu().numpy()
    stats["fused_checked"] = 0
    for a in agents:
        M = int(cn["npoly"][a])
        want_ok = bool(sn["ret"][a] != 0 and M > 0 and qn["status"][a] in (1, 2) and safe[a] != 0)
        assert bool(okf[a]) == want_ok, (sw.tick, a, okf[a], sn["ret"][a], M, qn["status"][a], safe[a])
        assert new[a].n_pieces == (M if want_ok else 0) and new[a].drone_id == int(ego[a])
        if want_ok:
            assert np.array_equal(np.array(new[a].cpts[:15 * M]), qn["cpts"][a, :15 * M]), (sw.tick, a)
            assert list(new[a].duration[:M]) == [pp.corridor_tau] * M and new[a].time_start == float(ts[a])
        stats["fused_checked"] += 1
    sw.own = driver.merge_latest(sw.new, sw.own, sw.ok)
    if getattr(sw, "neighbour_lag", 1) == 2:  # the flight's staleness rule: the next tick reads the table of ONE tick ago
        import torch
        prev = getattr(sw, "_lag_prev", None)
        sw._lag_prev, sw.all = sw.own.clone(), (prev if prev is not None else torch.zeros_like(sw.all))
    else:
        driver.exchange_records(sw.own, sw.all, sw.dist, sw.world)
    sw.tick += 1
    return stats


def _step_with_parity(pop, orc, sw, agents, cell_agents, expect):
    """One tick of the BENCH's own path — SwarmTick.step(): grid taken from the pool (reset through its mark log by
    k_reset_sectors, stamped inside the previous replan by k_prestamp_flow, adopted by sogm_update_prestamped), the
    dataflow sogm_replan — and then, on the map that tick planned on (still current after the step):
      * `expect` asserted against sogm_grid_history (which path built the grid),
      * every cell of `cell_agents`' grids against the oracle's build from zero
        (fake_particle_risk_voxel.cpp:107-108 fill + marks + overlay),
      * for every agent of `agents`: A* return / node counts / pop order / route, polytopes (bit-exact), QP status /
        iterations / coefficients (1e-4) of the staged entry points run on that map against the oracle run on ITS map,
      * the tick's own published record (sogm_replan's output) against those stage results, bit for bit."""
    import torch
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    spec, P = sw.spec, sw.planner
    ap, pp, qs = P.ap, P.pp, P.qs
    all_before = sw.all.clone()  # the table this tick's overlay and deconfliction read
    tick = sw.tick
    sw.step()
    torch.cuda.synchronize()
    assert sw.planner.flow_failures() == (0, 0), (tick, sw.planner.flow_failures())
    hist = sw.map.grid_history()
    for k, v in expect.items():
        if k.startswith("min_"):
            assert hist[k[4:]] >= v, (tick, hist, expect)
        else:
            assert hist[k] == v, (tick, hist, expect)
    pva, t_start, poses = sw.pva, sw.t_start, sw.poses  # this tick's inputs (written by the previous replan's pre-stamp)
    pv, ps, ts, now = pva.cpu().numpy(), poses.cpu().numpy(), t_start.cpu().numpy(), sw.now.cpu().numpy()
    # the map centres the pre-stamp filed with the grid are the ones the host sees
    for a in cell_agents:
        mt, mc = sw.map.map_state(a)
        assert mt == float(now[a]) and np.array_equal(mc, ps[a]), (tick, a, mt, now[a], mc, ps[a])
    s = P.search(pva, sw.goals, t_start, route_cap=64, trace_cap=12000)
    c = P.generateCorridors(pva, t_start, s["route"], s["route_len"])
    q = P.optimize(pva, c["goal"], c["polys"], c["nfaces"], c["npoly"])
    safe = P.isSafeAfterOpt(q["cpts"], c["npoly"], all_before, sw.A_tot, sw.dev["ego_ids"], sw.now).cpu().numpy()
    sn = {k: v.cpu().numpy() for k, v in s.items()}
    cn = {k: v.cpu().numpy() for k, v in c.items()}
    qn = {k: v.cpu().numpy() for k, v in q.items()}
    goals = sw.goals.cpu().numpy()
    recs = planner.records_from_bytes(all_before.cpu().numpy())
    new = planner.records_from_bytes(sw.new.cpu().numpy())
    okf = sw.ok.cpu().numpy()
    # the frame the map was built from: this tick's — or, for a grid the previous replan pre-stamped, the newest frame
    # that existed while that replan ran, i.e. the previous tick's (map_input_staleness_ticks = 1)
    inp = _Inputs(pop, sw, tick - (sw.map_input_staleness_ticks if hist["prestamped"] else 0))
    ego = sw.dev["ego_ids"].cpu().numpy()
    stats = {"agents": 0, "cells": 0, "expansions": 0, "polys": 0, "qp_ok": 0, "worst_dx": 0.0, "fused_ok": 0}
    for a in agents:
        stamp = float(now[a])
        g = orc.update_gt(spec, inp.cloud_of(a), inp.cyl, inp.n_cyl, ps[a])
        orc.project_neighbours(spec, g, recs, sw.A_tot, int(ego[a]), sw.map.body, ps[a], stamp)
        if a in cell_agents:
            got = sw.map.download(a)
            assert np.array_equal(got, g), f"tick {tick} agent {a}: SOGM differs in {(got != g).sum()} cells"
            assert int((g != 0).sum()) > 100  # (a populated map, not two empty ones)
            stats["cells"] += g.size
            del got
        w = orc.astar_search(spec, ap, g, ps[a], pv[a], goals[a], float(ts[a] - stamp), pp.corridor_tau, trace_cap=12000)
        assert sn["ret"][a] == w["ret"], (tick, a, sn["ret"][a], w["ret"])
        assert list(sn["stats"][a]) == w["stats"], (tick, a, sn["stats"][a], w["stats"])
        k = min(w["trace_len"], 12000)
        assert np.array_equal(sn["trace"][a, :k], w["trace"][:k]), f"tick {tick} agent {a}: A* pop order differs"
        n = len(w["route"])
        assert sn["route_len"][a] == n and np.array_equal(sn["route"][a, :n], w["route"])
        stats["expansions"] += w["stats"][1]
        cc = orc.corridor_generate(spec, pp, g, ps[a], stamp, pv[a], float(ts[a]), w["route"])
        del g
        M = cc["npoly"]
        assert cn["npoly"][a] == M, (tick, a, cn["npoly"][a], M)
        assert np.array_equal(cn["nfaces"][a][:max(M, 0)], cc["nfaces"][:max(M, 0)])
        for i in range(max(M, 0)):
            nf = cc["nfaces"][i]
            assert np.array_equal(cn["polys"][a, i, :nf], cc["polys"][i, :nf]), (tick, a, i)
            stats["polys"] += 1
        st = 0
        if M > 0:
            assert np.array_equal(cn["goal"][a], cc["goal"])
            goal = np.concatenate([cc["goal"], np.zeros(3)])
            st, x, it = orc.qp_solve(pv[a], goal, [pp.corridor_tau] * M, cc["polys"], cc["nfaces"], pp.max_faces,
                                     pp.opt_max_vel, pp.opt_max_acc, qs)
            assert qn["status"][a] == st and qn["iters"][a] == it, (tick, a, qn["status"][a], st, qn["iters"][a], it)
            if st in (1, 2):
                d = float(np.abs(qn["cpts"][a, :15 * M] - x).max())
                assert d <= TOL, (tick, a, d)
                stats["worst_dx"] = max(stats["worst_dx"], d)
                stats["qp_ok"] += 1
        # the tick's own output (dataflow sogm_replan on the same map): the stage results, bit for bit
        want_ok = bool(sn["ret"][a] != 0 and M > 0 and st in (1, 2) and safe[a] != 0)
        assert bool(okf[a]) == want_ok, (tick, a, okf[a], sn["ret"][a], M, st, safe[a])
        assert new[a].n_pieces == (M if want_ok else 0) and new[a].drone_id == int(ego[a])
        if want_ok:
            assert np.array_equal(np.array(new[a].cpts[:15 * M]), qn["cpts"][a, :15 * M]), (tick, a)
            assert new[a].time_start == float(ts[a])
            stats["fused_ok"] += 1
        stats["agents"] += 1
    return stats


def _sum(acc, s):
    for k, v in s.items():
        acc[k] = max(acc.get(k, 0.0), v) if k == "worst_dx" else acc.get(k, 0) + v
    return acc


def test_cfg2_bench_path_moving_world_on_reset_grids(pop, orc):
    """BASELINE configs[2], the path bench.py's headline times, at the bench's size: 128 agents, 200^3 x 20, a MOVING
    world (every tick's update takes that tick's sensor frame: cylinders advanced by v dt, their cloud points with them,
    cropped on the device around the agent's current map centre), the default pool of three grids, sparse reset, map
    update at the start of the tick (map_input_staleness_ticks = 0), publication inside the replan.  The pool's slots are
    first adopted after a reset through their logs at ticks 3, 4 and 5: those three ticks are checked — every cell of two
    agents' 640 MB grids against the oracle's build from zero of THAT tick's frame (a different pair per tick: all three
    slots, six agents), and the A* trace / polytopes / QP / published record of eight agents.  The grid content changes
    from tick to tick (asserted: the oracle's grids of consecutive frames differ for a fixed pose)."""
    driver = importlib.import_module("pred-occ-planner_amd.driver")
    sw = driver.SwarmTick("cfg2", 128, moving_world=True)
    assert sw.overlap_mode == 3 and not sw.prestamp and sw.publish and sw.map.sparse_reset_state()["enabled"]
    assert sw.map_input_staleness_ticks == 0 and sw.compute.timeline.moving
    agents = [0, 17, 33, 50, 64, 81, 99, 127]
    for _ in range(3):
        sw.step()
    acc, slots = {}, set()
    for k, cells in enumerate(([0, 81], [33, 127], [17, 64])):
        st = _step_with_parity(pop, orc, sw, agents, cells, {"min_sparse_resets": 1, "dense_clears": 1, "prestamped": False})
        slots.add(sw.map.grid_history()["slot"])
        _sum(acc, st)
    print("cfg2 bench path (moving world), ticks 3-5:", acc)
    assert slots == {0, 1, 2}
    assert acc["agents"] == 24 and acc["cells"] == 6 * 160_000_000 and acc["expansions"] > 100 and acc["polys"] > 50
    assert acc["qp_ok"] >= 16 and acc["fused_ok"] >= 14
    # the world really moved between the checked ticks: same pose, consecutive frames, different maps
    spec, tl = sw.spec, sw.compute.timeline
    pose = sw.poses[0].cpu().numpy()
    f3, f5 = tl.frame(3), tl.frame(5)
    g3 = orc.update_gt(spec, f3["cloud"], pop.scene.cylinders_to_struct(f3["cylinders"]), len(f3["cylinders"]), pose)
    g5 = orc.update_gt(spec, f5["cloud"], pop.scene.cylinders_to_struct(f5["cylinders"]), len(f5["cylinders"]), pose)
    assert int((g3 != g5).sum()) > 1000
    sw.close()


def test_cfg2_prestamped_variant_plans_on_a_one_tick_old_frame(pop, orc):
    """The pre-stamped variant on the moving world (bench.py's labelled variant, map_input_staleness_ticks = 1): the replan
    of tick k builds tick k + 1's map from frame k — the newest frame a live host could hand it.  On ticks whose grid was
    reset through its log AND pre-stamped, every cell of two agents' grids equals the oracle's build of the PREVIOUS
    tick's frame around the CURRENT pose, and the chain of eight agents matches the oracle on that map."""
    driver = importlib.import_module("pred-occ-planner_amd.driver")
    sw = driver.SwarmTick("cfg2", 128, moving_world=True, prestamp=True)
    assert sw.overlap_mode == 3 and sw.prestamp and sw.map_input_staleness_ticks == 1
    agents = [0, 17, 33, 50, 64, 81, 99, 127]
    for _ in range(3):
        sw.step()
    acc = {}
    for cells in ([0, 81], [33, 127]):
        _sum(acc, _step_with_parity(pop, orc, sw, agents, cells, {"min_sparse_resets": 1, "prestamped": True}))
    print("cfg2 pre-stamped variant (moving world), ticks 3-4:", acc)
    assert acc["agents"] == 16 and acc["cells"] == 4 * 160_000_000 and acc["qp_ok"] >= 10
    sw.close()


def test_cfg2_bench_scene_chain_parity(pop, orc):
    """BASELINE configs[2]: the bench's own 128-agent scene at 200^3 x 20 through the STAGED entry points (sogm_update_gt
    + sogm_project_neighbours: densely cleared grids, no pool), 3 ticks; 8 agents spread over the swarm are checked
    stage by stage.  (The bench's own path — pooled, sparse-reset, pre-stamped grids — is the test above.)"""
    driver = importlib.import_module("pred-occ-planner_amd.driver")
    sw = driver.SwarmTick("cfg2", 128, moving_world=True)
    agents = [0, 17, 33, 50, 64, 81, 99, 127]
    acc = {}
    for _ in range(3):
        _sum(acc, _tick_with_parity(pop, orc, sw, agents, check_grid_cells=(sw.tick in (0, 2))))
    print("cfg2:", acc)
    assert acc["agents"] == 24 and acc["expansions"] > 100 and acc["polys"] > 50 and acc["qp_ok"] >= 16
    assert acc["fused_checked"] == 24
    sw.close()


def test_cfg2_fused_tick_all_agents_against_oracle_replans(pop, orc):
    """The configuration the bench runs — SwarmTick.step() itself: 128 agents, pooled grids with the side-stream
    clear, the dataflow sogm_replan (persistent corridor / QP / finish kernels, speculative second search) — checked
    against the CPU oracle's full replan (search + corridors + QP + isSafeAfterOpt) for 16 agents spread over the
    swarm, on the FIFTH tick of the flight (tick index 4: agents and obstacles moving, neighbours' records in the overlay,
    and the map a pool slot that has been reset through its mark log and stamped from the tick's own sensor frame)."""
    import torch
    driver = importlib.import_module("pred-occ-planner_amd.driver")
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    sw = driver.SwarmTick("cfg2", 128, moving_world=True)
    for _ in range(4):
        sw.step()
    all_before = sw.all.clone()
    sw.step()
    torch.cuda.synchronize()
    hist = sw.map.grid_history()
    assert hist["sparse_resets"] >= 1 and not hist["prestamped"], hist
    if sw.planner.flow_failures() != (0, 0):  # say where the dataflow stopped
        import ctypes as C
        buf = np.zeros(11 + 6 * 128, np.int32)
        pop.lib().sogm_debug_flow_peek(sw.planner._p, buf.ctypes.data_as(C.c_void_p), buf.size)
        raise AssertionError((sw.planner.flow_failures(), sw.overlap_mode, sw.prestamp, buf[:11].tolist()))
    spec, P = sw.spec, sw.planner
    pv, ps = sw.pva.cpu().numpy(), sw.poses.cpu().numpy()
    ts, now = sw.t_start.cpu().numpy(), sw.now.cpu().numpy()
    goals = sw.goals.cpu().numpy()
    recs = planner.records_from_bytes(all_before.cpu().numpy())
    new = planner.records_from_bytes(sw.new.cpu().numpy())
    okf = sw.ok.cpu().numpy()
    inp = _Inputs(pop, sw, sw.tick - 1)
    ego = sw.dev["ego_ids"].cpu().numpy()
    n_ok, worst = 0, 0.0
    for a in range(3, 128, 8):  # 16 agents
        g = orc.update_gt(spec, inp.cloud_of(a), inp.cyl, inp.n_cyl, ps[a])
        orc.project_neighbours(spec, g, recs, sw.A_tot, int(ego[a]), sw.map.body, ps[a], float(now[a]))
        ok, rec, _ = orc.replan(spec, P.ap, P.pp, P.qs, g, ps[a], float(now[a]), pv[a], goals[a], float(ts[a]), int(ego[a]))
        del g
        if ok:
            ok = orc.safe_after_opt(np.asarray(rec.cpts[:15 * rec.n_pieces]), rec.n_pieces, recs, sw.A_tot, int(ego[a]),
                                    float(now[a]))
        assert bool(okf[a]) == bool(ok), (a, okf[a], ok)
        if ok:
            M = rec.n_pieces
            assert new[a].n_pieces == M
            d = float(np.abs(np.array(new[a].cpts[:15 * M]) - np.array(rec.cpts[:15 * M])).max())
            assert d <= TOL, (a, d)
            worst, n_ok = max(worst, d), n_ok + 1
        else:
            assert new[a].n_pieces == 0
    print("cfg2 fused vs oracle: ok", n_ok, "of 16, worst |dx|", worst)
    assert n_ok >= 10
    sw.close()


@pytest.mark.parametrize("seed,min_ok", [(0x5069, 4), (0x5071, 4), (0x5067, 0)])
def test_cfg0_static_pillars_chain_parity(pop, orc, seed, min_ok):
    """BASELINE configs[0]: 1 agent, 40 x 40 x 20 x 10 SOGM, static pillar map; 6 ticks of the closed loop.
    Seed 0x5067 is the case where the shrunk corridor excludes a waypoint and OSQP reports primal infeasibility
    (status -3) on every tick: the failure path must agree as well (same status, same iteration count)."""
    driver = importlib.import_module("pred-occ-planner_amd.driver")
    spec = pop.config.make_spec("cfg0")
    half = (spec.L // 2) * 0.15
    scene = pop.scene.make_scene(1, half, seed=seed, moving=False)
    sw = driver.SwarmTick("cfg0", 1, spec=spec, scene=scene)
    acc = {}
    for _ in range(6):
        _sum(acc, _tick_with_parity(pop, orc, sw, [0]))
    print("cfg0:", acc)
    assert acc["agents"] == 6 and acc["expansions"] > 10 and acc["qp_ok"] >= min_ok
    sw.close()


def test_cfg4_fp16_chain_parity(pop, orc):
    """BASELINE configs[4]: 300^3 x 30 with fp16 occupancy cells, 2 agents (the oracle's fp32 grid is 3.2 GB per
    agent; marks and neighbour counts are exact in fp16, so cells compare bit for bit).  Two ticks through the staged
    entry points, then the flight's own path — SwarmTick.step() — for five ticks: ticks 3 and 4 plan on pool slots
    reset through their logs (16-cell fp16 sectors) and stamped by the previous replan; both are checked cell by cell
    and stage by stage."""
    driver = importlib.import_module("pred-occ-planner_amd.driver")
    spec = pop.config.make_spec("cfg4")
    assert spec.storage & 1 == pop._abi.SOGM_STORE_F16
    sw = driver.SwarmTick("cfg4", 2, spec=spec)
    acc = {}
    for _ in range(2):
        _sum(acc, _tick_with_parity(pop, orc, sw, [0, 1], check_grid_cells=(sw.tick == 0)))
    print("cfg4:", acc)
    assert acc["agents"] == 4 and acc["polys"] > 8 and acc["qp_ok"] >= 2
    sw.close()
    sw = driver.SwarmTick("cfg4", 2, spec=spec)
    assert sw.overlap_mode == 3 and sw.prestamp and sw.map.sparse_reset_state()["enabled"]
    for _ in range(3):
        sw.step()
    acc = {}
    for cells in ([0], [1]):
        _sum(acc, _step_with_parity(pop, orc, sw, [0, 1], cells, {"min_sparse_resets": 1, "prestamped": True}))
    print("cfg4 flight path, ticks 3-4:", acc)
    assert acc["agents"] == 4 and acc["cells"] == 2 * 27_000_000 * 30 and acc["polys"] > 8 and acc["qp_ok"] >= 2
    sw.close()


def test_cfg1_perception_riskvoxel_chain_parity(pop, orc):
    """BASELINE configs[1]: 16 agents, 100^3 x 15, 640 x 480 depth clouds at 30 Hz through filterPointCloud ->
    DSPMap::update -> RiskVoxel::publishMap (+ set-to-1 overlay), then BaselinePlanner::replan with the RiskVoxel
    query rules and the non-fake A* / corridor rules.  Particle store of agent 0 against the oracle's DSP (bit-exact
    slots), planning of 4 agents against the oracle run on the published grids."""
    import torch
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    dsp = importlib.import_module("pred-occ-planner_amd.dsp")
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    A = 16
    spec = pop.config.make_spec("cfg1", map_kind=pop._abi.SOGM_MAP_RISKVOXEL)
    spec.risk_threshold_region = 0.2  # RiskVoxel's map/risk_threshold_region default (risk_voxel.cpp:23)
    m = sogm.SogmMap(spec, A)
    Pd = dsp.make_dsp_params(spec.T)
    tabs = dsp.make_tables(21, n_gauss=1 << 18, n_rand=1 << 12)
    g = dsp.DspMap(m, Pd, tabs)
    o0 = orc.DspOracle(spec, Pd, tabs)
    cap = 5000
    clouds = [pop.scene.make_depth_cloud(80 + a) for a in range(A)]
    n_pix = len(clouds[0])
    raw = sogm._dev(np.concatenate(clouds, axis=0), np.float32)
    rng = sogm._dev(np.stack([np.arange(A) * n_pix, (np.arange(A) + 1) * n_pix], axis=1), np.int32)
    base = torch.arange(A, dtype=torch.int32, device="cuda") * cap
    quat = sogm._dev(np.tile(np.float32([1, 0, 0, 0]), (A, 1)), np.float32)
    starts = np.array([[0.0, 0.6 * a - 4.5, 1.0] for a in range(A)])
    n_upd = 4
    for k in range(n_upd):
        pos = sogm._dev(starts.astype(np.float32))
        stamps = sogm._dev(np.full(A, 50.0 + k / 30.0), np.float64)
        pts, cnt = m.filterPointCloud(raw, rng, 0.15, cap)
        # labels = None: velocityEstimationThread (clustering + association) runs on the GPU / in the oracle
        g.update(pts.view(-1, 3), None, torch.stack([base, base + cnt], dim=1).contiguous(), pos, quat, stamps)
        n0 = int(cnt[0].item())
        p0 = pts[0, :n0].cpu().numpy()
        o0.update(p0, None, starts[0].astype(np.float32), np.float32([1, 0, 0, 0]), 50.0 + k / 30.0)
        gb, gc = g.download_born(0)
        wb, wc = o0.born()
        assert gc[3] == 0 and gc[:3] == wc and np.array_equal(gb, wb), (k, gc, wc)
    st_g = g.download_state(0)
    st_o = o0.state()
    gs, ws = st_g[0], st_o[0]  # [V][slots][9]: flag, payload; slot 8 (update time) is not stored on the GPU
    assert np.array_equal(gs[:, :, 0], ws[:, :, 0]), "slot flags of agent 0 differ from the oracle's"
    live = ws[:, :, 0] > 0.1
    assert live.sum() > 1000  # the map holds particles
    for f in range(1, 8):
        assert np.array_equal(gs[:, :, f][live], ws[:, :, f][live]), f"particle field {f} differs"
    assert st_g[2][10] == 0 and st_g[2][11] == 0, f"device error counters {st_g[2]}"
    g.publish()
    sc = {"n_agents": A, "starts": starts, "goals": starts + np.array([3.5, 0.0, 0.0]),
          "stamps": np.full(A, 50.0 + (n_upd - 1) / 30.0), "ego_ids": np.arange(A, dtype=np.int32)}
    recs = pop.scene.straight_records(sc, speed=1.0)
    ego = sogm._dev(sc["ego_ids"], np.int32)
    m.addOtherAgents(sogm._dev(recs), A, ego)
    ap, pp, qs = pop.config.make_astar_params(False), pop.config.make_planner_params(False), pop.config.make_qp_settings()
    P = planner.SogmPlanner(m, ap, pp, qs)
    pva = np.concatenate([starts, np.zeros((A, 6))], axis=1)
    goals = starts + np.array([2.5, 0.4, 0.0])
    t_start = sc["stamps"] + 0.02
    rec_d, ok_d = P.replan(sogm._dev(pva, np.float64), sogm._dev(goals, np.float64), sogm._dev(t_start, np.float64), ego)
    got = planner.records_from_bytes(rec_d.cpu().numpy())
    ok = ok_d.cpu().numpy()
    n_ok = 0
    for a in (0, 5, 10, 15):
        grid = m.download(a)
        assert float(grid.max()) > spec.risk_threshold
        w_ok, w, stage = orc.replan(spec, ap, pp, qs, grid, starts[a].astype(np.float32), float(sc["stamps"][a]),
                                    pva[a], goals[a], t_start[a], a)
        assert ok[a] == w_ok, (a, ok[a], w_ok, stage)
        assert got[a].n_pieces == w.n_pieces
        if w_ok:
            n_ok += 1
            k = w.n_pieces
            assert np.allclose(np.array(got[a].cpts[:15 * k]), np.array(w.cpts[:15 * k]), atol=TOL, rtol=0)
    print("cfg1 perception replans ok:", n_ok, "of 4 checked")
    assert n_ok >= 2
    P.close(); g.close(); o0.close(); m.close()
