"""Perception-in-the-loop replan (BASELINE configs[1] style): depth cloud -> filterPointCloud -> DSPMap::update ->
RiskVoxel::publishMap (+ set-to-1 neighbour overlay) -> BaselinePlanner::replan with the RiskVoxel query rules
(K = 125 kernel, fixed thresholds) and the non-fake corridor rules.  The planning half is compared with the oracle
run on the SAME published grid (downloaded from the GPU), so A* / corridors are bit-exact and the QP within 1e-4."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_dsp_publish_then_replan_matches_oracle(pop, orc):
    import torch
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    dsp = importlib.import_module("pred-occ-planner_amd.dsp")
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    A = 4
    spec = pop.config.make_spec("parity", map_kind=pop._abi.SOGM_MAP_RISKVOXEL)
    spec.risk_threshold_region = 0.2  # map/risk_threshold_region default of RiskVoxel (risk_voxel.cpp:23)
    m = sogm.SogmMap(spec, A)
    g = dsp.DspMap(m, dsp.make_dsp_params(spec.T), dsp.make_tables(21, n_gauss=1 << 18, n_rand=1 << 12))
    cap = 5000
    # the camera looks along +x: the wall / pillar / floor of make_depth_cloud lie ahead of every agent
    clouds = [pop.scene.make_depth_cloud(60 + a) for a in range(A)]
    n_pix = len(clouds[0])
    raw = sogm._dev(np.concatenate(clouds, axis=0), np.float32)
    rng = sogm._dev(np.stack([np.arange(A) * n_pix, (np.arange(A) + 1) * n_pix], axis=1), np.int32)
    labels = torch.zeros((A * cap, 4), dtype=torch.float32, device="cuda")
    base = torch.arange(A, dtype=torch.int32, device="cuda") * cap
    quat = sogm._dev(np.tile(np.float32([1, 0, 0, 0]), (A, 1)), np.float32)
    starts = np.array([[0.0, 0.6 * a - 0.9, 1.0] for a in range(A)])
    for k in range(8):
        pos = sogm._dev(starts.astype(np.float32))
        stamps = sogm._dev(np.full(A, 50.0 + k / 30.0), np.float64)
        pts, cnt = m.filterPointCloud(raw, rng, 0.15, cap)
        g.update(pts.view(-1, 3), labels, torch.stack([base, base + cnt], dim=1).contiguous(), pos, quat, stamps)
    g.publish()
    # neighbours: straight trajectories crossing in front of the agents (set-to-1 overlay of RiskVoxel)
    sc = {"n_agents": A, "starts": starts, "goals": starts + np.array([3.5, 0.0, 0.0]),
          "stamps": np.full(A, 50.0 + 7 / 30.0), "ego_ids": np.arange(A, dtype=np.int32)}
    recs = pop.scene.straight_records(sc, speed=1.0)
    ego = sogm._dev(sc["ego_ids"], np.int32)
    m.addOtherAgents(sogm._dev(recs), A, ego)
    grids = [m.download(a) for a in range(A)]
    assert max(float(x.max()) for x in grids) > spec.risk_threshold  # the map is not empty
    ap, pp, qs = pop.config.make_astar_params(), pop.config.make_planner_params(False), pop.config.make_qp_settings()
    P = planner.SogmPlanner(m, ap, pp, qs)
    pva = np.concatenate([starts, np.zeros((A, 6))], axis=1)
    goals = starts + np.array([2.5, 0.4, 0.0])
    t_start = sc["stamps"] + 0.02
    rec_d, ok_d = P.replan(sogm._dev(pva, np.float64), sogm._dev(goals, np.float64), sogm._dev(t_start, np.float64), ego)
    got = planner.records_from_bytes(rec_d.cpu().numpy())
    ok = ok_d.cpu().numpy()
    n_ok = 0
    for a in range(A):
        w_ok, w, stage = orc.replan(spec, ap, pp, qs, grids[a], starts[a].astype(np.float32), float(sc["stamps"][a]),
                                    pva[a], goals[a], t_start[a], a)
        assert ok[a] == w_ok, (a, ok[a], w_ok, stage)
        assert got[a].n_pieces == w.n_pieces
        if w_ok:
            n_ok += 1
            k = w.n_pieces
            assert np.allclose(np.array(got[a].cpts[:15 * k]), np.array(w.cpts[:15 * k]), atol=TOL, rtol=0)
    print("perception replans ok:", n_ok, "of", A)
    P.close()
    g.close()
    m.close()
