"""ParticleATC::isSafeAfterOpt (row f3): separating-plane LP, oracle on CPU, HIP vs oracle on the GPU."""
import importlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_separator_fixture(orc):
    fx = json.load(open(os.path.join(HERE, "golden", "separator_fixture.json")))
    assert orc.separable(fx["pointsA"], fx["pointsB"]) == fx["separable"]
    # swapping the sets cannot change separability; a point of B inside hull(A) breaks it
    assert orc.separable(fx["pointsB"], fx["pointsA"]) == 1
    inside = np.mean(np.asarray(fx["pointsA"]), axis=0)
    assert orc.separable(fx["pointsA"], fx["pointsB"] + [inside.tolist()]) == 0


def test_separable_random_against_scipy(orc):
    from scipy.optimize import linprog
    rng = np.random.default_rng(2)
    n_sep = 0
    for k in range(60):
        A = rng.normal(0, 1, (rng.integers(4, 40), 3))
        B = rng.normal(0, 1, (rng.integers(4, 40), 3)) + rng.uniform(0, 4) * np.array([1.0, 0.3, 0.0])
        rows = np.concatenate([np.hstack([-A, -np.ones((len(A), 1))]), np.hstack([B, np.ones((len(B), 1))])])
        res = linprog(np.zeros(4), A_ub=rows, b_ub=-np.ones(len(rows)), bounds=[(None, None)] * 4, method="highs")
        assert orc.separable(A, B) == int(res.status == 0), k
        n_sep += int(res.status == 0)
    assert 5 < n_sep < 55


def _crossing_swarm(pop, A=8):
    sc = pop.scene.make_scene(A, 4.95, seed=5, circle_radius=3.0, n_cyl=0)
    recs = pop.scene.straight_records(sc, speed=1.5, n_pieces=6, piece_dur=0.5)
    return sc, recs


def test_oracle_crossing_trajectories_are_unsafe(pop, orc):
    sc, recs = _crossing_swarm(pop)
    t_now = float(sc["stamps"][0])
    # antipodal goals on a circle: every straight path passes through the centre -> hulls intersect
    c0 = np.asarray(recs[0].cpts[:15 * 6])
    assert orc.safe_after_opt(c0, 6, recs, 8, 0, t_now) == 0
    # alone in the swarm (or everyone else not yet started) it is safe
    assert orc.safe_after_opt(c0, 6, recs, 1, 0, t_now) == 1
    late = pop.scene.straight_records(sc, speed=1.5, n_pieces=6, piece_dur=0.5, t_start=t_now + 1.0)
    assert orc.safe_after_opt(c0, 6, late, 8, 0, t_now) == 1
    # once the others are on their last piece only 5 control points remain: a short stub far away
    assert orc.safe_after_opt(c0, 6, recs, 8, 0, t_now + 2.8) in (0, 1)


@pytest.mark.gpu
def test_safe_after_opt_gpu_matches_oracle(pop, orc):
    import torch
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    A = 12
    spec = pop.config.make_spec("parity")
    sc = pop.scene.make_scene(A, 4.95, seed=9, circle_radius=3.0, n_cyl=0)
    rng = np.random.default_rng(4)
    recs = pop.scene.straight_records(sc, speed=1.2, n_pieces=6, piece_dur=0.4)
    t0 = float(sc["stamps"][0])
    for a in range(A):  # mixed phases: running, not started, ended, missing
        recs[a].time_start = t0 + [-0.05, -0.9, -1.7, 0.5, -5.0][a % 5]
    recs[7].n_pieces = 0
    m = sogm.SogmMap(spec, A)
    P = planner.SogmPlanner(m, pop.config.make_astar_params(), pop.config.make_planner_params(True),
                            pop.config.make_qp_settings())
    # candidate trajectories: short straight stubs with random headings from each start
    cpts = np.zeros((A, 16 * 15))
    npoly = np.zeros(A, np.int32)
    for a in range(A):
        M = int(rng.integers(1, 8)) if a != 3 else 0
        npoly[a] = M
        d = rng.normal(0, 1, 3) * np.array([1, 1, 0.1])
        d /= np.linalg.norm(d)
        for k in range(5 * M):
            cpts[a, k * 3:(k + 1) * 3] = sc["starts"][a] + d * 0.25 * k + rng.normal(0, 0.02, 3)
    t_now = np.full(A, t0) + rng.uniform(0, 0.3, A)
    got = P.isSafeAfterOpt(sogm._dev(cpts, np.float64), sogm._dev(npoly, np.int32), sogm._dev(recs), A,
                           sogm._dev(sc["ego_ids"], np.int32), sogm._dev(t_now, np.float64)).cpu().numpy()
    want = np.array([orc.safe_after_opt(cpts[a], int(npoly[a]), recs, A, a, t_now[a]) if npoly[a] > 0 else 1
                     for a in range(A)])
    assert np.array_equal(got, want), (got, want)
    assert 0 < want.sum() < A
    P.close()
    m.close()


@pytest.mark.gpu
def test_replan_with_swarm_applies_deconfliction(pop, orc):
    import torch
    from helpers import hard_cases
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    A = 8
    spec = pop.config.make_spec("parity")
    sc, pva = hard_cases(pop, A, 17)
    recs = pop.scene.straight_records(sc)
    dev = sogm.upload_scene(sc)
    m = sogm.SogmMap(spec, A)
    m.updateMap(dev["cloud"], dev["cloud_range"], dev["cylinders"], dev["n_cyl"], dev["poses"], dev["stamps"])
    m.addOtherAgents(sogm._dev(recs), A, dev["ego_ids"])
    P = planner.SogmPlanner(m, pop.config.make_astar_params(), pop.config.make_planner_params(True),
                            pop.config.make_qp_settings())
    t_start = sc["stamps"] + 0.05
    d_pva, d_ts, d_goal = sogm._dev(pva, np.float64), sogm._dev(t_start, np.float64), sogm._dev(sc["goals"], np.float64)
    rec0, ok0 = P.replan(d_pva, d_goal, d_ts, dev["ego_ids"])
    rec0 = planner.records_from_bytes(rec0.cpu().numpy())
    ok0 = ok0.cpu().numpy()
    d_recs, d_now = sogm._dev(recs), sogm._dev(sc["stamps"], np.float64)
    P.setSwarm(d_recs, A, dev["ego_ids"], d_now)
    rec1, ok1 = P.replan(d_pva, d_goal, d_ts, dev["ego_ids"])
    ok1 = ok1.cpu().numpy()
    want = np.array([int(ok0[a] and orc.safe_after_opt(np.asarray(rec0[a].cpts[:15 * rec0[a].n_pieces]),
                                                       rec0[a].n_pieces, recs, A, a, float(sc["stamps"][a])))
                     for a in range(A)])
    assert np.array_equal(ok1, want), (ok0, ok1, want)
    P.setSwarm(None, 0, None, None)
    _, ok2 = P.replan(d_pva, d_goal, d_ts, dev["ego_ids"])
    assert np.array_equal(ok2.cpu().numpy(), ok0)
    P.close()
    m.close()


@pytest.mark.gpu
def test_safe_after_opt_random_lp_cases(pop, orc):
    """Random overlapping / nearly touching control-point clouds: the bounding-box and candidate-normal
    pre-tests rarely decide, so the wave-parallel Seidel LP runs; flags must equal the oracle's."""
    import torch
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    A = 48
    spec = pop.config.make_spec("parity")
    rng = np.random.default_rng(77)
    SogmTrajRecord = pop._abi.SogmTrajRecord
    recs = (SogmTrajRecord * A)()
    cpts = np.zeros((A, 16 * 15))
    npoly = np.zeros(A, np.int32)
    centre = rng.uniform(-1, 1, (A, 3))
    for a in range(A):
        M = int(rng.integers(2, 9))
        npoly[a] = M
        # candidate: an elongated random cloud through the origin region
        R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        pts = (rng.normal(size=(5 * M, 3)) * np.array([1.5, 0.15, 0.15])) @ R.T + centre[a] * 0.3
        cpts[a, :15 * M] = pts.reshape(-1)
        r = recs[a]
        r.drone_id, r.n_pieces, r.time_start = a, int(rng.integers(2, 9)), 99.0
        Rb = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        off = rng.normal(size=3) * rng.choice([0.05, 0.3, 0.8])
        pb = (rng.normal(size=(5 * r.n_pieces, 3)) * np.array([1.5, 0.15, 0.15])) @ Rb.T + off
        for k in range(r.n_pieces):
            r.duration[k] = 0.4
        for k in range(15 * r.n_pieces):
            r.cpts[k] = float(pb.reshape(-1)[k])
    t_now = np.full(A, 99.5)
    m = sogm.SogmMap(spec, A)
    P = planner.SogmPlanner(m, pop.config.make_astar_params(), pop.config.make_planner_params(True),
                            pop.config.make_qp_settings())
    ego = np.arange(A, dtype=np.int32)
    got = P.isSafeAfterOpt(sogm._dev(cpts, np.float64), sogm._dev(npoly, np.int32), sogm._dev(recs), A,
                           sogm._dev(ego, np.int32), sogm._dev(t_now, np.float64)).cpu().numpy()
    want = np.array([orc.safe_after_opt(cpts[a], int(npoly[a]), recs, A, a, t_now[a]) for a in range(A)])
    assert np.array_equal(got, want), (got, want)
    # pairwise too (one record at a time) so that a single wrong LP cannot hide behind another "unsafe"
    n_sep = 0
    for b in range(0, A, 3):
        one = (SogmTrajRecord * 1)(recs[b])
        g1 = P.isSafeAfterOpt(sogm._dev(cpts, np.float64), sogm._dev(npoly, np.int32), sogm._dev(one), 1,
                              sogm._dev(ego, np.int32), sogm._dev(t_now, np.float64)).cpu().numpy()
        w1 = np.array([orc.safe_after_opt(cpts[a], int(npoly[a]), one, 1, a, t_now[a]) for a in range(A)])
        assert np.array_equal(g1, w1), (b, g1, w1)
        n_sep += int(w1.sum())
    assert 0 < n_sep < A * len(range(0, A, 3))
    P.close()
    m.close()
