"""FiniteStateMachine bookkeeping (row f2): the tensorised state update of driver.fsm_apply against the per-agent
restatement of the C++ switch in oracle/fsm_oracle.cpp (plan_manager/src/plan_manager.cpp:92-233); hover
records; isTrajSafe."""
import importlib

import numpy as np
import pytest
import torch


def test_fsm_apply_matches_switch(pop, orc):
    d = importlib.import_module("pred-occ-planner_amd.driver")
    rng = np.random.default_rng(3)
    A, ticks = 64, 120
    status = torch.full((A,), d.FSM_NEW_PLAN, dtype=torch.int32)
    fail = torch.zeros(A, dtype=torch.int32)
    success = torch.zeros(A, dtype=torch.bool)
    traj_start = torch.full((A,), 98.0, dtype=torch.float64)
    ref = [orc.FsmOracle(98.0, d.TICK_PERIOD, d.REPLAN_START_TIME, d.REPLAN_MAX_FAILURES) for _ in range(A)]
    p_ok = rng.uniform(0.05, 0.95, A)  # some agents fail almost always -> hover records, NEW_PLAN retries
    seen = set()
    for k in range(ticks):
        now_f = 100.0 + 0.1 * k
        now = torch.full((A,), now_f, dtype=torch.float64)
        ok = torch.from_numpy(rng.random(A) < p_ok)
        safe = torch.from_numpy(rng.random(A) < 0.9)
        reached = torch.from_numpy(rng.random(A) < 0.002)
        due_new, is_rep, t_start = d.fsm_plan_inputs(status, traj_start, now)
        status, fail, traj_start, success, pub_new, pub_hover, hover_start = d.fsm_apply(
            status, fail, traj_start, success, now, due_new, is_rep, ok & (due_new | is_rep), safe, reached)
        for a in range(A):
            pub = ref[a].tick(now_f, bool(ok[a]), bool(safe[a]), bool(reached[a]))
            assert int(status[a]) == ref[a].s.status, (k, a)
            assert int(fail[a]) == ref[a].s.num_replan_failures, (k, a)
            assert float(traj_start[a]) == ref[a].s.traj_start_time, (k, a)
            assert bool(pub_new[a]) == (pub == "new")
            assert bool(pub_hover[a]) == (isinstance(pub, tuple))
            if isinstance(pub, tuple):
                assert float(hover_start[a]) == pub[1]
            seen.add(ref[a].s.status)
            if isinstance(pub, tuple):
                seen.add("hover")
    assert {d.FSM_NEW_PLAN, d.FSM_EXEC_TRAJ, d.FSM_REPLAN, d.FSM_GOAL_REACHED, "hover"} <= seen


def test_hover_record_layout(pop):
    d = importlib.import_module("pred-occ-planner_amd.driver")
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    ids = torch.tensor([3, 7], dtype=torch.int32)
    pos = torch.tensor([[1.0, 2.0, 3.0], [-4.0, 5.5, 0.25]], dtype=torch.float64)
    rec = planner.records_from_bytes(d.hover_records(ids, torch.tensor([10.5, 11.0], dtype=torch.float64), pos).numpy())
    for i in range(2):
        r = rec[i]
        assert (r.drone_id, r.n_pieces, r.time_start, r.duration[0]) == (int(ids[i]), 1, [10.5, 11.0][i], 0.5)
        assert np.allclose(np.asarray(r.cpts[:15]).reshape(5, 3), pos[i].numpy())
        assert not np.asarray(r.cpts[15:]).any() and not np.asarray(r.duration[1:]).any()


@pytest.mark.gpu
def test_traj_safe_gpu_matches_oracle(pop, orc):
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    A = 24
    spec = pop.config.make_spec("parity")
    sc = pop.scene.make_scene(A, 4.95, seed=12, circle_radius=3.0, n_cyl=150)
    dev = sogm.upload_scene(sc)
    m = sogm.SogmMap(spec, A)
    m.updateMap(dev["cloud"], dev["cloud_range"], dev["cylinders"], dev["n_cyl"], dev["poses"], dev["stamps"])
    out = dict(sc)
    out["goals"] = sc["starts"] * np.array([3.0, 3.0, 1.0])  # fly outwards, through the obstacle ring
    recs = pop.scene.straight_records(out, speed=1.5)
    recs[5].n_pieces = 0
    t0 = float(sc["stamps"][0])
    rng = np.random.default_rng(0)
    t_now = t0 + rng.uniform(-0.1, 1.9, A)
    got = m.isTrajSafe(sogm._dev(recs), sogm._dev(t_now, np.float64), 3.0).cpu().numpy()
    want = np.array([orc.traj_safe(spec, m.download(a), sc["poses"][a], float(sc["stamps"][a]), recs[a], t_now[a], 3.0)
                     for a in range(A)])
    assert np.array_equal(got, want), (got, want)
    assert 0 < want.sum() < A
    m.close()


@pytest.mark.gpu
def test_fsm_closed_loop_flight_matches_the_switch(pop, orc):
    """25 ticks of SwarmTick(fsm=True) — every agent runs the reference's FiniteStateMachine around the HIP replan —
    against oracle/fsm_oracle.cpp (the C++ switch of plan_manager.cpp:92-233) fed the flight's own replan / isTrajSafe
    / goal results: state, failure counter, traj_start_time_ and the publication of EVERY tick must agree, a
    published trajectory must be this tick's plan from the state's start time (:110-135 now, :165-175 now + 0.02),
    a hover record the 0.5 s piece of publishEmptyTrajectory (:404-424) at the agent's position."""
    driver = importlib.import_module("pred-occ-planner_amd.driver")
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    A = 8
    sw = driver.SwarmTick("parity", A, fsm=True)
    ref = [orc.FsmOracle(float(sw.traj_start[a]), driver.TICK_PERIOD, driver.REPLAN_START_TIME,
                         driver.REPLAN_MAX_FAILURES) for a in range(A)]
    seen, n_new, n_hover = set(), 0, 0
    for k in range(25):
        before = [r.s.status for r in ref]
        own_before = sw.own.cpu().numpy().copy()
        sw.step()
        f = {key: (v.cpu().numpy() if hasattr(v, "cpu") else v) for key, v in sw.last_fsm.items()}
        status, fail, ts = sw.status.cpu().numpy(), sw.fail.cpu().numpy(), sw.traj_start.cpu().numpy()
        own = planner.records_from_bytes(sw.own.cpu().numpy())
        new = planner.records_from_bytes(sw.new.cpu().numpy())
        for a in range(A):
            pub = ref[a].tick(f["now"], bool(f["ok"][a]), bool(f["safe"][a]), bool(f["reached"][a]))
            assert (int(status[a]), int(fail[a]), float(ts[a])) == \
                (ref[a].s.status, ref[a].s.num_replan_failures, ref[a].s.traj_start_time), (k, a)
            seen.add(ref[a].s.status)
            if pub == "new":
                n_new += 1
                assert bool(f["pub_new"][a]) and own[a].n_pieces == new[a].n_pieces > 0
                assert bytes(own[a]) == bytes(new[a])
                want_start = f["now"] + (driver.REPLAN_START_TIME if before[a] == driver.FSM_REPLAN else 0.0)
                assert own[a].time_start == want_start == float(f["t_start"][a])
            elif isinstance(pub, tuple):
                n_hover += 1
                assert bool(f["pub_hover"][a])
                assert (own[a].n_pieces, own[a].duration[0], own[a].time_start) == (1, 0.5, pub[1])
                assert np.array_equal(np.asarray(own[a].cpts[:15]).reshape(5, 3), np.tile(f["pos"][a], (5, 1)))
            else:
                assert not f["pub_new"][a] and not f["pub_hover"][a]
                assert sw.own[a].cpu().numpy().tobytes() == own_before[a].tobytes()   # keeps executing its trajectory
    assert {driver.FSM_EXEC_TRAJ, driver.FSM_REPLAN} <= seen and n_new >= 20
    sw.close()
