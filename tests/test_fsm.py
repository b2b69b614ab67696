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
def test_fsm_closed_loop_runs(pop):
    driver = importlib.import_module("pred-occ-planner_amd.driver")
    sw = driver.SwarmTick("parity", 8, fsm=True)
    states = []
    for _ in range(25):
        sw.step()
        states.append(sw.status.cpu().numpy().copy())
    states = np.stack(states)
    assert (states == driver.FSM_EXEC_TRAJ).any() and (states == driver.FSM_REPLAN).any()
    sw.close()
