"""CPU: A* oracle behaviours that pin the reference's quirks."""
import numpy as np

from helpers import hard_cases, oracle_grids


def test_libm_and_detmath_give_identical_expansions(pop, orc):
    spec, ap = pop.config.make_spec("parity"), pop.config.make_astar_params()
    sc, pva = hard_cases(pop, 8, 17)
    recs = pop.scene.straight_records(sc)
    grids = oracle_grids(pop, orc, spec, sc, recs)
    total = 0
    for a in range(8):
        w = orc.astar_search(spec, ap, grids[a], sc["poses"][a], pva[a], sc["goals"][a], 0.05, 0.3)
        orc.astar_use_libm(1)
        w2 = orc.astar_search(spec, ap, grids[a], sc["poses"][a], pva[a], sc["goals"][a], 0.05, 0.3)
        orc.astar_use_libm(0)
        assert w["ret"] == w2["ret"] and np.array_equal(w["trace"], w2["trace"])
        assert np.array_equal(w["route"], w2["route"])
        total += w["stats"][1]
        # route is sampled every corridor_tau = 0.3 s over <= max_tau + one step
        assert 2 <= len(w["route"]) <= 9
    assert total > 50


def test_blocked_start_returns_no_path(pop, orc):
    spec, ap = pop.config.make_spec("parity"), pop.config.make_astar_params()
    V = spec.L * spec.W * spec.H
    g = np.ones((V, spec.T), np.float32)  # everything occupied
    pose = np.zeros(3, np.float32)
    pva = np.zeros(9)
    pva[:3] = (0, 0, 1)
    w = orc.astar_search(spec, ap, g, pose, pva, np.array([5.0, 0, 1]), 0.05, 0.3)
    assert w["ret"] == 0 and len(w["route"]) == 0 and w["stats"][3] == 2  # retried with init=false


def test_free_space_reaches_time_horizon(pop, orc):
    spec, ap = pop.config.make_spec("parity"), pop.config.make_astar_params()
    V = spec.L * spec.W * spec.H
    g = np.zeros((V, spec.T), np.float32)
    pose = np.zeros(3, np.float32)
    pva = np.zeros(9)
    pva[:3] = (0, 0, 1)
    w = orc.astar_search(spec, ap, g, pose, pva, np.array([9.0, 0, 1]), 0.05, 0.3)
    assert w["ret"] == 3  # REACH_HORIZON (time >= max_tau)
    assert len(w["route"]) == 8 and np.allclose(w["route"][0, :3], pva[:3])
    # near the goal: the one-shot trajectory succeeds -> REACH_END
    w = orc.astar_search(spec, ap, g, pose, pva, np.array([0.1, 0, 1]), 0.05, 0.3)
    assert w["ret"] == 4
