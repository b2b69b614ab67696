"""CPU: Seidel LP vs scipy (HiGHS), FIRI / MVIE geometric properties, corridor stage rules."""
import numpy as np
import pytest

from helpers import hard_cases, oracle_grids


def test_linprog_matches_highs(orc):
    from scipy.optimize import linprog as sp
    rng = np.random.default_rng(0)
    n_inf = 0
    for trial in range(400):
        d = 3 if trial % 2 == 0 else 4
        m = int(rng.integers(d + 1, 60))
        ctr = rng.uniform(-5, 5, d)
        A = rng.normal(size=(m, d))
        A /= np.linalg.norm(A, axis=1, keepdims=True)
        off = rng.uniform(-0.3 if trial % 5 == 0 else 0.1, 2.0, m)
        b = A @ ctr + off
        A2 = np.concatenate([A, np.eye(d), -np.eye(d)])
        b2 = np.concatenate([b, ctr + 8, -(ctr - 8)])
        c = rng.normal(size=d) if trial % 3 else np.zeros(d)
        v, x = orc.linprog(c, A2, b2)
        r = sp(c, A_ub=A2, b_ub=b2, bounds=[(None, None)] * d, method="highs")
        if r.status == 2:
            n_inf += 1
            assert v == np.inf
        else:
            assert np.isfinite(v) and abs(v - r.fun) < 1e-7 * (1 + abs(r.fun))
            assert (A2 @ x - b2).max() < 1e-8
    assert n_inf > 10


def test_linprog_degenerate_cases(orc):
    v, x = orc.linprog([1, 0, 0], [[0, 1, 0]], [1.0])
    assert v == -np.inf                                    # unbounded (sdlp.hpp:776-780 semantics)
    v, x = orc.linprog([0, 0, 0], np.zeros((0, 3)), np.zeros(0))
    assert v == 0.0                                        # no constraints, zero objective (:720-724)
    v, x = orc.linprog([0, 0, 1], np.zeros((0, 3)), np.zeros(0))
    assert v == -np.inf


def test_firi_polytope_contains_seed_and_excludes_points(orc):
    bd = np.array([[1, 0, 0, -3], [0, 1, 0, -3], [0, 0, 1, -3], [-1, 0, 0, -1], [0, -1, 0, -3], [0, 0, -1, 0.0]])
    rng = np.random.default_rng(0)
    pc = rng.uniform([-1, -3, 0], [3, 3, 3], (300, 3))
    a, b = np.array([0, 0, 1.5]), np.array([1.0, 0.2, 1.5])
    pc = pc[np.linalg.norm(pc - a, axis=1) > 0.6]
    hp, n, r = orc.firi(bd, pc, a, b, 2)
    assert n == len(hp) and n >= 6
    assert (hp[:, :3] @ a + hp[:, 3]).max() < 0 and (hp[:, :3] @ b + hp[:, 3]).max() < 0
    inside = (hp[:, :3] @ pc.T + hp[:, 3:4]).max(axis=0) < -1e-9
    assert not inside.any()
    # seed outside the bounding box -> firi returns false (firi.hpp:250-252)
    hp2, n2, _ = orc.firi(bd, pc, np.array([10.0, 0, 0]), b, 2)
    assert n2 == -1
    # no obstacle points -> the box itself
    hp3, n3, _ = orc.firi(bd, np.zeros((0, 3)), a, b, 2)
    assert n3 == 6


def test_mvie_ellipsoid_is_inscribed_and_large(orc):
    bd = np.array([[1, 0, 0, -2], [0, 1, 0, -1], [0, 0, 1, -0.5], [-1, 0, 0, -2], [0, -1, 0, -1], [0, 0, -1, -0.5]])
    ok, R, p, r = orc.mvie(bd, np.eye(3), np.zeros(3), np.ones(3) * 0.1)
    assert ok and abs(np.linalg.det(R) - 1) < 1e-9
    # the MVIE of a box is the axis-aligned ellipsoid with the box half-sizes
    assert np.allclose(sorted(r), [0.5, 1.0, 2.0], rtol=2e-3)
    assert np.abs(p).max() < 5e-3


def test_corridor_stage_rules(pop, orc):
    spec = pop.config.make_spec("parity")
    ap, pp = pop.config.make_astar_params(), pop.config.make_planner_params(True)
    sc, pva = hard_cases(pop, 6, 99)
    recs = pop.scene.straight_records(sc)
    grids = oracle_grids(pop, orc, spec, sc, recs)
    seen_nonbox = False
    for a in range(6):
        w = orc.astar_search(spec, ap, grids[a], sc["poses"][a], pva[a], sc["goals"][a], 0.05, 0.3)
        c = orc.corridor_generate(spec, pp, grids[a], sc["poses"][a], sc["stamps"][a], pva[a],
                                  sc["stamps"][a] + 0.05, w["route"])
        n = c["npoly"]
        assert 0 <= n <= len(w["route"]) - 1
        for i in range(n):
            h = c["polys"][i, :c["nfaces"][i]]
            seen_nonbox |= len(h) > 6
            # consecutive corridors intersect (checked by an LP in the stage): feasibility LP again
            if i + 1 < n:
                both = np.concatenate([h, c["polys"][i + 1, :c["nfaces"][i + 1]]])
                v, _ = orc.linprog([0, 0, 0], both[:, :3], -both[:, 3])
                assert np.isfinite(v)
        if n:
            # local goal = route point n-1 unless projected (baseline_fake.cpp:400-414)
            assert np.isfinite(c["goal"]).all()
    assert seen_nonbox
    # a route of a single point produces no corridor
    c = orc.corridor_generate(spec, pp, grids[0], sc["poses"][0], sc["stamps"][0], pva[0],
                              sc["stamps"][0] + 0.05, np.concatenate([pva[0, :6]])[None])
    assert c["npoly"] == 0
