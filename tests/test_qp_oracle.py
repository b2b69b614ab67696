"""CPU: pins the oracle's BezierOpt assembly + OSQP-algorithm solve against the reference's own
test fixture and tolerances (traj_opt/test/test_bezier_opt.cpp), and cross-checks optimality
against an independent tight solve (KKT conditions verified with numpy)."""
import copy
import json
import os

import numpy as np

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bezier_opt_fixture.json")))
MF = 8


def _polys(cubes):
    p = np.zeros((len(cubes), MF, 4))
    for i, c in enumerate(cubes):
        p[i, :6] = np.array(c, float)
    return p


def test_shapes_match_reference_test(pop, orc):
    s = G["single"]
    Q, A, l, u = orc.qp_assemble(s["start"], s["end"], s["t"], _polys([s["cube"]]), [6], MF, 3.0, 3.0)
    assert list(Q.shape) == s["Q_shape"]                      # TestMinJerkCost :130-135
    assert np.allclose(Q, Q.T) and np.linalg.eigvalsh(Q).min() > -1e-9
    t = G["three"]
    Q, A, l, u = orc.qp_assemble(t["start"], t["end"], t["t"], _polys(t["cubes"]), [6, 6, 6], MF, 3.0, 3.0)
    assert A.shape[1] == t["n_vars"] and len(u) == A.shape[0]  # TestOpt :137-142
    # 9(M+1) continuity + 21M dynamic + 5*sum(F) safety rows (bezier_optimizer.cpp:113-126)
    assert A.shape[0] == 9 * 4 + 21 * 3 + 5 * 18
    assert (l[:36] == u[:36]).all() and (l[-90:] <= -1e29).all()


def test_three_cube_corridor_solves_with_reference_tolerances(pop, orc):
    t = G["three"]
    qs = pop.config.make_qp_settings()
    st, x, it = orc.qp_solve(t["start"], t["end"], t["t"], _polys(t["cubes"]), [6, 6, 6], MF,
                             t["default_vmax"], t["default_amax"], qs)
    assert st in (1, 2)                                        # optimize() true  (:146-147)
    X = x.reshape(t["x_rows"], t["x_cols"])                   # :151-153
    d = np.array(t["t"], float)
    T = d.sum()
    start, end = np.array(t["start"], float), np.array(t["end"], float)
    tol = t["bc_tol"]
    for der in range(3):                                      # TestWaypoints :156-185
        assert np.abs(orc.bezier_eval(d, X, 0.0, der) - start[der]).max() < tol
        assert np.abs(orc.bezier_eval(d, X, T, der) - end[der]).max() < tol
    a, b = t["knot_probe"]                                    # C1 / C2 at the knot :192-199
    assert np.linalg.norm(orc.bezier_eval(d, X, a, 1) - orc.bezier_eval(d, X, b, 1)) < 2e-2 + tol
    for tt in (1.0, 2.0):                                     # TestContinuity :215-220
        fd = np.linalg.norm(orc.bezier_eval(d, X, tt, 1) - orc.bezier_eval(d, X, tt - 0.1, 1)) / 0.1
        av = np.linalg.norm(orc.bezier_eval(d, X, tt, 2) + orc.bezier_eval(d, X, tt - 0.1, 2)) / 2
        assert abs(fd - av) <= t["fd_tol"] + 1e-2
    # all control points inside their cubes (the safety rows) up to the solver tolerance
    for i, c in enumerate(t["cubes"]):
        c = np.array(c, float)
        assert (c[:, :3] @ X[i * 5:(i + 1) * 5].T + c[:, 3:4]).max() < 5e-3


def test_tight_solve_satisfies_kkt_and_bounds_the_1e3_solution(pop, orc):
    t = G["three"]
    Q, A, l, u = orc.qp_assemble(t["start"], t["end"], t["t"], _polys(t["cubes"]), [6, 6, 6], MF, 3.0, 3.0)
    qt = pop.config.make_qp_settings()
    qt.eps_abs = qt.eps_rel = 1e-9
    qt.max_iter = 400000
    st, xs, ys, it = orc.osqp_dense(Q, np.zeros(45), A, l, u, qt)
    assert st == 1
    assert np.abs(Q @ xs + A.T @ ys).max() < 1e-6                       # stationarity
    Ax = A @ xs
    assert (Ax - u).max() < 1e-6 and (l - Ax).max() < 1e-6             # primal feasibility
    assert (ys[Ax < u - 1e-5] <= 1e-6).all() and (ys[Ax > l + 1e-5] >= -1e-6).all()  # compl. slackness
    qs = pop.config.make_qp_settings()
    st1, x1, _ = orc.qp_solve(t["start"], t["end"], t["t"], _polys(t["cubes"]), [6, 6, 6], MF, 3.0, 3.0, qs)
    # documented: the reference's eps = 1e-3 stop leaves the coefficients ~1e-2 from the optimum
    assert np.abs(x1 - xs).max() < 0.1
    assert 0.5 * x1 @ Q @ x1 <= 0.5 * xs @ Q @ xs * 1.2 + 1e-6


def test_fixed_rho_vs_adaptive(pop, orc):
    """The reference's own 3-cube fixture under OSQP's defaults: with the rho estimate computed from the SCALED
    residuals (auxil.c compute_rho_estimate) it never leaves the factor-5 band around rho = 0.1, so adaptive rho and
    the fixed KKT factor the north_star asks for are the same solve here — max_iter 4000 is reached with status 2
    (SOLVED_INACCURATE: optimize() still returns true, bezier_optimizer.cpp:280-283, and the solution meets the
    test's 1e-3 boundary tolerances, see above); with more iterations both reach eps = 1e-3."""
    t = G["three"]
    out = {}
    for name, interval in (("fixed", 0), ("adaptive", 25)):
        for max_iter in (4000, 8000):
            qs = pop.config.make_qp_settings()
            qs.adaptive_rho_interval, qs.max_iter = interval, max_iter
            st, _, it = orc.qp_solve(t["start"], t["end"], t["t"], _polys(t["cubes"]), [6, 6, 6], MF, 3.0, 3.0, qs)
            out[name, max_iter] = (st, it)
    assert out["fixed", 4000] == out["adaptive", 4000] == (2, 4000)
    assert out["fixed", 8000] == out["adaptive", 8000] and out["fixed", 8000][0] == 1 and 4000 < out["fixed", 8000][1] < 6000


def test_infeasible_qp_is_reported(pop, orc):
    s = G["single"]
    cube = np.array(s["cube"], float)
    start = np.array(s["start"], float)
    start[0] = [10, 10, 10]  # start position outside the corridor -> primal infeasible
    qs = pop.config.make_qp_settings()
    st, x, it = orc.qp_solve(start, s["end"], s["t"], _polys([cube]), [6], MF, 3.0, 3.0, qs)
    assert st not in (1,)
