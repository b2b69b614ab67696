"""Pins the oracle's Bezier evaluation against the reference's own known-answer tests
(traj_utils/test/test_bernstein.cpp:58-108), re-expressed from tests/golden/bernstein_kat.json."""
import json
import os

import numpy as np

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bernstein_kat.json")))


def test_piece_endpoints_exact(orc):
    p = G["piece"]
    c = np.array(p["cpts"], float)
    # EXPECT_EQ in the reference => exact equality
    assert (orc.piece_eval(c, p["t0"], p["tf"], 2.0, 0) == np.array(p["pos_at_2"])).all()
    assert (orc.piece_eval(c, p["t0"], p["tf"], 2.0, 1) == np.array(p["vel_at_2"])).all()
    assert (orc.piece_eval(c, p["t0"], p["tf"], 2.0, 2) == np.array(p["acc_at_2"])).all()
    assert (orc.piece_eval(c, p["t0"], p["tf"], 4.0, 0) == np.array(p["pos_at_4"])).all()
    assert p["tf"] - p["t0"] == p["duration"]


def test_coeff_matrix_and_derivative_ctrl_pts(orc):
    p = G["piece"]
    assert (orc.bernstein_coeff() == np.array(p["coeff"], float)).all()
    v = orc.derivative_ctrl_pts(np.array(p["cpts"], float))
    assert (v == np.array(p["vel_cpts"], float)).all()
    a = orc.derivative_ctrl_pts(v)
    assert (a == np.array(p["acc_cpts"], float)).all()


def test_bezier_duration_pieces_and_max_rate(orc):
    b = G["bezier"]
    d = np.array(b["durations"], float)
    c = np.array(b["cpts"], float)
    assert d.sum() == b["total_duration"] and len(d) == b["n_pieces"]
    assert orc.bezier_max_rate(d, c, 1) > b["max_vel_rate_gt"]
    # the reference's TestTraj loop only prints; here: piecewise continuity at the C0 knots
    for knot in (1.0, 3.0):
        lo = orc.bezier_eval(d, c, knot - 1e-9)
        hi = orc.bezier_eval(d, c, knot + 1e-9)
        assert np.abs(lo - hi).max() < 1e-6
    # end point of the last piece
    assert np.allclose(orc.bezier_eval(d, c, 6.0 - 1e-12), [8, 6, 4], atol=1e-9)


def test_locate_piece_past_end_uses_last_piece(orc):
    b = G["bezier"]
    d = np.array(b["durations"], float)
    c = np.array(b["cpts"], float)
    # locatePiece returns M-1 beyond the end (bernstein.hpp:171): extrapolates the last piece
    p = orc.bezier_eval(d, c, 6.5)
    q = orc.piece_eval(c[10:15], 3.0, 6.0, 6.5, 0)
    assert (p == q).all()
