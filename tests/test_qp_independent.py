"""The C++ oracle's BezierOpt assembly (oracle/qp_oracle.cpp, `orc_qp_assemble`) against an INDEPENDENT numpy restatement
written from the reference text (tests/golden/make_qp_fixture.py -> qp_assembly_independent.json; traj_opt/src/
bezier_optimizer.cpp:27-270): cost matrix, constraint matrix, bounds and the ROW ORDER, entry by entry."""
import json
import os

import numpy as np
import pytest

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "qp_assembly_independent.json")))
MF = 8


def _dense(trip, shape):
    a = np.zeros(shape)
    for i, j, v in trip:
        a[i, j] = v
    return a


def _polys(case):
    p = np.zeros((len(case["polys"]), MF, 4))
    for i, c in enumerate(case["polys"]):
        p[i, :len(c)] = np.array(c, float)
    return p, [len(c) for c in case["polys"]]


@pytest.mark.parametrize("k", range(len(G["cases"])))
def test_oracle_assembly_equals_the_independent_restatement(orc, k):
    c = G["cases"][k]
    polys, nf = _polys(c)
    Q, A, l, u = orc.qp_assemble(c["start"], c["goal"], c["t"], polys, nf, MF, c["vmax"], c["amax"])
    assert Q.shape == (c["n"], c["n"]) and A.shape == (c["m"], c["n"])          # DM_, num (:107-118)
    Qi, Ai = _dense(c["Q"], Q.shape), _dense(c["A"], A.shape)
    # the structure exactly (which entries exist), the values to the last bits (Eigen's product order is not ours)
    assert np.array_equal(A != 0, Ai != 0)
    assert np.abs(A - Ai).max() <= 1e-13 * max(1.0, np.abs(Ai).max())
    assert np.abs(Q - Qi).max() <= 1e-12 * np.abs(Qi).max()
    ub = np.array(c["ub"])
    assert np.abs(u - ub).max() <= 1e-13 * max(1.0, np.abs(ub).max())
    for r, lo in enumerate(c["lb"]):
        if lo is None:
            assert l[r] <= -1e29                                                # -OSQP_INFTY (:264)
        else:
            assert abs(l[r] - lo) <= 1e-13 * max(1.0, abs(lo))
