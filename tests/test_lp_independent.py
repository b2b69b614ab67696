"""CPU: the C++ oracle's LP (oracle/lp_oracle.cpp through `orc_linprog_perm`) against an INDEPENDENT Python restatement of
sdlp::linprog<3> / <4> written from the reference text without reading oracle/ (tests/golden/make_lp_fixture.py ->
tests/golden/lp_independent.json: 116 problems — bounded polytopes up to 152 rows, the interior-point shape of
firi.hpp:142-170, infeasible, unbounded, zero-objective, degenerate and single-row cases, each with the permutation that
stands for rand_permutation's draw).  Optimum value and optimal POINT agree to the last bit (both readings sum the column norms and c.x left to right, where the
text leaves the order to Eigen); +-inf conventions exactly.  Two separately written readings of plan_manager/include/sfc_gen/sdlp.hpp agree;
that does not pin the oracle to the reference itself."""
import json
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fixture():
    with open(os.path.join(ROOT, "tests", "golden", "lp_independent.json")) as f:
        return json.load(f)["problems"]


def _value(s):
    return {"inf": math.inf, "-inf": -math.inf}.get(s, s)


def test_oracle_lp_gives_the_optimum_and_point_of_the_independent_restatement(orc):
    probs = _fixture()
    exact = 0
    kinds = {}
    for i, p in enumerate(probs):
        v, x = orc.linprog_perm(p["c"], np.asarray(p["A"], float).reshape(len(p["b"]), len(p["c"])), p["b"], p["perm"])
        want_v, want_x = _value(p["minimum"]), np.asarray(p["x"])
        assert v == want_v and np.array_equal(x, want_x), (i, p["kind"], v, want_v, x, want_x)   # to the last bit
        exact += 1
        kinds[p["kind"]] = kinds.get(p["kind"], 0) + 1
    print(f"{exact} of {len(probs)} problems bit-identical", kinds)
    assert exact >= len(probs) * 0.9
    assert set(kinds) >= {"bounded", "interior", "infeasible", "unbounded", "zero_objective", "degenerate", "single_row"}


@pytest.mark.gpu
def test_kernel_lp_gives_the_optimum_and_point_of_the_independent_restatement():
    """sogm_linprog_batched (HIP, through the C ABI) held to the fixture DIRECTLY, no oracle in between; the kernel inserts
    the rows in the library's own fixed order, for which the restatement's results are stored beside the drawn ones"""
    import importlib
    import torch
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    probs = _fixture()
    for d in (3, 4):
        sel = [p for p in probs if len(p["c"]) == d]
        cs = np.array([p["c"] for p in sel])
        As = np.concatenate([np.asarray(p["A"], float).reshape(-1, d) for p in sel])
        bs = np.concatenate([np.asarray(p["b"], float) for p in sel])
        ends = np.cumsum([len(p["b"]) for p in sel])
        rr = np.stack([ends - [len(p["b"]) for p in sel], ends], axis=1).astype(np.int32)
        x, v = planner.linprog_batched(torch.tensor(cs, device="cuda"), torch.tensor(As, device="cuda"),
                                       torch.tensor(bs, device="cuda"), torch.tensor(rr, device="cuda"))
        x, v = x.cpu().numpy(), v.cpu().numpy()
        for k, p in enumerate(sel):
            assert v[k] == _value(p["minimum_library_order"]), (d, k, p["kind"], v[k], p["minimum_library_order"])
            assert np.array_equal(x[k], np.asarray(p["x_library_order"])), (d, k, p["kind"], x[k], p["x_library_order"])
