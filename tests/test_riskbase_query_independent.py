"""RiskBase::getClearOcccupancy (risk_base.cpp:16-39,228-262: the query of the planner that reads a published SOGM — cubic
125-cell kernel, decayed region threshold, `occupied` outside the height band) against an INDEPENDENT restatement written
from the reference text (tests/golden/make_riskbase_query_fixture.py -> riskbase_query_independent.json), on a grid of
fractional risks so that the ORDER of the float sum matters.  CPU: the C++ oracle.  GPU: sogm_query_clear directly."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = json.load(open(os.path.join(ROOT, "tests", "golden", "riskbase_query_independent.json")))
MAPFX = json.load(open(os.path.join(ROOT, "tests", "golden", "map_independent.json")))


def _grid(pop, orc):
    """0.35 x the oracle's fake-map grid of map_independent.json case 0 — which tests/test_map_independent.py holds to the
    independent map restatement cell by cell"""
    sc = pop.scene.make_scene(MAPFX["agents"], 4.95, seed=MAPFX["seed"], moving=True)
    cyl = pop.scene.cylinders_to_struct(sc["cylinders"])
    pose = np.float32(FX["pose"])
    g = orc.update_gt(pop.config.make_spec("parity"), sc["cloud"], cyl, len(sc["cylinders"]), pose)
    return (g * np.float32(FX["scale"])).astype(np.float32), pose


def test_oracle_riskbase_query_equals_the_independent_restatement(pop, orc):
    g, pose = _grid(pop, orc)
    spec = pop.config.make_spec("parity", map_kind=1)
    pos = np.asarray(FX["query_pos"], np.float64)
    seen = set()
    for i in range(len(pos)):
        r_t = orc.query_clear(spec, g, pose, pos[i], int(FX["query_t"][i]), t_is_index=True)
        r_d = orc.query_clear(spec, g, pose, pos[i], float(FX["query_dt"][i]))
        assert r_t == FX["result_t"][i], (i, pos[i], FX["query_t"][i], r_t, FX["result_t"][i])
        assert r_d == FX["result_dt"][i], (i, pos[i], FX["query_dt"][i], r_d, FX["result_dt"][i])
        seen.add(r_t)
    assert seen == {-1, 0, 1}


@pytest.mark.gpu
def test_kernel_riskbase_query_equals_the_independent_restatement(pop, orc):
    import importlib
    import torch
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    g, pose = _grid(pop, orc)
    spec = pop.config.make_spec("parity", map_kind=1)
    m = sogm.SogmMap(spec, 1)
    m.futureRiskCallback(torch.from_numpy(g[None]).cuda().contiguous(), sogm._dev(pose[None], np.float32),
                         sogm._dev(np.float64([100.0]), np.float64))
    assert np.array_equal(m.download(0), g)
    n = len(FX["query_pos"])
    agent = sogm._dev(np.zeros(n, np.int32), np.int32)
    pos = sogm._dev(np.asarray(FX["query_pos"], np.float64), np.float64)
    got_t = m.getClearOcccupancy(agent, pos, sogm._dev(np.float64(FX["query_t"]), np.float64), True).cpu().numpy()
    got_d = m.getClearOcccupancy(agent, pos, sogm._dev(np.float64(FX["query_dt"]), np.float64), False).cpu().numpy()
    assert got_t.tolist() == FX["result_t"]
    assert got_d.tolist() == FX["result_dt"]
    m.close()
