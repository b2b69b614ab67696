"""The neighbour overlay (other agents' body particles along their Bezier trajectories, fake_particle_risk_voxel.cpp:175-218
with ParticleATC::getWaypoints / getParticlesWithRisk, particles.cpp:316-422) against an INDEPENDENT restatement written
from the reference text (tests/golden/make_overlay_fixture.py -> overlay_independent.json): every incremented (voxel, slice)
and its count, on an empty parity-size grid — incl. a missing record, a trajectory that starts exactly at the map stamp or
between slices 0 and 1 (dropped for the whole update), one that ends inside the horizon, a shuffled record table.
CPU: the C++ oracle.  GPU: sogm_project_neighbours directly."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = json.load(open(os.path.join(ROOT, "tests", "golden", "overlay_independent.json")))


def _records(pop, case):
    recs = (pop._abi.SogmTrajRecord * len(case["records"]))()
    for r, src in zip(recs, case["records"]):
        r.drone_id, r.n_pieces, r.time_start = src["id"], len(src["duration"]), src["time_start"]
        for i, d in enumerate(src["duration"]):
            r.duration[i] = d
        for i, p in enumerate(src["cpts"]):
            for k in range(3):
                r.cpts[3 * i + k] = p[k]
    return recs


def _want(spec, case):
    g = np.zeros((spec.L * spec.W * spec.H, spec.T), np.float32)
    for v, k, n in case["cells"]:
        g[v, k] = n
    return g


@pytest.mark.parametrize("c", range(len(FX["cases"])))
def test_oracle_overlay_equals_the_independent_restatement(pop, orc, c):
    case = FX["cases"][c]
    spec = pop.config.make_spec("parity")
    assert [spec.L, spec.W, spec.H, spec.T] == FX["grid"]
    body = np.asarray(FX["body"], np.float64)
    assert np.array_equal(body, pop.scene.received_body_particles())      # Point32 offsets: what the product uses too
    g = np.zeros((spec.L * spec.W * spec.H, spec.T), np.float32)
    orc.project_neighbours(spec, g, _records(pop, case), len(case["records"]), case["ego"], body, np.float32(case["pose"]),
                           case["stamp"])
    want = _want(spec, case)
    assert np.array_equal(g, want), (np.argwhere(g != want)[:10].tolist(), g.sum(0).tolist(), case["particles_per_slice"])
    assert want.sum() > 300


@pytest.mark.gpu
def test_kernel_overlay_equals_the_independent_restatement(pop):
    import importlib
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    spec = pop.config.make_spec("parity")
    for case in FX["cases"]:
        m = sogm.SogmMap(spec, 1)
        assert np.array_equal(m.body, np.asarray(FX["body"], np.float64))
        far = sogm._dev(np.full((1, 3), 1.0e6, np.float32), np.float32)          # an update with nothing to stamp: an empty map
        m.updateMap(far, sogm._dev(np.int32([[0, 1]]), np.int32), None, 0, sogm._dev(np.float32([case["pose"]]), np.float32),
                    sogm._dev(np.float64([case["stamp"]]), np.float64))
        assert not m.download(0).any()
        recs = sogm._dev(pop.scene.records_to_numpy(_records(pop, case)))
        m.addOtherAgents(recs, len(case["records"]), sogm._dev(np.int32([case["ego"]]), np.int32))
        got, want = m.download(0), _want(spec, case)
        assert np.array_equal(got, want), (np.argwhere(got != want)[:10].tolist(), got.sum(0).tolist(), case["particles_per_slice"])
        m.close()
