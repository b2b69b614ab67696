"""CPU: the C++ oracle's map (oracle/map_oracle.cpp) against an INDEPENDENT numpy restatement of the same reference text
(tests/golden/make_map_fixture.py -> tests/golden/map_independent.json): FakeParticleRiskVoxel::updateMap's marked cells
(count, per slice, SHA-256 of the sorted flat [V][T] indices) and getClearOcccupancy(pos, int) / (pos, double) on 2 x 400
queries per case.  Two separately written readings of plan_env/src/fake_particle_risk_voxel.cpp:80-170,309-346 and
plan_env/include/plan_env/map.h:153-205 have to agree — the fixture does not pin the oracle to the reference itself
(nothing here can: DESIGN.md section 4)."""
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fixture():
    with open(os.path.join(ROOT, "tests", "golden", "map_independent.json")) as f:
        return json.load(f)


def test_scene_generator_still_produces_the_fixture_inputs(pop):
    fx = _fixture()
    sc = pop.scene.make_scene(fx["agents"], 4.95, seed=fx["seed"], moving=True)
    assert len(sc["cloud"]) == fx["cloud_points"]
    assert hashlib.sha256(sc["cloud"].tobytes()).hexdigest() == fx["cloud_sha256"]


def test_oracle_update_map_marks_the_cells_of_the_independent_restatement(pop, orc):
    fx = _fixture()
    spec = pop.config.make_spec("parity")
    assert [spec.L, spec.W, spec.H, spec.T] == fx["grid"]
    sc = pop.scene.make_scene(fx["agents"], 4.95, seed=fx["seed"], moving=True)
    cyl = pop.scene.cylinders_to_struct(sc["cylinders"])
    for a, case in enumerate(fx["cases"]):
        g = orc.update_gt(spec, sc["cloud"], cyl, len(sc["cylinders"]), np.float32(case["pose"]))
        assert set(np.unique(g)) <= {0.0, 1.0}
        occ = np.flatnonzero(g.ravel()).astype(np.int64)
        assert len(occ) == case["occupied_cells"], (a, len(occ), case["occupied_cells"])
        assert [int((occ % spec.T == k).sum()) for k in range(spec.T)] == case["occupied_per_slice"]
        assert occ[:40].tolist() == case["occupied_first"]
        assert hashlib.sha256(occ.tobytes()).hexdigest() == case["occupied_sha256"], f"case {a}: a marked cell differs"


def test_oracle_collision_query_answers_like_the_independent_restatement(pop, orc):
    fx = _fixture()
    spec = pop.config.make_spec("parity")
    sc = pop.scene.make_scene(fx["agents"], 4.95, seed=fx["seed"], moving=True)
    cyl = pop.scene.cylinders_to_struct(sc["cylinders"])
    seen = set()
    for a, case in enumerate(fx["cases"]):
        pose = np.float32(case["pose"])
        g = orc.update_gt(spec, sc["cloud"], cyl, len(sc["cylinders"]), pose)
        pos = np.asarray(case["query_pos"], np.float64)
        for i in range(len(pos)):
            r_t = orc.query_clear(spec, g, pose, pos[i], int(case["query_t"][i]), t_is_index=True)
            r_d = orc.query_clear(spec, g, pose, pos[i], float(case["query_dt"][i]))
            assert r_t == case["result_t"][i], (a, i, pos[i], case["query_t"][i], r_t, case["result_t"][i])
            assert r_d == case["result_dt"][i], (a, i, pos[i], case["query_dt"][i], r_d, case["result_dt"][i])
            seen.add(r_t)
    assert seen == {-1, 0, 1}
    assert fx["kernel_cells"] == 25 and fx["inf_step"] == 2   # (0.45f / 0.15f truncates to 2; z-degenerate kernel)
