"""CPU checks of the particle-filter SOGM restatement (oracle/dsp_oracle.cpp, row a6).  The reference
has no test or fixture for dsp_map::DSPMap (parity unpinned); these pin the properties the written
algorithm guarantees (dsp_dynamic.h line references in the asserts)."""
import importlib

import numpy as np


def _setup(pop, orc, grid="parity", seed=0x71, n=6):
    dsp = importlib.import_module("pred-occ-planner_amd.dsp")
    spec = pop.config.make_spec(grid)
    P = dsp.make_dsp_params(spec.T)
    tabs = dsp.make_tables(3, n_gauss=1 << 16, n_rand=1 << 10)
    seq = pop.scene.make_dsp_sequence(seed, n)
    return spec, P, tabs, seq


def test_constants_match_reference_macros(pop, orc):
    spec, P, tabs, seq = _setup(pop, orc, n=1)
    o = orc.DspOracle(spec, P, tabs)
    _, _, c = o.state()
    # SAFE_PARTICLE_NUM_VOXEL = 14, SAFE_PARTICLE_NUM_PYRAMID = 20, observation_pyramid_num = 86*58
    # (map_parameters.h:40-51 with the 66x66x20 grid)
    assert list(c[7:10]) == [14, 20, 4988]
    o.close()


def test_first_update_has_zero_odometry_and_only_newborns(pop, orc):
    spec, P, tabs, seq = _setup(pop, orc, n=2)
    o = orc.DspOracle(spec, P, tabs)
    assert o.update(seq[0]["points"], seq[0]["labels"], seq[0]["pos"], seq[0]["quat"], seq[0]["stamp"]) == 1
    store, objnum, c = o.state()
    live = store[:, :, 0] > 0.1
    assert live.sum() > 1000
    # after mapOccupancyCalculationAndResample every surviving flag is 1.0 or 0.6 (:1048,1100)
    assert set(np.unique(store[:, :, 0][live])) <= {np.float32(1.0), np.float32(0.6)}
    # LIMIT_MOVEMENT_IN_XY_PLANE: vz == 0 for every particle (:934-936)
    assert np.all(store[:, :, 3][live] == 0)
    # particles sit in the voxel their position maps to (getParticleVoxelsIndex :1152-1166)
    half = np.float32(spec.resolution) * np.float32([spec.L, spec.W, spec.H]) * np.float32(0.5)
    v, p = np.nonzero(live)
    pos = store[v, p, 4:7]
    idx = ((pos + half) / np.float32(spec.resolution)).astype(np.int32)
    assert np.array_equal(idx[:, 2] * spec.W * spec.L + idx[:, 1] * spec.L + idx[:, 0], v)
    # position-noise table advanced by 3 * 20 per in-map point; no prediction-step velocity draws
    assert c[4] % 60 == 0 and c[4] > 0
    # voxel weight = sum of its particles' weights (:1055)
    np.testing.assert_allclose(objnum[:, 0], (store[:, :, 7] * live).sum(axis=1), rtol=1e-5, atol=1e-7)
    o.close()


def test_rejected_updates_change_nothing(pop, orc):
    spec, P, tabs, seq = _setup(pop, orc, n=3)
    o = orc.DspOracle(spec, P, tabs)
    for s in seq[:2]:
        o.update(s["points"], s["labels"], s["pos"], s["quat"], s["stamp"])
    before = o.state()
    s = seq[2]
    assert o.update(s["points"], s["labels"], s["pos"], np.asarray([1.01, 0, 0, 0], np.float32), s["stamp"]) == 0
    assert o.update(s["points"], s["labels"], s["pos"] + np.float32([11, 0, 0]), s["quat"], s["stamp"]) == 0
    assert o.update(s["points"], s["labels"], s["pos"], s["quat"], seq[0]["stamp"]) == 0  # dt < 0
    after = o.state()
    for b, a in zip(before, after):
        assert np.array_equal(b, a)
    o.close()


def test_publish_clears_future_accumulators(pop, orc):
    spec, P, tabs, seq = _setup(pop, orc, n=4)
    o = orc.DspOracle(spec, P, tabs)
    for s in seq:
        o.update(s["points"], s["labels"], s["pos"], s["quat"], s["stamp"])
    _, objnum, _ = o.state()
    grid, n_occ = o.publish(spec.risk_threshold, 2)
    assert n_occ == int((objnum[:, 0] > spec.risk_threshold).sum())
    # future status = accumulated weights since the last publish (:459-466), minus the cells the
    # inflate-kernel loop of RiskVoxel::publishMap zeroes (in-bounds indices only)
    diff = np.nonzero(grid != objnum[:, 4:])
    assert np.all(grid[diff] == 0) and np.all(diff[1] < 3)
    _, objnum2, _ = o.state()
    assert np.all(objnum2[:, 4:] == 0)
    o.close()
