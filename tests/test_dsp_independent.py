"""Particle SOGM (row a6) against an INDEPENDENT Python restatement of DSPMap::update's stages written from the reference text
without reading oracle/ (tests/golden/make_dsp_fixture.py -> dsp_independent.json; tables and the new-born list injected): four
updates.  Identical, bit for bit: which slots hold particles and their flags, every particle's velocity and position, the
table cursors, the observation tables.  To a tolerance (the normal-PDF table and the FOV plane normals are computed by the
fixture's interpreter, not by glibc's expf / sinf): weights, C_k, object numbers, future status.  The C++ oracle on the CPU,
sogm_update_dsp directly on the GPU."""
import hashlib
import importlib
import json
import os

import numpy as np
import pytest

from helpers import dsp_fixture_inputs

HERE = os.path.dirname(os.path.abspath(__file__))
OMAX = 100


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _fixture():
    with open(os.path.join(HERE, "golden", "dsp_independent.json")) as f:
        fx = json.load(f)
    tables, seq = dsp_fixture_inputs()
    assert len(seq) == len(fx["updates"])
    for s, u in zip(seq, fx["updates"]):
        blob = np.concatenate([s["points"].ravel(), s["labels"].ravel(), np.asarray(s["pos"], np.float32), np.asarray(s["quat"], np.float32)])
        assert _sha(blob) == u["in_sha256"]
    return fx, tables, seq


def _check(u, ok, store, objnum, cursors, nobs, pc, maxlen):
    assert ok == u["ok"], u["update"]
    occ = store[:, :, 0] > np.float32(0.1)
    assert int(occ.sum()) == u["n_particles"], (u["update"], int(occ.sum()), u["n_particles"])
    assert _sha(store[:, :, 0]) == u["flags_sha256"], f"update {u['update']}: slot flags differ"
    assert _sha(np.where(occ[:, :, None], store[:, :, 1:7], 0)) == u["vel_pos_sha256"], f"update {u['update']}: velocities / positions differ"
    assert [int(c) for c in cursors] == u["cursors"], (u["update"], cursors, u["cursors"])
    assert _sha(nobs) == u["nobs_sha256"] and int(nobs.sum()) == u["nobs_total"], u["update"]
    mask = (np.arange(OMAX)[None, :] < nobs[:, None])
    assert _sha(np.where(mask[:, :, None], pc[:, :, [0, 1, 2, 4]], 0)) == u["obs_xyzl_sha256"], u["update"]
    assert _sha(maxlen) == u["maxlen_sha256"], u["update"]
    w = store[:, :, 7][occ].astype(np.float64)
    np.testing.assert_allclose(w.sum(), u["weight_sum"], rtol=2e-5)
    np.testing.assert_allclose([w.min(), w.max()], [u["weight_min"], u["weight_max"]], rtol=1e-4)
    flat = store.reshape(-1, 9)
    got = np.array([flat[i, 7] for i, _ in u["weight_samples"]], np.float64)
    np.testing.assert_allclose(got, [v for _, v in u["weight_samples"]], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(np.where(mask, pc[:, :, 3], 0).astype(np.float64).sum(), u["ck_sum"], rtol=2e-5)
    np.testing.assert_allclose(objnum[:, 0].astype(np.float64).sum(), u["obj0_sum"], rtol=2e-5)
    np.testing.assert_allclose(objnum[:, 4:].astype(np.float64).sum(axis=0), u["future_sum"], rtol=1e-4, atol=1e-6)
    assert int((objnum[:, 0] > 0).sum()) == u["occupied_voxels"]


def test_oracle_against_the_independent_restatement(pop, orc):
    fx, tables, seq = _fixture()
    dsp = importlib.import_module("pred-occ-planner_amd.dsp")
    spec = pop.config.make_spec("parity")
    spec.map_kind = pop._abi.SOGM_MAP_RISKBASE
    assert [spec.L, spec.W, spec.H, spec.T] == fx["grid"]
    P = dsp.make_dsp_params(spec.T)
    NP = (P.half_fov_h * 2 // P.angle_resolution) * (P.half_fov_v * 2 // P.angle_resolution)
    o = orc.DspOracle(spec, P, tables)
    for s, u in zip(seq, fx["updates"]):
        ok = o.update(s["points"], s["labels"], s["pos"], s["quat"], s["stamp"])
        store, objnum, cnt = o.state()
        nobs, pc, ml = o.observations(NP)
        _check(u, ok, store, objnum, cnt[4:7], nobs, pc, ml)
        o.publish(0.7, 0)      # the consumer clears the future status (getOccupancyMapWithFutureStatus, dsp_dynamic.h:454-476)
    o.close()


@pytest.mark.gpu
def test_kernel_against_the_independent_restatement(pop):
    fx, tables, seq = _fixture()
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    dsp = importlib.import_module("pred-occ-planner_amd.dsp")
    spec = pop.config.make_spec("parity")
    spec.map_kind = pop._abi.SOGM_MAP_RISKBASE
    P = dsp.make_dsp_params(spec.T)
    m = sogm.SogmMap(spec, 1)
    g = dsp.DspMap(m, P, tables)
    for s, u in zip(seq, fx["updates"]):
        n = len(s["points"])
        rng = np.asarray([[0, n]], np.int32)
        ok = g.update(sogm._dev(s["points"], np.float32), sogm._dev(s["labels"], np.float32), sogm._dev(rng, np.int32),
                      sogm._dev(np.asarray(s["pos"], np.float32)[None], np.float32), sogm._dev(np.asarray(s["quat"], np.float32)[None], np.float32),
                      sogm._dev(np.asarray([s["stamp"]], np.float64), np.float64)).cpu().numpy()
        store, objnum, cnt = g.download_state(0)
        assert cnt[10] == 0 and cnt[11] == 0 and cnt[12] == 0, cnt
        nobs, pc, ml = g.download_observations(0)
        _check(u, int(ok[0]), store, objnum, cnt[4:7], nobs, pc, ml)
        g.publish()
    g.close()
    m.close()
