"""GridMap depth front end (row f1) against an INDEPENDENT Python restatement of projectDepthImage / raycastProcess (with
RayCaster) / clearAndInflateLocalMap written from the reference text without reading oracle/
(tests/golden/make_gridmap_fixture.py -> gridmap_independent.json): nine frames of a creeping camera; after every frame the
fp64 log-odds buffer, the inflated occupancy and the local bounds must be identical, bit for bit — the C++ oracle on the CPU,
sogm_gridmap_update directly on the GPU."""
import hashlib
import importlib
import json
import os

import numpy as np
import pytest

from helpers import gridmap_fixture_frames, gridmap_fixture_params

HERE = os.path.dirname(os.path.abspath(__file__))


def _fixture():
    with open(os.path.join(HERE, "golden", "gridmap_independent.json")) as f:
        fx = json.load(f)
    frames = gridmap_fixture_frames()
    assert len(frames) == len(fx["frames"])
    for (img, cam, R), fr in zip(frames, fx["frames"]):   # the inputs are the ones the fixture was made from
        assert hashlib.sha256(img.tobytes() + np.asarray(cam).tobytes() + np.asarray(R).tobytes()).hexdigest() == fr["in_sha256"]
    return fx, frames


def _params(pop):
    gm = importlib.import_module("pred-occ-planner_amd.gridmap")
    p, d = gm.make_gridmap_params(), gridmap_fixture_params()
    for k, v in d.items():
        if isinstance(v, list):
            getattr(p, k)[:] = v
        else:
            setattr(p, k, v)
    return p


def _check(fr, upd, occ, inf, bounds):
    assert upd == fr["updated"], fr["frame"]
    if fr["updated"]:   # (before the first raycast the reference's local bounds are uninitialised members: each reading picks a value)
        assert [int(v) for v in bounds] == fr["bounds"], (fr["frame"], bounds, fr["bounds"])
    assert int((occ > np.log(0.8 / 0.2)).sum()) == fr["cells_occupied"] and int(inf.sum()) == fr["cells_inflated"], fr["frame"]
    assert hashlib.sha256(np.ascontiguousarray(occ, np.float64).tobytes()).hexdigest() == fr["occ_sha256"], \
        f"frame {fr['frame']}: the log-odds buffer differs (sum {occ.sum()!r} vs {fr['occ_sum']!r})"
    assert hashlib.sha256(np.ascontiguousarray(inf, np.int8).tobytes()).hexdigest() == fr["inflate_sha256"], fr["frame"]


def test_oracle_against_the_independent_restatement(pop, orc):
    fx, frames = _fixture()
    o = orc.GridMapOracle(_params(pop))
    assert o.nv == fx["voxels"]
    for (img, cam, R), fr in zip(frames, fx["frames"]):
        upd = o.update(img, cam, R)
        occ, inf, b = o.state()
        _check(fr, upd, occ, inf, b)
    assert fx["frames"][-1]["cells_occupied"] > 100      # (the flight reaches occupied cells and inflates them)
    o.close()


@pytest.mark.gpu
def test_kernel_against_the_independent_restatement(pop):
    import torch
    fx, frames = _fixture()
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    gm = importlib.import_module("pred-occ-planner_amd.gridmap")
    g = gm.GridMap(_params(pop), 1)
    for (img, cam, R), fr in zip(frames, fx["frames"]):
        depth = img[None].view(np.int16)
        upd = g.update(torch.from_numpy(depth).cuda(), sogm._dev(np.asarray(cam)[None], np.float64),
                       sogm._dev(np.asarray(R).reshape(1, 9), np.float64)).cpu().numpy()
        occ, inf, b, cnt = g.download(0)
        assert cnt[3] == 0, cnt
        _check(fr, int(upd[0]), occ, inf, b)
    g.close()
