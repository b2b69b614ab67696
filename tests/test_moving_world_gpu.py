"""GPU: the map update from per-tick sensor frames (SogmWorld / sogm_update_world) and flights through a MOVING world.

The reference rebuilds the SOGM from the cloud and the obstacle states that arrived for that update
(plan_env/src/map.cpp:170-171 cloudCallback -> updateMap, fake_particle_risk_voxel.cpp:244-264
groundTruthStateCallback) and crops the cloud around the CURRENT pose (fake_particle_risk_voxel.cpp:88-104).  Here:
scene.WorldTimeline produces the frame of every tick (cylinders advanced by v dt, their cloud points with them), the
device crops it per agent through the block index, and every stage is held to the oracle run on the same frame."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fp():
    return importlib.import_module("test_full_size_parity")


@pytest.mark.parametrize("block_points", [64, 256, 1000])
def test_update_world_equals_update_gt_on_the_whole_cloud(pop, orc, block_points):
    """sogm_update_world (device-side crop through the block lists) marks exactly the cells sogm_update_gt_swarm marks
    when every agent is handed the whole cloud, frame after frame — and both equal the oracle's build."""
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    spec = pop.config.make_spec("parity")
    A = 5
    sc = pop.scene.make_scene(A, 4.95, seed=0x61, moving=True)
    tl = pop.scene.WorldTimeline(sc, 0.1, moving=True)
    mw, mr = sogm.SogmMap(spec, A), sogm.SogmMap(spec, A)
    ego = sogm._dev(sc["ego_ids"], np.int32)
    for k in (0, 9, 23):
        f = tl.frame(k)
        poses = (sc["poses"] + np.float32([0.21 * k, -0.13 * k, 0.01 * k])).astype(np.float32)
        stamps = sc["stamps"] + 0.1 * k
        sck = dict(sc, cloud=f["cloud"], cylinders=f["cylinders"], poses=poses, stamps=stamps)
        recs = pop.scene.straight_records(sck, speed=1.0 + 0.05 * k)
        w = sogm.World(f["cloud"], f["cylinders"], block_points=block_points)
        assert w.n_blocks == -(-len(f["cloud"]) // block_points)
        d_poses, d_stamps, d_recs = sogm._dev(poses, np.float32), sogm._dev(stamps, np.float64), sogm._dev(recs)
        mw.updateWorld(w, d_poses, d_stamps, d_recs, A, ego)
        dev = sogm.upload_scene(sck)
        mr.updateMapSwarm(dev["cloud"], dev["cloud_range"], dev["cylinders"], dev["n_cyl"], d_poses, d_stamps, d_recs, A, ego)
        cyl = pop.scene.cylinders_to_struct(f["cylinders"])
        for a in range(A):
            gw, gr = mw.download(a), mr.download(a)
            assert np.array_equal(gw, gr), f"frame {k}, agent {a}: {(gw != gr).sum()} cells differ between the two crops"
            want = orc.update_gt(spec, f["cloud"], cyl, len(f["cylinders"]), poses[a])
            orc.project_neighbours(spec, want, recs, A, a, mw.body, poses[a], stamps[a])
            assert np.array_equal(gw, want), f"frame {k}, agent {a} differs from the oracle"
            assert int((want != 0).sum()) > 100
    mw.close()
    mr.close()


def test_update_world_with_nothing_in_the_window(pop):
    """an empty cloud, and a cloud whose blocks all lie outside every agent's window: empty maps, no error"""
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    spec = pop.config.make_spec("parity")
    A = 3
    sc = pop.scene.make_scene(A, 4.95, seed=0x62)
    m = sogm.SogmMap(spec, A)
    poses, stamps = sogm._dev(sc["poses"], np.float32), sogm._dev(sc["stamps"], np.float64)
    far = sc["cloud"] + np.float32([500.0, 0.0, 0.0])
    for cloud in (np.zeros((0, 3), np.float32), far):
        m.updateWorld(sogm.World(cloud, sc["cylinders"]), poses, stamps)
        for a in range(A):
            assert not m.download(a).any()
    # and back to a populated frame on the same context
    m.updateWorld(sogm.World(sc["cloud"], sc["cylinders"]), poses, stamps)
    assert sum(int(m.download(a).any()) for a in range(A)) > 0
    m.close()


def test_invalid_world_is_rejected(pop):
    import ctypes as C
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    spec = pop.config.make_spec("parity")
    sc = pop.scene.make_scene(2, 4.95, seed=0x63)
    m = sogm.SogmMap(spec, 2)
    poses, stamps = sogm._dev(sc["poses"], np.float32), sogm._dev(sc["stamps"], np.float64)
    w = sogm.World(sc["cloud"], sc["cylinders"])
    bad = pop._abi.SogmWorld.from_buffer_copy(bytes(w.c))
    bad.n_blocks += 1  # does not match ceil(n_points / block_points)
    rc = pop.lib().sogm_update_world(m.ctx, C.byref(bad), poses.data_ptr(), stamps.data_ptr(), None, 0, None, None)
    assert rc == pop._abi.SOGM_ERR_INVALID_ARG
    bad = pop._abi.SogmWorld.from_buffer_copy(bytes(w.c))
    bad.block_points = 8
    rc = pop.lib().sogm_update_world(m.ctx, C.byref(bad), poses.data_ptr(), stamps.data_ptr(), None, 0, None, None)
    assert rc == pop._abi.SOGM_ERR_INVALID_ARG
    m.close()


def test_parity_grid_flight_through_a_moving_world_stage_by_stage(pop, orc):
    """Six agents, 66 x 66 x 20 x 6, eight closed-loop ticks through the moving world: at every tick the map built from
    that tick's frame (every cell), the A* pop order, the polytopes and the QP of every agent against the oracle on the
    same frame; the frames differ from tick to tick (asserted on the oracle's maps for a fixed pose)."""
    driver = importlib.import_module("pred-occ-planner_amd.driver")
    fp = _fp()
    sw = driver.SwarmTick("parity", 6, moving_world=True, prestamp=False)
    assert sw.compute.use_world and sw.compute.timeline.moving and sw.map_input_staleness_ticks == 0
    acc = {}
    for _ in range(8):
        fp._sum(acc, fp._tick_with_parity(pop, orc, sw, list(range(6))))
    print("parity grid, moving world:", acc)
    assert acc["agents"] == 48 and acc["polys"] > 40 and acc["qp_ok"] >= 20 and acc["fused_checked"] == 48
    tl, spec = sw.compute.timeline, sw.spec
    pose = np.float32([0.0, 0.0, 1.0])
    grids = []
    for k in (0, 4, 7):
        f = tl.frame(k)
        grids.append(orc.update_gt(spec, f["cloud"], pop.scene.cylinders_to_struct(f["cylinders"]), len(f["cylinders"]), pose))
    assert (grids[0] != grids[1]).sum() > 50 and (grids[1] != grids[2]).sum() > 50
    sw.close()


@pytest.mark.parametrize("prestamp", [False, True])
def test_parity_grid_flight_on_the_tick_path_moving_world(pop, orc, prestamp):
    """SwarmTick.step() itself (pooled grids, sparse reset, dataflow replan) through the moving world, with the map update
    at the start of the tick (staleness 0: the map is the oracle's build of THIS tick's frame) and with the pre-stamp
    (staleness 1: of the PREVIOUS tick's frame around the current pose), ticks 3..6 checked cell by cell and stage by
    stage."""
    driver = importlib.import_module("pred-occ-planner_amd.driver")
    fp = _fp()
    sw = driver.SwarmTick("parity", 6, moving_world=True, prestamp=prestamp, grids=3)
    assert sw.prestamp == prestamp and sw.map_input_staleness_ticks == (1 if prestamp else 0)
    for _ in range(3):
        sw.step()
    acc = {}
    for _ in range(4):
        fp._sum(acc, fp._step_with_parity(pop, orc, sw, list(range(6)), list(range(6)),
                                          {"min_sparse_resets": 1, "prestamped": prestamp}))
    print("parity grid tick path, moving world, prestamp", prestamp, acc)
    assert acc["agents"] == 24 and acc["qp_ok"] >= 8
    sw.close()
