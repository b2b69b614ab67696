"""The trajectory exchange behind the C ABI (sogm_traj_allgather = ncclAllGather on the context's exchange stream).
One GPU per box here, so the communicator has ONE rank: the collective degenerates to a copy, but the whole path
runs — RCCL resolved with dlopen, communicator from a unique id, launch on the exchange stream, consumers ordered
behind it by events.  (N > 1 on hardware is the driver's SCALE run; the N > 1 bookkeeping is covered on CPU with
gloo in test_driver_gloo.py.)"""
import ctypes as C
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_allgather_world1_and_consumer_ordering(pop, orc):
    import torch
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    abi, lib = pop._abi, pop.lib()
    A = 6
    spec = pop.config.make_spec("parity")
    sc = pop.scene.make_scene(A, 4.95, seed=31, circle_radius=2.5, n_cyl=20)
    dev = sogm.upload_scene(sc)
    m = sogm.SogmMap(spec, A)
    ident = C.create_string_buffer(abi.SOGM_COMM_ID_BYTES)
    abi.check(lib.sogm_comm_unique_id(ident), "sogm_comm_unique_id")
    comm = C.c_void_p()
    abi.check(lib.sogm_comm_create(ident.raw, 0, 1, 0, C.byref(comm)), "sogm_comm_create")
    handle = lib.sogm_comm_handle(comm)
    assert handle
    recs = pop.scene.straight_records(sc)
    own = sogm._dev(recs)                                   # uint8 [A, 2064]
    allr = torch.zeros_like(own)
    m.updateMap(dev["cloud"], dev["cloud_range"], dev["cylinders"], dev["n_cyl"], dev["poses"], dev["stamps"])
    for rep in range(3):
        abi.check(lib.sogm_traj_allgather(m.ctx, handle, own.data_ptr(), A, allr.data_ptr(), sogm._stream()),
                  "sogm_traj_allgather")
        # the consumer waits for the collective inside the library: no host or stream sync here
        m.updateMap(dev["cloud"], dev["cloud_range"], dev["cylinders"], dev["n_cyl"], dev["poses"], dev["stamps"])
        m.addOtherAgents(allr, A, dev["ego_ids"])
    abi.check(lib.sogm_exchange_wait(m.ctx, sogm._stream()), "sogm_exchange_wait")
    assert torch.equal(allr, own)
    cyl = pop.scene.cylinders_to_struct(sc["cylinders"])
    for a in range(A):
        want = orc.update_gt(spec, sc["cloud"], cyl, dev["n_cyl"], sc["poses"][a])
        orc.project_neighbours(spec, want, recs, A, a, m.body, sc["poses"][a], sc["stamps"][a])
        assert np.array_equal(m.download(a), want)
    # argument checks
    assert lib.sogm_traj_allgather(m.ctx, None, own.data_ptr(), A, allr.data_ptr(), None) == abi.SOGM_ERR_INVALID_ARG
    assert lib.sogm_comm_create(ident.raw, 1, 1, 0, C.byref(C.c_void_p())) == abi.SOGM_ERR_INVALID_ARG
    torch.cuda.synchronize()
    lib.sogm_comm_destroy(comm)
    m.close()
