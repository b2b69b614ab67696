"""CPU, world_size 2 over gloo: the N>1 path of the tick driver — agent sharding, the single
all-gather of trajectory records per tick, latest-wins merge — without a GPU."""
import importlib
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pop = importlib.import_module("pred-occ-planner_amd")
    drv = importlib.import_module("pred-occ-planner_amd.driver")
    abi = pop._abi
    A_loc = 3
    lo, hi = drv.shard_bounds(rank, world, A_loc)
    sc = pop.scene.make_scene(A_loc * world, 4.95, seed=7)
    recs = pop.scene.straight_records(sc)
    allb = np.frombuffer(bytes(recs), dtype=np.uint8).reshape(A_loc * world, abi.TRAJ_RECORD_BYTES)
    own_new = torch.from_numpy(allb[lo:hi].copy())
    own_old = torch.zeros_like(own_new)
    ok = torch.tensor([1, 0, 1], dtype=torch.int32)          # middle agent's replan "failed"
    own = drv.merge_latest(own_new, own_old, ok)
    gathered = torch.zeros((A_loc * world, abi.TRAJ_RECORD_BYTES), dtype=torch.uint8)
    for _tick in range(2):                                   # exactly one collective per tick
        drv.exchange_records(own, gathered, dist, world)
    got = gathered.numpy()
    good = True
    for r in range(world):
        l2, h2 = drv.shard_bounds(r, world, A_loc)
        want = allb[l2:h2].copy()
        want[1] = 0
        good &= np.array_equal(got[l2:h2], want)
    # record identity survives the exchange: drone ids are the global agent indices
    arr = (abi.SogmTrajRecord * (A_loc * world)).from_buffer_copy(got.tobytes())
    ids = [arr[i].drone_id for i in range(A_loc * world) if arr[i].n_pieces > 0]
    good &= ids == [i for i in range(A_loc * world) if i % A_loc != 1]
    out[rank] = bool(good)
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_allgather_and_sharding():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0] and out[1]


def test_shard_bounds_partition_agents():
    pop = importlib.import_module("pred-occ-planner_amd")
    drv = importlib.import_module("pred-occ-planner_amd.driver")
    seen = []
    for r in range(8):
        lo, hi = drv.shard_bounds(r, 8, 64)
        seen += list(range(lo, hi))
    assert seen == list(range(512))      # BASELINE configs[3]: 512 agents over 8 GPUs
