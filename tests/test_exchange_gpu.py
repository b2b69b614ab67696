"""The trajectory exchange behind the C ABI (sogm_traj_allgather = ncclAllGather on the context's exchange stream).
One GPU per box here, so the communicator has ONE rank: the collective degenerates to a copy, but the whole path
runs — RCCL resolved with dlopen, communicator from a unique id, launch on the exchange stream, consumers ordered
behind it by events.  (N > 1 on hardware is the driver's SCALE run; the N > 1 bookkeeping is covered on CPU with
gloo in test_driver_gloo.py.)"""
import ctypes as C
import importlib

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]  # (pytest-timeout: a stuck collective must not hang the suite)


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


def test_swarm_tick_under_a_one_rank_process_group_takes_the_abi_exchange(pop):
    """driver.SwarmTick under torch.distributed (backend nccl = RCCL, world size 1): the communicator comes through
    sogm_comm_* with the id broadcast from rank 0, every tick's records travel through sogm_traj_allgather on the
    exchange stream, and the flight — ok flags and records of 5 ticks — is the one the process-group-less run gives."""
    import socket
    import torch
    import torch.distributed as dist
    driver = importlib.import_module("pred-occ-planner_amd.driver")

    def flight(d):
        sw = driver.SwarmTick("parity", 6, 0, 1, 0, dist=d)
        oks = [sw.step().cpu().numpy().copy() for _ in range(5)]
        table = sw.records_all().cpu().numpy().copy()
        info = (sw.exchange.active, sw.exchange.fallback_reason, sw.distributed)
        sw.close()
        return oks, table, info

    ref_oks, ref_table, ref_info = flight(None)
    assert ref_info == (False, None, False)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        oks, table, info = flight(dist)
    finally:
        dist.destroy_process_group()
    assert info == (True, None, True)
    assert all(np.array_equal(a, b) for a, b in zip(oks, ref_oks)) and sum(int(o.sum()) for o in oks) >= 12
    assert np.array_equal(table, ref_table), np.argwhere((table != ref_table).any(axis=1)).ravel().tolist()


_TWO_RANK_SCRIPT = r"""
import ctypes as C, importlib, os, sys, threading
import numpy as np, torch
sys.path.insert(0, os.environ["SOGM_REPO"])
pop = importlib.import_module("pred-occ-planner_amd")
sogm = importlib.import_module("pred-occ-planner_amd.sogm")
abi, lib = pop._abi, pop.lib()
WORLD, A_LOC, TICKS = 2, 3, 4
spec = pop.config.make_spec("parity")
ident = C.create_string_buffer(abi.SOGM_COMM_ID_BYTES)
abi.check(lib.sogm_comm_unique_id(ident), "unique_id")
assert ident.raw.startswith(b"fake-rccl-"), ident.raw[:16]          # the stand-in, not torch's RCCL
# per tick and rank: distinct records (drone ids = global agent index, time_start = tick)
def make(tick, rank):
    sc = pop.scene.make_scene(WORLD * A_LOC, 4.95, seed=100 + tick)
    sc["stamps"] = np.full(WORLD * A_LOC, float(tick))
    recs = pop.scene.straight_records(sc)
    b = np.frombuffer(bytes(recs), dtype=np.uint8).reshape(WORLD * A_LOC, abi.TRAJ_RECORD_BYTES)
    return b[rank * A_LOC:(rank + 1) * A_LOC].copy(), b.copy()
snaps, errs = {}, []
def run(rank):
    try:
        torch.cuda.set_device(0)
        st = torch.cuda.Stream()
        m = sogm.SogmMap(spec, A_LOC)
        comm = C.c_void_p()
        abi.check(lib.sogm_comm_create(ident.raw, rank, WORLD, 0, C.byref(comm)), "comm_create")
        handle = lib.sogm_comm_handle(comm)
        own = torch.zeros((A_LOC, abi.TRAJ_RECORD_BYTES), dtype=torch.uint8, device="cuda")
        allr = torch.zeros((WORLD * A_LOC, abi.TRAJ_RECORD_BYTES), dtype=torch.uint8, device="cuda")
        big = torch.zeros(64 << 20, dtype=torch.float32, device="cuda")
        out = []
        with torch.cuda.stream(st):
            for tick in range(TICKS):
                mine, _ = make(tick, rank)
                src = torch.from_numpy(mine).cuda()
                for _ in range(4):
                    big.add_(1.0)                     # a slow producer: the local records land ~1 ms after the call
                own.copy_(src)
                abi.check(lib.sogm_traj_allgather(m.ctx, handle, own.data_ptr(), A_LOC, allr.data_ptr(),
                                                  C.c_void_p(st.cuda_stream)), "allgather")
                # consumer on the caller's stream, ordered inside the library only (no host / stream sync here)
                if not os.environ.get("SOGM_TEST_SKIP_CONSUMER_WAIT"):   # (negative control: see the test below)
                    abi.check(lib.sogm_exchange_wait(m.ctx, C.c_void_p(st.cuda_stream)), "exchange_wait")
                out.append(allr.clone())
        st.synchronize()
        snaps[rank] = [o.cpu().numpy() for o in out]
        torch.cuda.synchronize()
        lib.sogm_comm_destroy(comm)
        m.close()
    except Exception as e:  # noqa: BLE001
        errs.append((rank, repr(e)))
ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(WORLD)]   # a rank stuck in the rendezvous must not keep the process alive
[t.start() for t in ts]
[t.join(120) for t in ts]
if errs or any(t.is_alive() for t in ts):
    print("FAILED", errs, [t.is_alive() for t in ts], flush=True)
    os._exit(3)
for tick in range(TICKS):
    _, want = make(tick, 0)
    for r in range(WORLD):
        assert np.array_equal(snaps[r][tick], want), (tick, r)
print("two-rank exchange ok")
"""


def test_allgather_two_ranks_event_ordering_against_a_slow_collective(pop, tmp_path):
    """sogm_traj_allgather with TWO ranks and a collective that is not a no-op: tests/fake_rccl.cpp (an in-process
    stand-in for librccl, two host threads = two ranks on one GPU, loaded through SOGM_RCCL_LIB) moves the bytes on
    the exchange streams behind a 1 ms spin.  The producer is slow too, so a collective that did not wait for the
    caller's stream would gather the previous tick's records, and a consumer not ordered behind the completion event
    would read the previous tick's table: 4 ticks, both ranks must see exactly the table of each tick."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = tmp_path / "libfake_rccl.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC",
                           os.path.join(root, "tests", "fake_rccl.cpp"), "-o", str(so)])
    env = dict(os.environ, SOGM_RCCL_LIB=str(so), SOGM_REPO=root)
    r = subprocess.run([sys.executable, "-c", _TWO_RANK_SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "two-rank exchange ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    # negative control: the same flight with the consumer NOT ordered behind the collective must read stale tables —
    # i.e. the stand-in's collective really is slow enough for this test to see a missing wait
    env["SOGM_TEST_SKIP_CONSUMER_WAIT"] = "1"
    r = subprocess.run([sys.executable, "-c", _TWO_RANK_SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "AssertionError" in r.stderr and "two-rank exchange ok" not in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


_TWO_RANK_FLIGHT = r"""
import ctypes as C, importlib, os, sys, threading
import numpy as np, torch
sys.path.insert(0, os.environ["SOGM_REPO"])
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
WORLD, A_LOC, TICKS = 2, 4, 6
tls = threading.local()

class ThreadDist:
    # what driver.RecordExchange needs of torch.distributed, between the threads of one process
    def __init__(self, world):
        self.world, self.bar, self.box, self.shared = world, threading.Barrier(world), [None] * world, None
    def is_initialized(self): return True
    def get_backend(self): return "nccl"
    def broadcast_object_list(self, lst, src=0):
        if tls.rank == src: self.shared = list(lst)
        self.bar.wait(); lst[:] = self.shared; self.bar.wait()
    def all_gather_object(self, out, obj):
        self.box[tls.rank] = obj
        self.bar.wait(); out[:] = list(self.box); self.bar.wait()

dist = ThreadDist(WORLD)
res, errs = {}, []
def run(rank):
    try:
        tls.rank = rank
        torch.cuda.set_device(0)
        with torch.cuda.stream(torch.cuda.Stream()):
            sw = driver.SwarmTick("parity", A_LOC, rank, WORLD, 0, dist=dist)
            assert sw.exchange.active and sw.distributed and sw.publish      # RCCL (stand-in) behind the C ABI
            oks = [int(sw.step().sum().item()) for _ in range(TICKS)]
            table = sw.records_all().cpu().numpy().copy()
            own = sw.own.cpu().numpy().copy()
            torch.cuda.current_stream().synchronize()
            res[rank] = (oks, table, own)
            dist.bar.wait()          # nobody destroys its communicator while the other is still in a collective
            sw.close()
    except Exception as e:  # noqa: BLE001
        import traceback
        errs.append((rank, traceback.format_exc()))
        try: dist.bar.abort()
        except Exception: pass
ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(WORLD)]
[t.start() for t in ts]
[t.join(240) for t in ts]
if errs or any(t.is_alive() for t in ts):
    print("FAILED", errs, [t.is_alive() for t in ts], flush=True)
    os._exit(3)
# the same flight in one process (no process group, the table refreshed by the replan's publication)
sw = driver.SwarmTick("parity", A_LOC * WORLD)
ref_oks = [int(sw.step().sum().item()) for _ in range(TICKS)]
ref_table = sw.records_all().cpu().numpy().copy()
sw.close()
assert np.array_equal(res[0][1], res[1][1]), "ranks hold different tables"
assert np.array_equal(res[0][1], ref_table), "two-rank table differs from the single-process flight"
for r in range(WORLD):
    assert np.array_equal(res[r][2], ref_table[r * A_LOC:(r + 1) * A_LOC])
assert [a + b for a, b in zip(res[0][0], res[1][0])] == ref_oks and sum(ref_oks) >= 3 * TICKS
print("two-rank flight ok", ref_oks)
"""


@pytest.mark.own_device  # (a child process with two planners, ~20 streams: flown before this process holds queues of its own)
def test_two_rank_flight_on_one_gpu_matches_the_single_process_flight(pop, tmp_path):
    """The N > 1 tick on real kernels: two SwarmTick ranks (two host threads, one GPU, world size 2) fly 6 ticks with
    HipCompute, the publication inside the replan and the trajectory exchange through sogm_comm_* /
    sogm_traj_allgather — RCCL replaced by the in-process stand-in (tests/fake_rccl.cpp), torch.distributed by a
    thread rendezvous — and must end with the table, the own records and the per-tick ok counts of ONE process flying
    all 8 agents: rank-local overlays and deconfliction read the all-gathered table (one tick stale, like the ROS
    broadcast), consumers are ordered behind the collective inside the library only."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = tmp_path / "libfake_rccl.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC",
                           os.path.join(root, "tests", "fake_rccl.cpp"), "-o", str(so)])
    env = dict(os.environ, SOGM_RCCL_LIB=str(so), SOGM_REPO=root)
    r = subprocess.run([sys.executable, "-c", _TWO_RANK_FLIGHT], env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0 and "two-rank flight ok" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


_TWO_RANK_CHUNKED_FLIGHT = _TWO_RANK_FLIGHT[:_TWO_RANK_FLIGHT.index("dist = ThreadDist(WORLD)")] + r"""
dist = ThreadDist(WORLD)
GUARD = threading.Lock()
res, errs = {}, []
def run(rank):
    try:
        tls.rank = rank
        torch.cuda.set_device(0)
        with torch.cuda.stream(torch.cuda.Stream()):
            sw = driver.SwarmTick("parity", A_LOC, rank, WORLD, 0, dist=dist, moving_world=True, prestamp=False)
            sw.flight_guard = GUARD
            assert sw.exchange.active and sw.distributed
            ok_l, rec_l = sw.fly(TICKS)
            torch.cuda.current_stream().synchronize()
            res[rank] = (ok_l.cpu().numpy().copy(), rec_l.cpu().numpy().copy(), sw.all.cpu().numpy().copy(),
                         sw.own.cpu().numpy().copy())
            dist.bar.wait()          # nobody destroys its communicator while the other is still in a collective
            sw.close()
    except Exception as e:  # noqa: BLE001
        import traceback
        errs.append((rank, traceback.format_exc()))
        try: dist.bar.abort()
        except Exception: pass
ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(WORLD)]
[t.start() for t in ts]
[t.join(240) for t in ts]
if errs or any(t.is_alive() for t in ts):
    print("FAILED", errs, [t.is_alive() for t in ts], flush=True)
    os._exit(3)
# the same flight in one process, one six-tick call
sw = driver.SwarmTick("parity", A_LOC * WORLD, moving_world=True, prestamp=False)
ok_r, rec_r = sw.fly(TICKS)
torch.cuda.synchronize()
ok_r, rec_r, tab_r, own_r = ok_r.cpu().numpy(), rec_r.cpu().numpy(), sw.all.cpu().numpy().copy(), sw.own.cpu().numpy().copy()
sw.close()
assert ok_r.sum() >= 2 * TICKS
for r in range(WORLD):
    lo, hi = r * A_LOC, (r + 1) * A_LOC
    assert np.array_equal(res[r][0], ok_r[:, lo:hi]), ("ok flags differ", r)
    for k in range(TICKS):
        assert np.array_equal(res[r][1][k], rec_r[k, lo:hi]), ("per-tick records differ", r, k)
    assert np.array_equal(res[r][2], tab_r), ("last table differs", r)
    assert np.array_equal(res[r][3], own_r[lo:hi])
print("two-rank chunked flight ok", ok_r.sum(axis=1).tolist())
"""


@pytest.mark.own_device
def test_two_rank_flight_run_in_two_tick_calls_matches_the_single_process_flight(pop, tmp_path):
    """sogm_flight_run over several ranks (review item: the N > 1 stand-in on the flight kernels): two SwarmTick ranks as two
    host threads on one GPU fly 6 ticks through `fly()` — calls of two ticks with n_total = 8, agent0 = 0 / 4, each
    followed by the all-gather (stand-in RCCL behind the C ABI, in place) of the two table versions it finished — and
    must log, tick by tick, exactly the records of ONE process flying all 8 agents in a single six-tick call: the
    staleness rule (own record k - 1, neighbours' k - 2) makes the schedule free, the ranks included.  (The two ranks'
    calls are serialised by a lock: two flights at once on ONE device hold each other's compute-unit partitions.)"""
    import os
    import subprocess
    import sys
    assert "class ThreadDist" in _TWO_RANK_CHUNKED_FLIGHT
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = tmp_path / "libfake_rccl.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC",
                           os.path.join(root, "tests", "fake_rccl.cpp"), "-o", str(so)])
    env = dict(os.environ, SOGM_RCCL_LIB=str(so), SOGM_REPO=root, SOGM_FLIGHT_EXCHANGE="host")
    r = subprocess.run([sys.executable, "-c", _TWO_RANK_CHUNKED_FLIGHT], env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0 and "two-rank chunked flight ok" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


# Two ranks, ONE sogm_flight_run call of 20 ticks each, the exchange behind the call (SogmFlight::nccl_comm): per tick a
# device-side wait, the in-place all-gather of the rank's rows (stand-in RCCL) and the release of the overlays parked at the
# gate run on each context's exchange stream — no host step between ticks.  Both flights are in the air AT ONCE on one GPU:
# each takes two of the four shader engines (tuning keys flight_engines / flight_engine_first; 2 + 2 + 2 + 2 units).
_TWO_RANK_DEVICE_EXCHANGE_FLIGHT = _TWO_RANK_FLIGHT[:_TWO_RANK_FLIGHT.index("dist = ThreadDist(WORLD)")] + r"""
TICKS = 20
dist = ThreadDist(WORLD)
res, errs = {}, []
# Two ranks in ONE process share the pool of hardware queues, and a rank's stream holds a barrier packet for the whole length
# of its flight (the call's closing kernel waits for the four kernels): the OTHER rank's stream must not sit in the same
# queue behind it — each thread's stream is therefore created with a compute-unit mask (every unit), which gives it a
# hardware queue of its own.  (One rank per process, the deployment, has no such neighbour.)
torch.cuda.init()
_hip = C.CDLL(next(m.split()[-1] for m in open("/proc/self/maps") if "libamdhip64" in m))
_hip.hipExtStreamCreateWithCUMask.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
def own_queue_stream():
    st, mask = C.c_void_p(), (C.c_uint32 * 8)(*([0xFFFFFFFF] * 8))
    assert _hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, mask) == 0
    return torch.cuda.ExternalStream(st.value)
def run(rank):
    try:
        tls.rank = rank
        torch.cuda.set_device(0)
        with torch.cuda.stream(own_queue_stream()):
            tuning = {"flight_engines": 2, "flight_engine_first": 2 * rank, "flight_qp_units": 2, "flight_search_units": 2,
                      "flight_map_units": 2}
            sw = driver.SwarmTick("parity", A_LOC, rank, WORLD, 0, dist=dist, moving_world=True, prestamp=False, tuning=tuning)
            assert sw.exchange.active and sw.distributed
            sw.compute.prepare(0, TICKS + 1)     # the sensor frames, uploaded (a pageable upload synchronises)
            sw.planner.flight_prepare(len(sw.scene["cloud"]))   # queues and buffers of both flights exist before either is in the air
            dist.bar.wait()
            ok_l, rec_l = sw.fly(TICKS)          # ONE call: n_total = 8, agent0 = 4 rank, comm = the stand-in's
            torch.cuda.current_stream().synchronize()
            ms, hdr = sw.planner.flight_stats()
            assert hdr[pop._abi.FLIGHT_HDR_ERR] == 0 and hdr[pop._abi.FLIGHT_HDR_FINISHED] == A_LOC * TICKS, hdr.tolist()
            assert hdr[pop._abi.FLIGHT_HDR_LATE_WGS] == 0, hdr.tolist()
            res[rank] = (ok_l.cpu().numpy().copy(), rec_l.cpu().numpy().copy(), sw.all.cpu().numpy().copy(),
                         sw.own.cpu().numpy().copy(), sw._fl_tables.cpu().numpy().copy())
            dist.bar.wait()          # nobody destroys its communicator while the other is still in a collective
            sw.close()
    except Exception as e:  # noqa: BLE001
        import traceback
        errs.append((rank, traceback.format_exc()))
        try: dist.bar.abort()
        except Exception: pass
ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(WORLD)]
[t.start() for t in ts]
[t.join(240) for t in ts]
if errs or any(t.is_alive() for t in ts):
    print("FAILED", errs, [t.is_alive() for t in ts], flush=True)
    os._exit(3)
# the same flight in one process, one twenty-tick call
sw = driver.SwarmTick("parity", A_LOC * WORLD, moving_world=True, prestamp=False)
ok_r, rec_r = sw.fly(TICKS)
torch.cuda.synchronize()
ok_r, rec_r, tab_r, own_r = ok_r.cpu().numpy(), rec_r.cpu().numpy(), sw.all.cpu().numpy().copy(), sw.own.cpu().numpy().copy()
ring_r = sw._fl_tables.cpu().numpy().copy()
sw.close()
assert ok_r.sum() >= 2 * TICKS
for r in range(WORLD):
    lo, hi = r * A_LOC, (r + 1) * A_LOC
    assert np.array_equal(res[r][0], ok_r[:, lo:hi]), ("ok flags differ", r)
    for k in range(TICKS):
        assert np.array_equal(res[r][1][k], rec_r[k, lo:hi]), ("per-tick records differ", r, k)
    assert np.array_equal(res[r][2], tab_r), ("last table differs", r)
    assert np.array_equal(res[r][3], own_r[lo:hi])
    assert np.array_equal(res[r][4], ring_r), ("the ring of table versions differs", r)   # every rank holds every row of every version
print("two-rank device-exchange flight ok", ok_r.sum(axis=1).tolist())
"""


@pytest.mark.own_device
def test_two_rank_flight_with_the_exchange_behind_the_call_matches_the_single_process_flight(pop, tmp_path):
    """a multi-rank flight that does not go back to the host every two ticks.  Two SwarmTick ranks as two
    host threads on one GPU each fly 20 ticks in ONE sogm_flight_run call (n_total = 8, agent0 = 0 / 4, nccl_comm = the
    stand-in RCCL's communicator); the table versions are exchanged by collectives queued behind the call, the gate of tick
    k's overlay is the all-gather of ver(k - 2).  Records, ok flags, own records, the last table and the whole ring of four
    versions equal ONE process flying all 8 agents in a single call.  (Reference: the drones never wait for each other and
    read whatever arrived last, plan_manager.cpp:92-233, particles.cpp:179-190.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = tmp_path / "libfake_rccl.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC",
                           os.path.join(root, "tests", "fake_rccl.cpp"), "-o", str(so)])
    env = dict(os.environ, SOGM_RCCL_LIB=str(so), SOGM_REPO=root)
    r = subprocess.run([sys.executable, "-c", _TWO_RANK_DEVICE_EXCHANGE_FLIGHT], env=env, capture_output=True, text=True,
                       timeout=500)
    assert r.returncode == 0 and "two-rank device-exchange flight ok" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
