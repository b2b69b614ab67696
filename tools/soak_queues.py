"""Soak: the full-size tick (128 agents, 200^3 x 20) flown N times for 8 ticks each WITHOUT host synchronisation, under
the environment it is started in (GPU_MAX_HW_QUEUES=4, SOGM_TUNING=...): counts flights with a failed tick."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
driver = importlib.import_module("pred-occ-planner_amd.driver")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bad = 0
for i in range(N):
    sw = driver.SwarmTick("cfg2", 128)
    t0 = time.perf_counter()
    try:
        for _ in range(8):
            sw.step()
        torch.cuda.synchronize()
        ff = sw.planner.flow_failures()
    except Exception as e:
        ff = ("exception", repr(e)[:120])
    dt = (time.perf_counter() - t0) / 8 * 1e3
    if ff != (0, 0):
        bad += 1
        import ctypes as C
        hdr = np.zeros(11, np.int32)
        importlib.import_module("pred-occ-planner_amd").lib().sogm_debug_flow_peek(sw.planner._p, hdr.ctypes.data_as(C.c_void_p), hdr.size)
        print(f"flight {i}: FAILED {ff} ({dt:.1f} ms per tick) flow header {hdr.tolist()}")
    sw.close()
print(f"{N - bad} of {N} flights clean ({os.environ.get('GPU_MAX_HW_QUEUES')} queues, tuning {os.environ.get('SOGM_TUNING')})")
