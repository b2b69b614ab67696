"""Diagnostic: run-to-run determinism of the closed-loop flight (parity grid): the same flight N times in one process;
prints where the own-record tables first differ (tick, agent, what)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
planner = importlib.import_module("pred-occ-planner_amd.planner")
A, TICKS, N = int(os.environ.get("A", 8)), int(os.environ.get("TICKS", 6)), int(sys.argv[1]) if len(sys.argv) > 1 else 6
grid = os.environ.get("GRID", "parity")
ref = None
for run in range(N):
    sw = driver.SwarmTick(grid, A)
    hist = []
    for k in range(TICKS):
        ok = sw.step()
        it = np.zeros((A, 16), np.int64)
        if not os.environ.get("NOSYNC"):
            torch.cuda.synchronize()
            import ctypes as C
            pop.lib().sogm_debug_qp_stats(sw.planner._p, it.ctypes.data_as(C.c_void_p))
        hist.append((ok, sw.new.clone(), it[:, 6].copy(), it[:, 3].copy()))
    torch.cuda.synchronize()
    hist = [(h[0].cpu().numpy().copy(), h[1].cpu().numpy().copy(), h[2], h[3]) for h in hist]
    if True:
        pass
    ff = sw.planner.flow_failures()
    sw.close()
    if ref is None:
        ref = hist
        print("run 0: ok per tick", [int(h[0].sum()) for h in hist], "flow failures", ff)
        continue
    msg = "identical"
    for k in range(TICKS):
        for a in range(A):
            if ref[k][0][a] != hist[k][0][a] or not np.array_equal(ref[k][1][a], hist[k][1][a]):
                r0 = planner.records_from_bytes(ref[k][1][a:a + 1])[0]
                r1 = planner.records_from_bytes(hist[k][1][a:a + 1])[0]
                d = np.abs(np.array(r0.cpts[:]) - np.array(r1.cpts[:])).max()
                msg = (f"tick {k} agent {a}: ok {ref[k][0][a]} vs {hist[k][0][a]}, pieces {r0.n_pieces} vs {r1.n_pieces}, "
                       f"max |d cpts| {d:.3e}, QP iterations {ref[k][2][a]} vs {hist[k][2][a]}, refactorisations {ref[k][3][a]} vs {hist[k][3][a]}")
                break
        if msg != "identical":
            break
    print(f"run {run}: {msg}; flow failures {ff}")
