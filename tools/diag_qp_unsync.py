"""Diagnostic: the QP's ADMM iteration inside ticks that are NOT host-synchronised (the bench's timed region) against
host-synchronised ticks (tools/diag_qp_time.py, the bench's sustained block): blocks of `n` ticks, the solve clocks of the
block's last tick (QpWorkspace::dbg) for every solve of >= 1000 iterations.

    python tools/diag_qp_unsync.py [blocks=6] [ticks per block=5]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
A = 128
sw = driver.SwarmTick("cfg2", A)
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib = pop.lib()
lib.sogm_debug_qp_stats.argtypes = [C.c_void_p, C.c_void_p]
def last_tick():
    st = np.zeros((A, 16), np.int64)
    lib.sogm_debug_qp_stats(sw.planner._p, st.ctypes.data_as(C.c_void_p))
    it = st[:, 6]
    plain = (st[:, 0] - st[:, 1] - st[:, 2] - st[:, 4]) / 100.0
    big = it >= 1000
    return plain[big].sum() / max(it[big].sum(), 1), int(big.sum()), st[big, 4].sum() / 100.0 / max(st[big, 5].sum(), 1)
for _ in range(3):
    sw.step()
torch.cuda.synchronize()
for b in range(blocks):
    for mode in ("unsynchronised", "synchronised"):
        for _ in range(n):
            sw.step()
            if mode == "synchronised":
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        us, cnt, chk = last_tick()
        print(f"block {b} {mode:15s}: {us:.3f} us per iteration, {chk:.2f} us per check over the {cnt} solves of >= 1000 iterations in the block's last tick")
print("flow failures", sw.planner.flow_failures())
sw.close()
