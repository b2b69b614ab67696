"""Diagnostic: single-process flights (sogm_flight_run) against the same staleness rule flown lock-step, several swarm
sizes and repetitions: which ticks differ.  python tools/diag_flight_vs_lockstep.py"""
import importlib, sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
import test_flight_gpu as T
for A, K in ((8, 6), (8, 6), (6, 10), (8, 12)):
    ok_l, rec_l, own_l, cnt_l = T._lockstep_lag2(driver, "parity", A, K)
    for rep in range(3):
        ok_f, rec_f, own_f, last_f, cnt_f, ms = T._flight(driver, "parity", A, [K])
        bad = [k for k in range(K) if not np.array_equal(rec_f[k], rec_l[k])]
        print(A, K, rep, "ok equal", np.array_equal(ok_f, ok_l), "bad ticks", bad, flush=True)
