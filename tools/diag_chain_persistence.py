#!/usr/bin/env python
"""How predictable is the tick's critical agent?  Flies the bench swarm lock-step (plain update), records every agent's
chain length (search start -> finished) per tick, and reports how often the agents with the longest chains of tick k were
among the K longest of tick k - 1 (what a priority order for the map update could exploit).
    python tools/diag_chain_persistence.py [ticks]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
A = 128
sw = driver.SwarmTick("cfg2", A, moving_world=True, prestamp=False)
sw.compute.prepare(0, N + 2)
lib = pop.lib()
chains, parts = [], []
for k in range(N):
    sw.step()
    torch.cuda.synchronize()
    ts = np.zeros((A, 8), np.int64)
    lib.sogm_debug_flow_times(sw.planner._p, ts.ctypes.data_as(C.c_void_p))
    chains.append((ts[:, 6] - ts[:, 0]) / 1e5)
    parts.append(np.stack([(ts[:, 1] - ts[:, 0]), (ts[:, 3] - ts[:, 1]), (ts[:, 5] - ts[:, 3])], 1) / 1e5)
ch = np.stack(chains)[10:]
pa = np.stack(parts)[10:]
print(f"{len(ch)} ticks; chain ms mean {ch.mean():.2f}, per-tick max mean {ch.max(1).mean():.2f}; agents within 1.0 ms of the tick's max: "
      f"mean {(ch > ch.max(1, keepdims=True) - 1.0).sum(1).mean():.1f}, within 0.5 ms: {(ch > ch.max(1, keepdims=True) - 0.5).sum(1).mean():.1f}")
for K in (8, 16, 32, 48, 64):
    hit1 = hit_all = 0
    gain = []
    for k in range(1, len(ch)):
        prev_top = set(np.argsort(-ch[k - 1])[:K].tolist())
        crit = int(np.argmax(ch[k]))
        hit1 += crit in prev_top
        near = np.flatnonzero(ch[k] > ch[k].max() - 1.0)
        hit_all += all(int(a) in prev_top for a in near)
        # tick length if batch A (prev_top) starts 1.1 * (1 - K / A) ms earlier than the rest, which start 0.05 ms later
        early = 1.1 * (1 - K / A)
        inA = np.array([a in prev_top for a in range(A)])
        gain.append(ch[k].max() - max((ch[k][inA] - early).max(), (ch[k][~inA] + 0.05).max()))
    print(f"K = {K:3d}: critical agent in the previous tick's top K: {hit1 / (len(ch) - 1):.2f}; ALL agents within 1 ms of the max in it: "
          f"{hit_all / (len(ch) - 1):.2f}; modelled gain per tick {np.mean(gain):.3f} ms")
# other predictors: the previous tick's QP time / the agent's mean chain so far
print("correlation of chain(k) with chain(k-1):", np.corrcoef(ch[1:].ravel(), ch[:-1].ravel())[0, 1].round(3),
      "| of QP(k) with QP(k-1):", np.corrcoef(pa[1:, :, 2].ravel(), pa[:-1, :, 2].ravel())[0, 1].round(3))
