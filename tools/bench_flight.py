#!/usr/bin/env python
"""Flights (sogm_flight_run) of the bench swarm, timed: ms per tick of the whole swarm, per-agent stage sums, for a list of
tuning variants (each in a fresh context).  python tools/bench_flight.py [ticks] ["k=v,k=v" ...]"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np
import torch

pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")


def run(tuning, ticks, grid="cfg2", A=None, warm=3, chunk=20):
    A = A or pop.config.AGENTS[grid]
    sw = driver.SwarmTick(grid, A, moving_world=True, prestamp=False, grids=1, tuning=tuning)
    sw.compute.prepare(0, warm + ticks + 1)
    sw.fly(warm)
    torch.cuda.synchronize()
    oks, t0 = [], time.perf_counter()
    acc = np.zeros(8)
    left = ticks
    per_chunk = []
    while left > 0:
        n = min(chunk, left)
        t1 = time.perf_counter()
        ok, _ = sw.fly(n)
        torch.cuda.synchronize()
        per_chunk.append((time.perf_counter() - t1) / n * 1e3)
        oks.append(ok)
        ms, hdr = sw.planner.flight_stats()
        assert hdr[pop._abi.FLIGHT_HDR_ERR] == 0, hdr
        acc += ms.sum(axis=0)
        prof = prof + hdr[16:32].astype(np.float64) if "prof" in dir() else hdr[16:32].astype(np.float64)
        left -= n
    dt = time.perf_counter() - t0
    n_ok = int(torch.cat(oks).sum().item())
    out = {"tuning": tuning, "ticks": ticks, "ms_per_tick": dt / ticks * 1e3, "replans_per_s": A * ticks / dt,
           "ok_fraction": n_ok / (A * ticks), "chunk_ms_per_tick": [round(x, 3) for x in per_chunk],
           "per_agent_tick_ms": dict(zip(pop._abi.FLIGHT_STAT_NAMES[:7], (acc[:7] / acc[7]).round(3).tolist())),
           "map_wave_ms_per_tick": dict(zip(("idle", "reset", "bits", "marks", "splat"), (prof[:5] / 100.0 / ticks).round(1).tolist())),
           "light_wave_ms_per_tick": dict(zip(("idle", "corridor", "finish"), (prof[6:9] / 100.0 / ticks).round(1).tolist())),
           "wgs": None}
    sw.close()
    return out


if __name__ == "__main__":
    ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    variants = sys.argv[2:] or [""]
    for v in variants:
        tuning = {k: float(x) for k, x in (kv.split("=") for kv in v.split(",") if kv)}
        print("FLIGHT " + json.dumps(run(tuning, ticks, grid=os.environ.get("GRID", "cfg2"))), flush=True)
