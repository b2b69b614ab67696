#!/usr/bin/env python
"""Kernel timeline of the last lock-step ticks in a rocprofv3 --kernel-trace database: every kernel between two
k_cull_cylinders launches (the update's first kernel), start and end in ms after that launch's start.
    python tools/tick_timeline.py <results.db> [ticks from the end, default 2]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cols = [d[0] for d in con.execute("select * from kernels limit 1").description]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = con.execute(f"select name, start, end, {q}, vgpr_count, accum_vgpr_count, lds_size, grid_x * grid_y * grid_z / (workgroup_x * workgroup_y * workgroup_z), workgroup_x from kernels order by start").fetchall()
culls = [i for i, r in enumerate(rows) if "k_cull_cylinders" in r[0]]
for j in range(max(0, len(culls) - n - 1), len(culls) - 1):
    i0, i1 = culls[j], culls[j + 1]
    t0 = rows[i0][1]
    # kernels launched shortly before the cull belong to the tick too (sogm_tick_inputs)
    k = i0
    while k > 0 and t0 - rows[k - 1][1] < 100000:
        k -= 1
    print(f"--- tick starting at kernel {i0}: {(rows[i1][1] - t0) / 1e6:.3f} ms until the next update")
    for name, s, e, qq, vg, ag, lds, wgs, wx in rows[k:i1]:
        short = name.replace("sogm::", "").split("(")[0][:60]
        print(f"  {(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f}  ({(e - s) / 1e3:8.1f} us)  q{qq}  {wgs:6d} x {wx:4d}  vgpr {vg}+{ag} lds {lds:6d}  {short}")
