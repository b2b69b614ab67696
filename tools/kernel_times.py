#!/usr/bin/env python
"""Mean duration (us) per kernel over the lock-step ticks of a rocprofv3 --kernel-trace database (first 4 ticks skipped),
and the mean tick (update's first kernel to the next one).   python tools/kernel_times.py <results.db> [name filter]"""
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = con.execute("select name, start, end from kernels order by start").fetchall()
culls = [i for i, r in enumerate(rows) if "k_cull_cylinders" in r[0]]
acc = defaultdict(list)
ticks = []
for j in range(4, len(culls) - 1):
    ticks.append((rows[culls[j + 1]][1] - rows[culls[j]][1]) / 1e6)
    per = defaultdict(float)
    for name, s, e in rows[culls[j]:culls[j + 1]]:
        per[name.replace("sogm::", "").replace("void ", "").split("(")[0]] += (e - s) / 1e3
    for k, v in per.items():
        acc[k].append(v)
print(f"{len(ticks)} ticks, mean {sum(ticks) / len(ticks):.3f} ms, median {sorted(ticks)[len(ticks) // 2]:.3f} ms")
for k in sorted(acc, key=lambda k: -sum(acc[k])):
    if flt in k:
        print(f"  {sum(acc[k]) / len(acc[k]):9.1f} us  {k}")
