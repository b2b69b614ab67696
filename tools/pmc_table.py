#!/usr/bin/env python
"""Per kernel: mean value of every counter in a rocprofv3 --pmc database.   python tools/pmc_table.py <results.db> [name filter]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration)/1000.0 from counters_collection "
     "group by kernel_name, counter_name order by kernel_name")
for kn, cn, n, v, d in con.execute(q):
    short = kn.replace("sogm::", "").replace("void ", "").split("(")[0]
    if flt in short:
        print(f"{short[:32]:32s} {cn:40s} n {n:4d}  avg {v:16.1f}  kernel {d:9.1f} us")
