"""Summarise rocprofv3 rocpd (.db) outputs into small text files for profiles/.
usage: rocprof_summary.py <trace.db> [<pmc.db> ...] > summary.md"""
import sqlite3
import sys


def short(name):
    n = name.split("(")[0]
    return n[-60:]


def pmc_only():
    for pmc in sys.argv[2:]:
        c2 = sqlite3.connect(pmc).cursor()
        print(f"\n## PMC pass {pmc.split('/')[-1]} (separate run, --pmc only)\n")
        print("| kernel | counter | dispatches | mean value (KB) | mean duration us |")
        print("|---|---|---|---|---|")
        q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration)/1000.0 from counters_collection "
             "where kernel_name like '%sogm::%' group by kernel_name, counter_name")
        for kn, cn, n, v, d in c2.execute(q):
            print(f"| {short(kn)} | {cn} | {n} | {v:.1f} | {d:.1f} |")


def main():
    trace = sys.argv[1]
    if trace == "-":  # PMC passes only
        pmc_only()
        return
    con = sqlite3.connect(trace)
    cur = con.cursor()
    print("## kernel-trace stats (rocprofv3 --kernel-trace --stats), durations in ms\n")
    print("| kernel | calls | total_ms | avg_ms | % |")
    print("|---|---|---|---|---|")
    for name, calls, tot, avg, pct in cur.execute("select * from top_kernels"):
        if pct < 0.01:
            continue
        print(f"| {short(name)} | {calls} | {tot/1e3:.1f} | {avg/1e3:.1f} | {pct:.2f} |")
    print("\n### per-dispatch durations of the SOGM kernels (us)\n")
    rows = cur.execute("select name, (end-start)/1000.0 from kernels order by start").fetchall()
    by = {}
    for n, d in rows:
        if "sogm::" in n or "k_pack" in n:
            by.setdefault(short(n), []).append(d)
    for k, v in by.items():
        if len(v) > 64:  # keep the file readable: distribution instead of every dispatch
            w = sorted(v)
            print(f"- {k}: n={len(w)} min {w[0]:.0f} median {w[len(w)//2]:.0f} mean {sum(w)/len(w):.0f} p90 {w[int(len(w)*0.9)]:.0f} max {w[-1]:.0f}")
        else:
            print(f"- {k}: " + ", ".join(f"{x:.0f}" for x in v))
    for pmc in sys.argv[2:]:
        c2 = sqlite3.connect(pmc).cursor()
        print(f"\n## PMC pass {pmc.split('/')[-1]} (separate run, --pmc only)\n")
        print("| kernel | counter | dispatches | mean value (KB) | mean duration us |")
        print("|---|---|---|---|---|")
        q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration)/1000.0 from counters_collection "
             "where kernel_name like '%sogm::%' group by kernel_name, counter_name")
        for kn, cn, n, v, d in c2.execute(q):
            print(f"| {short(kn)} | {cn} | {n} | {v:.1f} | {d:.1f} |")


main()
