"""From a rocprofv3 kernel trace (rocpd .db) of bench.py: for every tick of the dataflow replan, when its last QP, its
finishing kernel, its pre-stamp and the next tick's overlay ended, and when the next tick began (ms from the tick's
k_flow_reset) — the tick's tail behind the slowest agent's chain."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "k_flow_reset" in r[0]]
print("tick | length | astar end | corridor_flow end | qp_flow end | finish end | prestamp end | overlay end | report end")
for a, b in zip(starts[:-1], starts[1:]):
    t0 = rows[a][1]
    def end(tag):
        v = [e for n, s, e in rows[a:b + 40] if tag in n and s >= t0 and s < rows[b][1] + 2e6]
        return (v[0] - t0) / 1e6 if v else float("nan")
    print(f"{starts.index(a):4d} | {(rows[b][1] - t0) / 1e6:6.2f} | {end('k_astar'):5.2f} | {end('k_corridor_flow'):5.2f} | {end('k_qp_flow'):5.2f} | "
          f"{end('k_finish_flow'):5.2f} | {end('k_prestamp_flow'):5.2f} | {end('k_splat_neighbours'):5.2f} | {end('k_flow_report'):5.2f}")
