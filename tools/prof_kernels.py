"""Print per-kernel call counts / average durations from a rocprofv3 rocpd database (kernel-trace)."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select * from top_kernels").fetchall()
for name, calls, tot, avg, pct in rows:
    if pct > float(sys.argv[2] if len(sys.argv) > 2 else 0.3):
        print(f"{name.split('(')[0][-44:]:44s} calls {calls:5d} avg_us {avg/1e3:9.1f} total_ms {tot/1e6:9.2f} pct {pct:5.1f}")
