"""Per-kernel mean of every PMC counter in a rocprofv3 rocpd database (diagnostic)."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if not view:
    print("tables:", tabs); sys.exit(0)
cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
kn = "kernel_name" if "kernel_name" in cols else "name"
rows = cur.execute(f"select {kn}, counter_name, avg(value), count(*) from {view} group by {kn}, counter_name").fetchall()
for k, c, v, n in sorted(rows):
    print(f"{k.split('(')[0][-40:]:40s} {c:28s} n={n:4d} mean={v:14.1f}")
