"""Timeline of one tick from a rocprofv3 kernel trace (rocpd .db): start offset / duration of every SOGM kernel."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
# a tick starts at each k_stamp_cloud
mark = "k_tick_inputs" if any("k_tick_inputs" in r[0] for r in rows) else "k_stamp_cloud"
if any("k_prestamp_flow" in r[0] for r in rows):  # pre-stamped ticks have no k_tick_inputs: a tick = replan to replan
    mark = "k_flow_reset"
stamps = [i for i, r in enumerate(rows) if mark in r[0]]
k = len(stamps) // 2  # a tick from the middle of the timed region (the run ends with stage-pass updates)
a, b = stamps[k], stamps[k + 1]
t0 = rows[a][1]
print(f"tick length {(rows[b][1] - t0) / 1e6:.2f} ms")
show_all = len(sys.argv) > 2 and sys.argv[2] == "--all"  # every kernel, incl. the host framework's small ones
for name, s, e in rows[a:b]:
    n = name.split("(")[0].split("::")[-1][:28]
    if show_all:
        n = name.replace("void ", "")[:60]
        print(f"{n:60s} start {(s - t0) / 1e6:7.3f}  dur {(e - s) / 1e6:7.3f}  end {(e - t0) / 1e6:7.3f}")
    elif (e - s) > 150e3 or "k_" in n:
        print(f"{n:28s} start {(s - t0) / 1e6:7.2f}  dur {(e - s) / 1e6:7.2f}  end {(e - t0) / 1e6:7.2f}")
