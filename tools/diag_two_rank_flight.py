"""Diagnostic: tests/test_exchange_gpu.py's two-rank chunked flight (two host threads, two-tick calls, stand-in RCCL) against
the single-process six-tick flight, with the comparison spelled out: which rank / tick / agent differs first.
SOGM_RCCL_LIB must name the stand-in library (hipcc -shared tests/fake_rccl.cpp)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "tests"))
os.environ.setdefault("SOGM_REPO", root)
import test_exchange_gpu as T
src = T._TWO_RANK_CHUNKED_FLIGHT
cut = src.index("assert ok_r.sum() >= 2 * TICKS")
src = src[:cut].replace("            torch.cuda.current_stream().synchronize()\n            res[rank]", "            torch.cuda.current_stream().synchronize()\n            print('rank', rank, 'flight hdr', sw.planner.flight_stats()[1][:15].tolist(), 'flow failures', sw.planner.flow_failures(), flush=True)\n            res[rank]") + r'''
bad = False
for r in range(WORLD):
    lo, hi = r * A_LOC, (r + 1) * A_LOC
    if not np.array_equal(res[r][0], ok_r[:, lo:hi]):
        bad = True
        print("rank", r, "ok flags differ:\n", res[r][0], "\n", ok_r[:, lo:hi])
    for k in range(TICKS):
        d = (res[r][1][k] != rec_r[k, lo:hi]).any(axis=1)
        if d.any():
            bad = True
            print("rank", r, "tick", k, "records differ for local agents", np.flatnonzero(d).tolist())
    if not np.array_equal(res[r][2], tab_r):
        bad = True
        print("rank", r, "last table differs in rows", np.flatnonzero((res[r][2] != tab_r).any(axis=1)).tolist())
print("DIFFERENT" if bad else "identical", ok_r.sum(axis=1).tolist())
'''
exec(compile(src, "two_rank_chunked", "exec"))
