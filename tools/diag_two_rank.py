"""Diagnostic: the two-rank flight of tests/test_exchange_gpu.py (two host threads, stand-in RCCL) against the
single-process flight, tick by tick (device clones in stream order, no host synchronisation inside the flights):
prints the first tick / agent whose record or ok flag differs.   SOGM_RCCL_LIB must name the stand-in library."""
import ctypes as C, importlib, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
planner = importlib.import_module("pred-occ-planner_amd.planner")
WORLD, A_LOC, TICKS = 2, 4, 6
tls = threading.local()
class ThreadDist:
    def __init__(self, world):
        self.world, self.bar, self.box, self.shared = world, threading.Barrier(world), [None] * world, None
    def is_initialized(self): return True
    def get_backend(self): return "nccl"
    def broadcast_object_list(self, lst, src=0):
        if tls.rank == src: self.shared = list(lst)
        self.bar.wait(); lst[:] = self.shared; self.bar.wait()
    def all_gather_object(self, out, obj):
        self.box[tls.rank] = obj
        self.bar.wait(); out[:] = list(self.box); self.bar.wait()
lib = pop.lib()
lib.sogm_debug_planner_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
lib.sogm_debug_copy_grid.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
def internals(sw, A):
    mf = sw.planner.pp.max_faces
    shapes = {0: ((A, 16, mf, 4), np.float64), 1: ((A, 16), np.int32), 2: ((A,), np.int32), 3: ((A, 6), np.float64),
              4: ((A, 64, 6), np.float64), 5: ((A,), np.int32), 7: ((A,), np.int32), 9: ((A,), np.int32), 10: ((A, 4), np.int32)}
    out = {}
    for k, (shp, dt) in shapes.items():
        a = np.zeros(shp, dt)
        rc = lib.sogm_debug_planner_buffer(sw.planner._p, k, a.ctypes.data_as(C.c_void_p), a.nbytes)
        assert rc == 0, (k, rc)
        out[k] = a
    return out
NAMES = {0: "polytopes", 1: "faces", 2: "npoly", 3: "local goal", 4: "route", 5: "route length", 7: "QP iterations", 9: "search ret"}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    dist = ThreadDist(WORLD)
    res, errs, ints, grids, dgr = {}, [], {}, {}, {}
    def run(rank):
        try:
            tls.rank = rank
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                sw = driver.SwarmTick("parity", A_LOC, rank, WORLD, 0, dist=dist)
                hist = []
                for _ in range(TICKS):
                    ok = sw.step()
                    if os.environ.get("GRIDS"):
                        gs = []
                        for a in range(A_LOC):
                            t_ = torch.empty(sw.spec.T * sw.spec.L * sw.spec.W * sw.spec.H, dtype=torch.float32, device="cuda")
                            lib.sogm_debug_copy_grid(sw.map.ctx, a, C.c_void_p(t_.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
                            gs.append(t_)
                        dgr.setdefault(rank, []).append(gs)
                    st_ = np.zeros((A_LOC, 16), np.int64)
                    if os.environ.get("SYNC_TICKS"):
                        torch.cuda.current_stream().synchronize()
                        pop.lib().sogm_debug_qp_stats(sw.planner._p, st_.ctypes.data_as(C.c_void_p))
                        ints.setdefault(rank, []).append(internals(sw, A_LOC))
                        grids.setdefault(rank, []).append([sw.map.download(a) for a in range(A_LOC)])
                    hist.append((ok, sw.new.clone(), sw.own.clone(), sw.pva.clone(), sw.poses.clone(), sw.all.clone(), torch.from_numpy(st_)))
                torch.cuda.current_stream().synchronize()
                res[rank] = ([tuple(x.cpu().numpy() for x in h) for h in hist], sw.planner.flow_failures())
                dist.bar.wait()
                sw.close()
        except Exception:
            import traceback
            extra = ""
            try:  # what the device saw (sogm_planner_flow_failures + the control block's header)
                hdr = np.zeros(11, np.int32)
                lib.sogm_debug_flow_peek(sw.planner._p, hdr.ctypes.data_as(C.c_void_p), hdr.size)
                extra = f" flow_failures {sw.planner.flow_failures()} flow header {hdr.tolist()}"
            except Exception as e2:
                extra = f" (no flow state: {e2!r})"
            errs.append((rank, traceback.format_exc()[-600:] + extra))
            try: dist.bar.abort()
            except Exception: pass
    ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(WORLD)]
    [t.start() for t in ts]
    [t.join(240) for t in ts]
    if errs:
        print("FAILED", errs); continue
    sw = driver.SwarmTick("parity", A_LOC * WORLD)
    ref, ref_grids, ref_dgr = [], [], []
    for _ in range(TICKS):
        ok = sw.step()
        st_ = np.zeros((A_LOC * WORLD, 16), np.int64)
        torch.cuda.synchronize()
        pop.lib().sogm_debug_qp_stats(sw.planner._p, st_.ctypes.data_as(C.c_void_p))
        ref_grids.append([sw.map.download(a) for a in range(A_LOC * WORLD)] if os.environ.get('SYNC_TICKS') else None)
        if os.environ.get('GRIDS'):
            gs = []
            for a in range(A_LOC * WORLD):
                t_ = torch.empty(sw.spec.T * sw.spec.L * sw.spec.W * sw.spec.H, dtype=torch.float32, device='cuda')
                lib.sogm_debug_copy_grid(sw.map.ctx, a, C.c_void_p(t_.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
                gs.append(t_)
            ref_dgr.append(gs)
        ref.append(tuple(x.cpu().numpy().copy() for x in (ok, sw.new, sw.own, sw.pva, sw.poses)) + (sw.all.cpu().numpy().copy(), st_, internals(sw, A_LOC * WORLD)))
    sw.close()
    msg = "identical"
    for k in range(TICKS):
        for r in range(WORLD):
            for a in range(A_LOC):
                g = r * A_LOC + a
                names = ("ok", "new record", "own record", "start state", "map centre")
                bad = [names[i] for i in range(5) if not np.array_equal(res[r][0][k][i][a], ref[k][i][g])]
                if bad:
                    r0 = planner.records_from_bytes(res[r][0][k][1][a:a + 1])[0]
                    r1 = planner.records_from_bytes(ref[k][1][g:g + 1])[0]
                    d = np.abs(np.array(r0.cpts[:]) - np.array(r1.cpts[:])).max()
                    tab_same = [bool(np.array_equal(res[r][0][kk][5], res[1 - r][0][kk][5])) for kk in range(TICKS)]
                    qa, qb = res[r][0][k][6][a], ref[k][6][g]
                    for kk in range(k + 1):
                        tb = [int(q) for q in range(A_LOC * WORLD) if not np.array_equal(res[r][0][kk][5][q], ref[kk][5][q])]
                        if tb: print(f"   table AFTER tick {kk}: rank {r}'s copy differs from the single-process table for agents {tb}")
                    if dgr:
                        V = sw.spec.L * sw.spec.W * sw.spec.H
                        for kk in range(k + 1):
                            for rr in range(WORLD):
                                for aa in range(A_LOC):
                                    x, y = dgr[rr][kk][aa].cpu().numpy(), ref_dgr[kk][rr * A_LOC + aa].cpu().numpy()
                                    if not np.array_equal(x, y):
                                        idx = np.flatnonzero(x != y)
                                        print(f"   MAP of tick {kk} agent {rr * A_LOC + aa}: {len(idx)} cells differ; (slice, voxel, two-rank value, one-process value): "
                                              f"{[(int(i // V), int(i % V), float(x[i]), float(y[i])) for i in idx[:6]]}")
                    if grids:
                        for kk in range(k + 1):
                            for rr in range(WORLD):
                                for aa in range(A_LOC):
                                    x, y = grids[rr][kk][aa], ref_grids[kk][rr * A_LOC + aa]
                                    if not np.array_equal(x, y):
                                        idx = np.argwhere(x != y)
                                        print(f"   MAP of tick {kk} agent {rr * A_LOC + aa}: {len(idx)} cells differ; first (voxel, slice) {idx[:4].tolist()} values {[(float(x[i, j]), float(y[i, j])) for i, j in idx[:4]]}")
                    print("   QP iterations / refactorisations / checks:", qa[6], qa[3], qa[5], "vs", qb[6], qb[3], qb[5])
                    if rank_ints := ints.get(r):
                        X, Y = rank_ints[k], ref[k][7]
                        M = int(Y[2][g])
                        print("   npoly", X[2][a], Y[2][g], "faces", X[1][a][:max(M, 0)].tolist(), Y[1][g][:max(M, 0)].tolist(), "route len", X[5][a], Y[5][g])
                        for i in range(max(M, 0)):
                            nf = int(min(X[1][a][i], Y[1][g][i]))
                            dpl = np.abs(X[0][a, i, :nf] - Y[0][g, i, :nf]).max() if nf else 0.0
                            if dpl or X[1][a][i] != Y[1][g][i]:
                                print(f"   polytope {i}: faces {X[1][a][i]} vs {Y[1][g][i]}, max |d| over common faces {dpl:.3e}")
                        L = int(min(X[5][a], Y[5][g]))
                        print("   search ret", X[9][a], Y[9][g], "stats", X[10][a].tolist(), Y[10][g].tolist())
                        np.set_printoptions(precision=4, suppress=True, linewidth=200)
                        print("   route (two-rank):", X[4][a, :L, :3].round(4).tolist())
                        print("   route (one proc):", Y[4][g, :L, :3].round(4).tolist())
                        print("   route max |d|", np.abs(X[4][a, :L] - Y[4][g, :L]).max() if L else 0.0, "goal max |d|", np.abs(X[3][a] - Y[3][g]).max())
                        for key, nm in {}.items():
                            x, y = rank_ints[k][key][a], ref[k][7][key][g]
                            if not np.array_equal(x, y):
                                dd = np.abs(x.astype(np.float64) - y.astype(np.float64))
                                print(f"   stage output '{nm}' differs: max |d| {dd.max():.3e} at {np.unravel_index(dd.argmax(), dd.shape)}")
                    msg = (f"tick {k} rank {r} agent {g}: differs in {bad}; ok {res[r][0][k][0][a]} vs {ref[k][0][g]}; pieces "
                           f"{r0.n_pieces} vs {r1.n_pieces}; max |d cpts| {d:.3e}; ranks' tables equal per tick {tab_same}")
                    break
            if msg != "identical": break
        if msg != "identical": break
    print(f"rep {rep}: {msg}; flow failures {[res[r][1] for r in range(WORLD)]}")
