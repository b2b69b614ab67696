"""Diagnostic: where the QP stage of the bench tick spends its time, agent by agent (QpWorkspace::dbg through
sogm_debug_qp_stats): set-up, refactorisations, checks and the plain ADMM iterations of every agent's solve inside the
dataflow replan, over a few ticks of the bench workload.

    python tools/diag_qp_time.py [ticks=10] [print every n-th tick=3]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
A = 128
sw = driver.SwarmTick("cfg2", A)
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 10
every = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lib = pop.lib()
lib.sogm_debug_qp_stats.argtypes = [C.c_void_p, C.c_void_p]
tot = np.zeros(8)
fac = np.zeros(4)
chk = np.zeros(3)
stp = np.zeros(5)
for k in range(ticks):
    sw.step()
    torch.cuda.synchronize()
    st = np.zeros((A, 16), np.int64)
    lib.sogm_debug_qp_stats(sw.planner._p, st.ctypes.data_as(C.c_void_p))
    us = st[:, [0, 1, 2, 4]] / 100.0
    it = st[:, 6]
    plain = us[:, 0] - us[:, 1] - us[:, 2] - us[:, 3]
    ok = it > 0
    tot[:4] += us[ok].sum(0); tot[4] += plain[ok].sum(); tot[5] += it[ok].sum(); tot[6] += st[ok, 3].sum(); tot[7] += st[ok, 5].sum()
    fac[:3] += st[ok, 8:11].sum(0) / 100.0; fac[3] += (st[ok, 3] + 1).sum()
    chk[:2] += st[ok, 12:14].sum(0) / 100.0; chk[2] += st[ok, 5].sum()
    w = st[ok, 14]
    s1, s2, s3 = (w & 0xFFFFF) / 100.0, ((w >> 20) & 0xFFFFF) / 100.0, ((w >> 40) & 0xFFFFF) / 100.0
    stp += np.array([s1.sum(), (s2 - s1).sum(), (s3 - s2).sum(), (st[ok, 1] / 100.0 - s3).sum(), ok.sum()])
    if k % every:
        continue
    order = np.argsort(-us[:, 0])[:8]
    print(f"tick {k}: slowest solves (agent: total ms | set-up | refactor (n) | checks (n) | iterations -> us/iter)")
    for a in order:
        print(f"   {a:3d}: {us[a,0]/1000:5.2f} | {us[a,1]/1000:4.2f} | {us[a,2]/1000:4.2f} ({st[a,3]}) | {us[a,3]/1000:4.2f} ({st[a,5]}) | "
              f"{it[a]:4d} -> {plain[a]/max(it[a],1):.3f}  fast={st[a,7]}")
print(f"all solves of {ticks} ticks: set-up {tot[1]/tot[0]:.1%}, refactorisations {tot[2]/tot[0]:.1%} ({tot[2]/max(tot[6],1):.1f} us each), "
      f"checks {tot[3]/tot[0]:.1%} ({tot[3]/max(tot[7],1):.2f} us each), iterations {tot[4]/tot[0]:.1%} ({tot[4]/max(tot[5],1):.3f} us each)")
print(f"shader clock during the solves of the last tick: {np.median(st[ok, 11] / (st[ok, 0] * 10.0)):.2f} GHz (median over agents)")
print(f"factor() phases, us per factorisation: block assembly {fac[0]/fac[3]:.1f}, forward sweeps {fac[1]/fac[3]:.1f}, backward rows + registers {fac[2]/fac[3]:.1f}")
print(f"checks, us each: spill of the row state {chk[0]/chk[2]:.2f}, residual pass {chk[1]/chk[2]:.2f}, tests / certificate {tot[3]/tot[7] - (chk[0]+chk[1])/chk[2]:.2f}")
print(f"set-up, us per solve: assembly + CSC {stp[0]/stp[4]:.1f}, Ruiz scaling {stp[1]/stp[4]:.1f}, rho + K1 {stp[2]/stp[4]:.1f}, first factorisation + load {stp[3]/stp[4]:.1f}")
