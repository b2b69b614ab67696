"""Diagnostic: run the DSP parity scenario and print the first differences in detail."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pop = importlib.import_module("pred-occ-planner_amd")
sogm = importlib.import_module("pred-occ-planner_amd.sogm")
dsp = importlib.import_module("pred-occ-planner_amd.dsp")
orc = importlib.import_module("oracle.binding")
grid = sys.argv[1] if len(sys.argv) > 1 else "cfg0"
seed = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0x81
n_up = int(sys.argv[3]) if len(sys.argv) > 3 else 12
spec = pop.config.make_spec(grid)
P = dsp.make_dsp_params(spec.T)
tabs = dsp.make_tables(11, n_gauss=1 << 18, n_rand=1 << 12)
half = (spec.L * 0.15 / 2, spec.W * 0.15 / 2, spec.H * 0.15 / 2)
seq = pop.scene.make_dsp_sequence(seed, n_up, half=half)
m = sogm.SogmMap(spec, 1)
g = dsp.DspMap(m, P, tabs)
o = orc.DspOracle(spec, P, tabs)
for k, s in enumerate(seq):
    n = len(s["points"])
    rng = np.asarray([[0, n]], np.int32)
    g.update(sogm._dev(s["points"]), sogm._dev(s["labels"]), sogm._dev(rng), sogm._dev(s["pos"][None]),
             sogm._dev(s["quat"][None]), sogm._dev(np.asarray([s["stamp"]])))
    o.update(s["points"], s["labels"], s["pos"], s["quat"], s["stamp"])
    ws, wo, wc = o.state()
    gs, go, gc = g.download_state(0)
    bad = np.nonzero((gs[:, :, 0] != ws[:, :, 0]).any(axis=1))[0]
    print("update", k, "gc", gc, "wc", wc[:10], "bad voxels", len(bad))
    for v in bad[:4]:
        print(" voxel", v)
        print("  want flags", ws[v, :, 0]); print("  got  flags", gs[v, :, 0])
        print("  want w", ws[v, :, 7]); print("  got  w", gs[v, :, 7])
        print("  want px", ws[v, :, 4]); print("  got  px", gs[v, :, 4])
    if len(bad):
        break
