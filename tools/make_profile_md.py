"""Assemble profiles/r01_end_rocprof.md + profiles/r01_pmc_traffic.json from gpurun_out/profile/ (tools/make_profile.sh)."""
import json, re, sys
P = 'gpurun_out/profile/'
last = lambda f: open(P + f).read().strip().splitlines()[-1]
summ = open(P + 'summary.md').read()
tl = open(P + 'timeline.txt').read()
plain, trace, mode1, cfg4, dsp, gm = (last(f) for f in ('bench_plain.json', 'bench_trace.json', 'bench_mode1.json', 'bench_cfg4.json', 'bench_dsp.json', 'bench_gridmap.json'))
d, m1, c4 = json.loads(plain), json.loads(mode1), json.loads(cfg4)


def pm(counter):
    m = re.search(r"k_clear_slabs<true> \| %s \| (\d+) \| ([\d.]+) \|" % counter, summ)
    return int(m.group(1)), float(m.group(2))


nf, fk = pm('FETCH_SIZE')
nw, wk = pm('WRITE_SIZE')
traffic = int((fk + wk) * 1024)
json.dump({"kernel": "k_clear_slabs", "source": "profiles/r01_end_rocprof.md (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, 128 agents 200x200x200x20; mean over the narrow in-tick launches and the full-width stage-pass launch)",
           "fetch_kb": fk, "write_kb": wk, "bytes_per_launch": traffic, "algorithmic_bytes_per_launch": 81920000000}, open('profiles/r01_pmc_traffic.json', 'w'))


def stats_block(text):
    out = []
    for line in text.splitlines():
        m = re.match(r"- (.*?): ([\d, ]+)$", line)
        if m:
            v = sorted(float(x) for x in m.group(2).split(','))
            n = len(v)
            out.append(f"- {m.group(1)}: n={n} min {v[0]:.0f} median {v[n//2]:.0f} mean {sum(v)/n:.0f} max {v[-1]:.0f}")
        else:
            out.append(line)
    return "\n".join(out)


summ = stats_block(summ)
r = d['roofline']
md = f"""# Round 1 (end of round, numbers of record) — rocprofv3 profile of `python bench.py` (MI355X, 128 agents, 200^3 x 20)

Collected by `tools/make_profile.sh` on the GPU box (`cd /tmp && export TMPDIR=/tmp`), assembled by `tools/make_profile_md.py`:
- `rocprofv3 --kernel-trace --stats -d … -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline`
- separate PMC passes (no trace domains): `rocprofv3 --pmc FETCH_SIZE -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline`
  and the same with `--pmc WRITE_SIZE`.
Summarised by `tools/rocprof_summary.py` / `tools/tick_timeline.py` from the rocpd databases (the .db files stay
in gpurun_out/).

State: 512-lane register-resident ADMM iteration in `k_qp` (1.27 us per iteration, was 3.9 at the start of the
session), double-buffered SOGM with a narrow (64-workgroup) streaming clear under the whole replan, obstacle-point
scan split from the FIRI kernel, stage-wise costMVIE reduction, reworked A* expansion (19 us, was 27), every planner
kernel free of scratch memory, 8 agent-group streams.

Unprofiled bench of the same box (`python bench.py --steps 30 --warmup 3`): **{d['value']:.0f} replans/s**,
{d['ms_per_step']:.2f} ms per tick, replans_ok {d['config']['replans_ok_fraction']:.3f}; `k_clear_slabs` inside the tick
{r['avg_launch_ms']:.2f} ms per launch = {r['achieved']:.0f} GB/s = {r['frac']:.3f} of the 8 TB/s HBM peak (narrow launch sharing
the machine with the planner kernels), {r['standalone']['frac']:.3f} for the full-width launch of the stage pass (0.72-0.85 across
boxes and runs); cpu_baseline {d['cpu_baseline']['value']:.1f} replans/s on 8 host threads.
Single-grid mode (`SOGM_DOUBLE_BUFFER=0`, full-width clear in place under the QP stage): {m1['value']:.0f} replans/s,
clear at {m1['roofline']['frac']:.3f}.  300^3 x 30 with fp16 cells (207 GB, single grid): {c4['value']:.0f} replans/s, clear at {c4['roofline']['frac']:.3f}.

Reading guide:
- `k_clear_slabs` is the roofline kernel (SOGM voxel update, 81.92 GB algorithmic bytes per launch =
  128 agents x 640 MB); PMC: FETCH_SIZE {fk:.0f} KB + WRITE_SIZE {wk:.0f} KB = {traffic/1e9:.2f} GB per launch (KB = 1024 B),
  i.e. {traffic/81.92e9:.4f} x the algorithmic bytes.
- Planner kernels are launched once per agent group (8 groups of 16 agents) per tick; their per-dispatch
  durations overlap in time (separate HIP streams), so the "total" column is not wall time — the tick timeline
  below shows what is on the critical path (stamp -> A* -> obstacle points -> FIRI -> the slowest QP -> deconfliction).
- the clear runs on a side stream from the start of the replan to ~75 % of the tick and is off the critical path.

{summ}

## timeline of one tick (ms from the tick's k_stamp_cloud)

```
{tl.strip()}
```

## bench.py JSON lines

- unprofiled, same box (`python bench.py --steps 30 --warmup 3`):

```
{plain}
```

- under `--kernel-trace --stats`:

```
{trace}
```

- single-grid mode (`SOGM_DOUBLE_BUFFER=0 python bench.py --steps 30 --warmup 3 --no-cpu-baseline`):

```
{mode1}
```

- BASELINE configs[4] (`python bench.py --grid cfg4 --steps 10 --warmup 2 --no-cpu-baseline`):

```
{cfg4}
```

## Perception side (tools/bench_dsp.py, tools/bench_gridmap.py; same box)

```
{dsp}
{gm}
```
"""
open('profiles/r01_end_rocprof.md', 'w').write(md)
print("wrote profiles/r01_end_rocprof.md", len(md), "bytes; traffic", traffic)
