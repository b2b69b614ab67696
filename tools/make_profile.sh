#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box into gpurun_out/profile/ (run from the repo root):
#   bench lines (default run incl. the sustained block and the CPU baseline; grouped path; single grid; cfg4),
#   kernel-trace + stats of `python bench.py`, separate PMC passes (FETCH_SIZE, WRITE_SIZE; no trace domains),
#   the per-agent chain of the dataflow replan, and kernel-trace + PMC passes of the perception side benches.
# tools/make_profile_md.py assembles profiles/r03_*.md from it.
set -u
OUT=$PWD/gpurun_out/profile
rm -rf "$OUT"; mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
# clock / power state of the box next to every figure: before, and again after the default run
rocm-smi --showclocks --showpower --showtemp > $OUT/smi_before.txt 2>&1
python $REPO/bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocm-smi --showclocks --showpower --showtemp > $OUT/smi_after.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustained 0 > $OUT/bench_trace.json 2> $OUT/trace.err
# (counter collection serialises kernels: the dataflow replan's persistent kernels cannot overlap then and a tick would
#  run into its 3 s wait limit — the grouped path is profiled instead; the rated kernel, k_clear_slabs, is the same)
SOGM_FLOW=0 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustained 0 > /dev/null 2> $OUT/fetch.err
SOGM_FLOW=0 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustained 0 > /dev/null 2> $OUT/write.err
# the in-tick clear kernels (k_clear_chunks) and the stamp by themselves (see tools/diag_clear_pmc.py)
SOGM_CLEAR_EARLY=1 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_cfetch -- python $REPO/tools/diag_clear_pmc.py > /dev/null 2> $OUT/cfetch.err
SOGM_CLEAR_EARLY=1 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_cwrite -- python $REPO/tools/diag_clear_pmc.py > $OUT/clear_alone.txt 2> $OUT/cwrite.err
for b in dsp gridmap; do
  rocprofv3 --kernel-trace --stats -d /tmp/prof_${b}_trace -- python $REPO/tools/bench_$b.py > $OUT/bench_${b}_trace.json 2> $OUT/${b}_trace.err
  rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_${b}_fetch -- python $REPO/tools/bench_$b.py > /dev/null 2> $OUT/${b}_fetch.err
  rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_${b}_write -- python $REPO/tools/bench_$b.py > /dev/null 2> $OUT/${b}_write.err
done
cd $REPO
db() { find "$1" -name "*.db" | head -1; }
python tools/rocprof_summary.py "$(db /tmp/prof_trace)" "$(db /tmp/prof_fetch)" "$(db /tmp/prof_write)" "$(db /tmp/prof_cfetch)" "$(db /tmp/prof_cwrite)" > $OUT/summary.md 2> $OUT/summary.err
python tools/tick_timeline.py /tmp/prof_trace > $OUT/timeline.txt 2>&1
for b in dsp gridmap; do
  python tools/rocprof_summary.py "$(db /tmp/prof_${b}_trace)" "$(db /tmp/prof_${b}_fetch)" "$(db /tmp/prof_${b}_write)" > $OUT/summary_$b.md 2> $OUT/summary_$b.err
done
python tools/diag_flow.py 12 > $OUT/flow.txt 2>&1
python tools/diag_capacity.py 323 > $OUT/capacity.txt 2>&1
python tools/diag_cfg4_residuals.py cfg4 > $OUT/cfg4_residuals.txt 2>&1
python tools/diag_qp_infeasible.py dump 23 > $OUT/qp_dump.log 2>&1
python tools/diag_qp_parity.py > $OUT/qp_parity.txt 2>&1
SOGM_FLOW=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained 100 > $OUT/bench_flow0.json 2>/dev/null
SOGM_GRIDS=2 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained 0 > $OUT/bench_grids2.json 2>/dev/null
SOGM_DOUBLE_BUFFER=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained 0 > $OUT/bench_mode1.json 2>/dev/null
python bench.py --grid cfg4 --steps 10 --warmup 2 --no-cpu-baseline --sustained 0 > $OUT/bench_cfg4.json 2>/dev/null
python tools/bench_dsp.py > $OUT/bench_dsp.json 2>/dev/null
python tools/bench_gridmap.py > $OUT/bench_gridmap.json 2>/dev/null
tail -c 400 $OUT/bench_plain.json
