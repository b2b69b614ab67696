#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box into gpurun_out/profile/ (run from the repo root), in two
# parts so that one gpurun call stays short (every step under its own `timeout`):
#   bash tools/make_profile.sh core    bench lines (default run incl. the sustained block and the CPU baseline),
#                                      kernel-trace + stats of `python bench.py`, separate PMC passes (FETCH_SIZE,
#                                      WRITE_SIZE; no trace domains) of the dense clear kernels and of the sparse
#                                      reset, the dense-clear variant of the bench, tick timeline, per-agent chain,
#                                      rocm-smi state
#   bash tools/make_profile.sh rest    variants (grouped path, two grids, single grid, cfg4), perception side benches
#                                      with their trace / PMC passes, the capacity / QP / residual diagnostics
# tools/make_profile_md.py assembles profiles/r03_*.md from it.
set -u
PART=${1:-core}
OUT=$PWD/gpurun_out/profile
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*.db" | head -1; }
if [ "$PART" = core ]; then
  # clock / power state of the box next to every figure: before, and again after the default run
  rocm-smi --showclocks --showpower --showtemp > $OUT/smi_before.txt 2>&1
  timeout 600 python $REPO/bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
  rocm-smi --showclocks --showpower --showtemp > $OUT/smi_after.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustained 0 > $OUT/bench_trace.json 2> $OUT/trace.err
  # (counter collection serialises kernels: the dataflow replan's persistent kernels cannot overlap then and a tick
  #  would run into its wait limits — the grouped path is profiled for k_clear_slabs, and the in-tick clear kernels
  #  (k_clear_chunks) and the stamp run by themselves in tools/diag_clear_pmc.py)
  SOGM_FLOW=0 timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustained 0 > /dev/null 2> $OUT/fetch.err
  SOGM_FLOW=0 timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustained 0 > /dev/null 2> $OUT/write.err
  SOGM_SPARSE_RESET=0 SOGM_TUNING=clear_early=1 timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_cfetch -- python $REPO/tools/diag_clear_pmc.py > /dev/null 2> $OUT/cfetch.err
  SOGM_SPARSE_RESET=0 SOGM_TUNING=clear_early=1 timeout 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_cwrite -- python $REPO/tools/diag_clear_pmc.py > $OUT/clear_alone.txt 2> $OUT/cwrite.err
  # the sparse reset (k_reset_sectors) and the logging stamp / overlay by themselves; counters for the variant the tick
  # runs under the replan (2 lanes, 1 entry per trip), launch times for it and for the in-stream variant (4 lanes x 8)
  export SOGM_TUNING=reset_lanes=2,reset_unroll=1
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_rfetch -- python $REPO/tools/diag_reset_pmc.py > /dev/null 2> $OUT/rfetch.err
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_rwrite -- python $REPO/tools/diag_reset_pmc.py > $OUT/reset_alone.txt 2> $OUT/rwrite.err
  timeout 200 python $REPO/tools/diag_reset_pmc.py > $OUT/reset_alone_plain.txt 2>/dev/null
  unset SOGM_TUNING
  timeout 200 python $REPO/tools/diag_reset_pmc.py > $OUT/reset_alone_wide.txt 2>/dev/null
  # the dense clear in the tick (the path of rounds 1-2), for comparison on this box
  SOGM_SPARSE_RESET=0 timeout 300 python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained 100 > $OUT/bench_dense.json 2>/dev/null
  cd $REPO
  python tools/rocprof_summary.py "$(db /tmp/prof_trace)" "$(db /tmp/prof_fetch)" "$(db /tmp/prof_write)" "$(db /tmp/prof_cfetch)" "$(db /tmp/prof_cwrite)" "$(db /tmp/prof_rfetch)" "$(db /tmp/prof_rwrite)" > $OUT/summary.md 2> $OUT/summary.err
  python tools/tick_timeline.py /tmp/prof_trace > $OUT/timeline.txt 2>&1
  timeout 200 python tools/diag_flow.py 12 > $OUT/flow.txt 2>&1
  tail -c 400 $OUT/bench_plain.json
else
  for b in dsp gridmap; do
    timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_${b}_trace -- python $REPO/tools/bench_$b.py > $OUT/bench_${b}_trace.json 2> $OUT/${b}_trace.err
    timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_${b}_fetch -- python $REPO/tools/bench_$b.py > /dev/null 2> $OUT/${b}_fetch.err
    timeout 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_${b}_write -- python $REPO/tools/bench_$b.py > /dev/null 2> $OUT/${b}_write.err
  done
  cd $REPO
  for b in dsp gridmap; do
    python tools/rocprof_summary.py "$(db /tmp/prof_${b}_trace)" "$(db /tmp/prof_${b}_fetch)" "$(db /tmp/prof_${b}_write)" > $OUT/summary_$b.md 2> $OUT/summary_$b.err
  done
  timeout 200 python tools/diag_capacity.py 323 > $OUT/capacity.txt 2>&1
  timeout 200 python tools/diag_cfg4_residuals.py cfg4 > $OUT/cfg4_residuals.txt 2>&1
  timeout 200 python tools/diag_qp_infeasible.py dump 23 > $OUT/qp_dump.log 2>&1
  timeout 400 python tools/diag_qp_parity.py > $OUT/qp_parity.txt 2>&1
  SOGM_FLOW=0 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained 100 > $OUT/bench_flow0.json 2>/dev/null
  SOGM_GRIDS=2 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained 0 > $OUT/bench_grids2.json 2>/dev/null
  SOGM_DOUBLE_BUFFER=0 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained 0 > $OUT/bench_mode1.json 2>/dev/null
  timeout 300 python bench.py --grid cfg4 --steps 10 --warmup 2 --no-cpu-baseline --sustained 0 > $OUT/bench_cfg4.json 2>/dev/null
  timeout 200 python tools/bench_dsp.py > $OUT/bench_dsp.json 2>/dev/null
  timeout 200 python tools/bench_gridmap.py > $OUT/bench_gridmap.json 2>/dev/null
  ls $OUT | wc -l
fi
