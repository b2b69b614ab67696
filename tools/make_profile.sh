#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box into gpurun_out/profile/ (run from the repo root); every step
# under its own `timeout`.  tools/make_profile_md.py assembles profiles/r04_*.{md,json} from it.
#   bash tools/make_profile.sh core   default bench line, kernel-trace + stats of `python bench.py`, separate PMC passes
#                                     (FETCH_SIZE, WRITE_SIZE; no trace domains) of the map kernels by themselves (sparse
#                                     reset, stamp, overlay) and of the dense clear kernels, tick timeline, per-agent chain,
#                                     QP clock split, rocm-smi state
#   bash tools/make_profile.sh rest   variants (dense clear, grouped path, two grids, single grid, cfg4), residency test
set -u
PART=${1:-core}
OUT=$PWD/gpurun_out/profile
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*.db" | head -1; }
if [ "$PART" = core ]; then
  rocm-smi --showclocks --showpower --showtemp > $OUT/smi_before.txt 2>&1
  timeout 900 python $REPO/bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
  rocm-smi --showclocks --showpower --showtemp > $OUT/smi_after.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustained 0 --dense-ticks 0 > $OUT/bench_trace.json 2> $OUT/trace.err
  # counter collection serialises kernels: the dataflow replan's persistent kernels cannot overlap then, so the map
  # kernels run by themselves (tools/diag_reset_pmc.py: reset + stamp + overlay per update; the variant of the reset the
  # tick runs under the replan) and the in-tick dense clear kernels in tools/diag_clear_pmc.py
  export SOGM_TUNING=reset_lanes=2,reset_unroll=1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_rfetch -- python $REPO/tools/diag_reset_pmc.py > /dev/null 2> $OUT/rfetch.err
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_rwrite -- python $REPO/tools/diag_reset_pmc.py > /dev/null 2> $OUT/rwrite.err
  timeout 200 python $REPO/tools/diag_reset_pmc.py > $OUT/reset_alone_plain.txt 2>/dev/null
  unset SOGM_TUNING
  timeout 200 python $REPO/tools/diag_reset_pmc.py > $OUT/reset_alone_wide.txt 2>/dev/null
  SOGM_SPARSE_RESET=0 SOGM_TUNING=clear_early=1 timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_cfetch -- python $REPO/tools/diag_clear_pmc.py > /dev/null 2> $OUT/cfetch.err
  SOGM_SPARSE_RESET=0 SOGM_TUNING=clear_early=1 timeout 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_cwrite -- python $REPO/tools/diag_clear_pmc.py > $OUT/clear_alone.txt 2> $OUT/cwrite.err
  cd $REPO
  python tools/rocprof_summary.py "$(db /tmp/prof_trace)" "$(db /tmp/prof_rfetch)" "$(db /tmp/prof_rwrite)" "$(db /tmp/prof_cfetch)" "$(db /tmp/prof_cwrite)" > $OUT/summary.md 2> $OUT/summary.err
  python tools/tick_timeline.py /tmp/prof_trace > $OUT/timeline.txt 2>&1
  timeout 200 python tools/diag_flow.py 12 > $OUT/flow.txt 2>&1
  timeout 200 python tools/diag_qp_time.py 12 4 > $OUT/qp_time.txt 2>&1
  tail -c 300 $OUT/bench_plain.json
else
  cd $REPO
  SOGM_SPARSE_RESET=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained 100 --dense-ticks 0 > $OUT/bench_dense.json 2>/dev/null
  SOGM_FLOW=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained 100 --dense-ticks 0 > $OUT/bench_flow0.json 2>/dev/null
  SOGM_GRIDS=2 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained 0 --dense-ticks 0 > $OUT/bench_grids2.json 2>/dev/null
  SOGM_DOUBLE_BUFFER=0 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained 0 --dense-ticks 0 > $OUT/bench_mode1.json 2>/dev/null
  timeout 300 python bench.py --grid cfg4 --steps 10 --warmup 2 --no-cpu-baseline --sustained 0 --dense-ticks 0 > $OUT/bench_cfg4.json 2>/dev/null
  timeout 600 python -m pytest tests/test_residency_gpu.py -q -s 2>&1 | grep "residency:" > $OUT/residency.txt
  timeout 200 python tools/diag_capacity.py 323 > $OUT/capacity.txt 2>&1
  ls $OUT | wc -l
fi
