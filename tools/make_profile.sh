#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box into gpurun_out/profile/ (run from the repo root); every step
# under its own `timeout`.  tools/make_profile_md.py assembles profiles/r06_*.{md,json} from it.
#   bash tools/make_profile.sh core   default bench line; kernel-trace + stats of `python bench.py`; separate PMC passes
#                                     (FETCH_SIZE, WRITE_SIZE; no trace domains) of the map kernels by themselves (sparse
#                                     reset, stamp, overlay) under BOTH cell orders, and of the dense clear kernels; flights
#                                     (timed + timeline); per-agent chain; rocm-smi state
#   bash tools/make_profile.sh rest   variants (cell order rows, dense clear, grouped path), the residency tests
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
  cp $REPO/gpurun_out/bench_last_detail.json $OUT/bench_plain_detail.json
  rocm-smi --showclocks --showpower --showtemp > $OUT/smi_after.txt 2>&1
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustained 0 --dense-ticks 0 --no-variants > $OUT/bench_trace.json 2> $OUT/trace.err
  cp $REPO/gpurun_out/bench_last_detail.json $OUT/bench_trace_detail.json
  # counter collection serialises kernels: the dataflow replan's persistent kernels cannot overlap then, so the map
  # kernels run by themselves (tools/diag_reset_pmc.py: reset + stamp + overlay per update, the reset in the variant the
  # tick runs under the replan) and the in-tick dense clear kernels in tools/diag_clear_pmc.py
  export SOGM_TUNING=reset_lanes=2,reset_unroll=1
  for L in tiled rows; do
    SOGM_LAYOUT=$L timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_rfetch_$L -- python $REPO/tools/diag_reset_pmc.py > /dev/null 2> $OUT/rfetch_$L.err
    SOGM_LAYOUT=$L timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_rwrite_$L -- python $REPO/tools/diag_reset_pmc.py > /dev/null 2> $OUT/rwrite_$L.err
    SOGM_LAYOUT=$L timeout 200 python $REPO/tools/diag_reset_pmc.py > $OUT/reset_alone_$L.txt 2>/dev/null
  done
  unset SOGM_TUNING
  timeout 200 python $REPO/tools/diag_reset_pmc.py > $OUT/reset_alone_wide.txt 2>/dev/null
  SOGM_SPARSE_RESET=0 SOGM_TUNING=clear_early=1 timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_cfetch -- python $REPO/tools/diag_clear_pmc.py > /dev/null 2> $OUT/cfetch.err
  SOGM_SPARSE_RESET=0 SOGM_TUNING=clear_early=1 timeout 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_cwrite -- python $REPO/tools/diag_clear_pmc.py > $OUT/clear_alone.txt 2> $OUT/cwrite.err
  cd $REPO
  python tools/rocprof_summary.py "$(db /tmp/prof_trace)" "$(db /tmp/prof_rfetch_tiled)" "$(db /tmp/prof_rwrite_tiled)" "$(db /tmp/prof_cfetch)" "$(db /tmp/prof_cwrite)" > $OUT/summary.md 2> $OUT/summary.err
  python tools/rocprof_summary.py - "$(db /tmp/prof_rfetch_rows)" "$(db /tmp/prof_rwrite_rows)" > $OUT/summary_rows.md 2>> $OUT/summary.err
  timeout 300 python tools/bench_flight.py 20 "" "flight_pace_us=0" "flight_admit=128" > $OUT/flight.txt 2>&1
  timeout 200 python tools/diag_flight.py 20 > $OUT/flight_timeline.txt 2>&1
  timeout 200 python tools/diag_flow.py 12 > $OUT/flow.txt 2>&1
  tail -c 300 $OUT/bench_plain.json
else
  cd $REPO
  SOGM_LAYOUT=rows timeout 400 python bench.py --no-cpu-baseline --no-variants --sustained 100 --dense-ticks 0 > $OUT/bench_rows.json 2>/dev/null
  cp $REPO/gpurun_out/bench_last_detail.json $OUT/bench_rows_detail.json
  SOGM_LAYOUT=rows timeout 300 python tools/bench_flight.py 20 "" > $OUT/flight_rows.txt 2>&1
  SOGM_SPARSE_RESET=0 timeout 300 python bench.py --no-cpu-baseline --no-variants --sustained 100 --dense-ticks 0 > $OUT/bench_dense.json 2>/dev/null
  cp $REPO/gpurun_out/bench_last_detail.json $OUT/bench_dense_detail.json
  SOGM_FLOW=0 timeout 300 python bench.py --no-cpu-baseline --no-variants --sustained 100 --dense-ticks 0 > $OUT/bench_flow0.json 2>/dev/null
  cp $REPO/gpurun_out/bench_last_detail.json $OUT/bench_flow0_detail.json
  timeout 900 python -m pytest tests/test_residency_gpu.py -q -s 2>&1 | grep -E "residency:|flight:" > $OUT/residency.txt
  ls $OUT | wc -l
fi
