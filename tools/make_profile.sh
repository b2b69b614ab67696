#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box into gpurun_out/profile/ (run from the repo root):
#   kernel-trace + stats of `python bench.py`, separate PMC passes (FETCH_SIZE, WRITE_SIZE), the tick timeline,
#   the unprofiled bench line and the perception side benches.
set -u
OUT=$PWD/gpurun_out/profile
rm -rf "$OUT"; mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py --steps 30 --warmup 3 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/write.err
cd $REPO
T=$(find /tmp/prof_trace -name "*.db" | head -1); F=$(find /tmp/prof_fetch -name "*.db" | head -1); W=$(find /tmp/prof_write -name "*.db" | head -1)
python tools/rocprof_summary.py "$T" "$F" "$W" > $OUT/summary.md 2> $OUT/summary.err
python tools/tick_timeline.py /tmp/prof_trace > $OUT/timeline.txt 2>&1
SOGM_DOUBLE_BUFFER=0 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > $OUT/bench_mode1.json 2>/dev/null
python bench.py --grid cfg4 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg4.json 2>/dev/null
python tools/bench_dsp.py > $OUT/bench_dsp.json 2>/dev/null
python tools/bench_gridmap.py > $OUT/bench_gridmap.json 2>/dev/null
tail -c 300 $OUT/bench_plain.json
