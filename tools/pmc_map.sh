#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no trace domains) of the map kernels by themselves
# (tools/diag_reset_pmc.py: single grid, every update = sparse reset + stamp + overlay; counter collection serialises
# kernels, under which the dataflow replan cannot run).  Output: gpurun_out/pmc_map/{fetch,write}.txt + plain.txt
set -u
OUT=$PWD/gpurun_out/pmc_map
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
export SOGM_TUNING=reset_lanes=2,reset_unroll=1   # the variant the tick runs under the replan
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pm_fetch -- python $REPO/tools/diag_reset_pmc.py > /dev/null 2> $OUT/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pm_write -- python $REPO/tools/diag_reset_pmc.py > /dev/null 2> $OUT/write.err
timeout 200 python $REPO/tools/diag_reset_pmc.py > $OUT/plain.txt 2>/dev/null
cd $REPO
python tools/pmc_kernels.py /tmp/pm_fetch > $OUT/fetch.txt 2>&1
python tools/pmc_kernels.py /tmp/pm_write > $OUT/write.txt 2>&1
cat $OUT/plain.txt; grep -h "reset_sectors\|stamp_\|splat\|clear" $OUT/fetch.txt $OUT/write.txt
