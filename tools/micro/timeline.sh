#!/bin/bash
# one-tick kernel timeline of the bench under the environment passed on the command line (KEY=VALUE ...)
REPO=$PWD
export GPU_MAX_HW_QUEUES=32
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
env "$@" timeout 200 rocprofv3 --kernel-trace -d /tmp/tl -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustained 0 2>/dev/null | tail -c 300
echo
python $REPO/tools/tick_timeline.py /tmp/tl $TL_ALL
