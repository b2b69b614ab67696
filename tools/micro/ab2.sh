#!/bin/bash
# A/B aid: like ab.sh, plus the slowest QP's microseconds per iteration and the chain end of the timed region's last tick
export GPU_MAX_HW_QUEUES=32
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 200 python bench.py --no-cpu-baseline --sustained ${SUST:-0} --steps ${STEPS:-20} --dense-ticks 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d.get('sustained') or {}; c=d['chain_ms']; q=c['slowest_qp']
        print('value', round(d['value']), 'ms', round(d['ms_per_step'],2), 'chain_end', round(c['chain_end'],2), 'corr_max', round(c['corridor_max'],2), 'slowest qp', round(q['ms'],2), q['iterations'], 'it', round(q['us_per_iteration'],3), 'us/it', 'sustained', round(s.get('value',0)), round(s.get('tick_ms_p99',0),2))
"
done
