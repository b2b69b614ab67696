#!/bin/bash
# A/B aid: compact bench lines under the environments given as arguments ("K=V K=V" per argument)
export GPU_MAX_HW_QUEUES=32
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 150 python bench.py --no-cpu-baseline --sustained ${SUST:-0} --steps ${STEPS:-20} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; s=d.get('sustained') or {}
        print('value', round(d['value']), 'ms', round(d['ms_per_step'],2), 'clear_ms', round(r.get('avg_launch_ms'),2), 'frac', round(r['frac'],3), 'standalone', round(r['standalone']['frac'],3), 'ok', d['config'].get('replans_ok_fraction'), 'sustained', round(s.get('value',0)), s.get('tick_ms_mean'))
"
done
