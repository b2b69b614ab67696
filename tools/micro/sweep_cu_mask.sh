export GPU_MAX_HW_QUEUES=16
run() { echo "== $*"; env "$@" timeout 150 python bench.py --no-cpu-baseline --sustained 0 --steps 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('value', round(d['value']), 'ms', round(d['ms_per_step'],2), 'clear_ms', r.get('avg_launch_ms'), 'frac', round(r['frac'],3))
"; }
run A=0
run SOGM_CLEAR_CUS=4 SOGM_CLEAR_WGS=128 SOGM_CLEAR_THROTTLE=0
run SOGM_CLEAR_CUS=8 SOGM_CLEAR_WGS=256 SOGM_CLEAR_THROTTLE=0
run SOGM_CLEAR_CUS=8 SOGM_CLEAR_WGS=512 SOGM_CLEAR_THROTTLE=0
run SOGM_CLEAR_CUS=12 SOGM_CLEAR_WGS=384 SOGM_CLEAR_THROTTLE=0
run SOGM_CLEAR_CUS=16 SOGM_CLEAR_WGS=512 SOGM_CLEAR_THROTTLE=0
run SOGM_CLEAR_CUS=8 SOGM_CLEAR_WGS=64 SOGM_CLEAR_THROTTLE=4
