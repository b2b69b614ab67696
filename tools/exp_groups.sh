run() { timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), round(d['value'],1))"; }
SOGM_GROUPS=8 run "g8"
SOGM_GROUPS=10 run "g10"
SOGM_GROUPS=12 run "g12"
SOGM_GROUPS=14 run "g14"
SOGM_GROUPS=8 run "g8 again"
SOGM_GROUPS=14 run "g14 again"
