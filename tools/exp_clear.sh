run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), round(d['value'],1), 'clear', round(d['stage_ms']['clear'],2), 'frac', round(d['roofline']['frac'],3), 'grids', d['config']['sogm_grids_per_agent'])"; }
run "default(auto)"
SOGM_DOUBLE_BUFFER=0 run "mode1"
python bench.py --grid cfg4 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4', round(d['ms_per_step'],2), round(d['value'],1), 'frac', round(d['roofline']['frac'],3), 'grids', d['config']['sogm_grids_per_agent'])"
