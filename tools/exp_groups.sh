run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), round(d['value'],1))"; }
SOGM_GROUPS=8 run "g8 q16"
SOGM_GROUPS=14 run "g14 q16"
SOGM_GROUPS=16 GPU_MAX_HW_QUEUES=24 run "g16 q24"
SOGM_GROUPS=32 GPU_MAX_HW_QUEUES=40 run "g32 q40"
SOGM_GROUPS=32 GPU_MAX_HW_QUEUES=16 run "g32 q16"
SOGM_GROUPS=64 GPU_MAX_HW_QUEUES=72 run "g64 q72"
