show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],2), d['config']['cloud_sha1'][:10], d['config'].get('cloud_matches_oracle_golden'), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items() if k in ('pso_pass','cost_evaluation_sum_of_launches')})"; }
python scripts/microbench_eval.py 1200000 2>&1 | tail -2
python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | show pawn
python bench.py --scene ring --steps 1 --warmup 1 --no-cpu-baseline --max-rounds 60 2>/dev/null | show ring60
python bench.py --scene dome --steps 1 --warmup 1 --max-rounds 3 --parents-per-round 1024 --no-cpu-baseline 2>/dev/null | show dome
PAIS_TILE=0 python bench.py --scene dome --steps 1 --warmup 1 --max-rounds 3 --parents-per-round 1024 --no-cpu-baseline 2>/dev/null | show dome_notile
