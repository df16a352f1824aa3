show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],1), d['config']['cloud_sha1'][:10], {k:round(v,1) for k,v in d['kernel_ms_per_step'].items() if k in ('pso_pass','cost_evaluation_sum_of_launches','host_enumerate')})"; }
PAIS_TILE=0 python bench.py --scene ring --steps 1 --warmup 1 --no-cpu-baseline --max-rounds 60 2>/dev/null | show ring_head_notile
python bench.py --scene ring --steps 1 --warmup 1 --no-cpu-baseline --max-rounds 60 2>/dev/null | show ring_head_tile
(cd _r02 && python bench.py --scene ring --steps 1 --warmup 1 --no-cpu-baseline --max-rounds 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ring_r02', round(d['value'],1), round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items() if k in ('pso_pass','cost_evaluation_sum_of_launches')})")
python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | show pawn_head
(cd _r02 && python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pawn_r02', round(d['value'],1), round(d['ms_per_step'],1))")
