show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()})"; }
(cd _r02 && python bench.py --scene ring --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | show r02)
PAIS_ENUM_THREADS=1 PAIS_TILE=0 python bench.py --scene ring --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | show head_notile
PAIS_ENUM_THREADS=1 PAIS_TILE=0 PAIS_PART_FILL=1.0 python bench.py --scene ring --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | show head_notile_fill1
(cd _r02 && python bench.py --scene ring --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | show r02_again)
