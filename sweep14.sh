mkdir -p gpurun_out/r03j
for i in 1 2; do
PAIS_TILE_DEBUG=1 python bench.py --scene dome --steps 1 --warmup 1 --max-rounds 3 --parents-per-round 1024 --no-cpu-baseline 2>gpurun_out/r03j/dome_rank.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dome depth-rank', round(d['value'],1), round(d['ms_per_step'],1), d['config']['cloud_sha1'][:10], {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()})"
grep "pais tile" gpurun_out/r03j/dome_rank.err | tail -2
done
PAIS_TILE_STRIP1=41 PAIS_TILE_STRIP2=42 PAIS_TILE_DEBUG=1 python bench.py --scene dome --steps 1 --warmup 1 --max-rounds 3 --parents-per-round 1024 --no-cpu-baseline 2>gpurun_out/r03j/dome_rank.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dome depth-rank one strip', round(d['value'],1), round(d['ms_per_step'],1), d['config']['cloud_sha1'][:10], {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()})"
grep "pais tile" gpurun_out/r03j/dome_rank.err | tail -2
