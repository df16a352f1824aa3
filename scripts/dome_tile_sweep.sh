# strip-length sweep of the tile kernel on the full-size dome: bash scripts/dome_tile_sweep.sh on the GPU box -> profiles/r03_dome_tile_sweep.txt
mkdir -p gpurun_out/r03e
run() { name=$1; shift
  env "$@" PAIS_TILE_DEBUG=1 timeout 600 python bench.py --scene dome --steps 1 --warmup 0 --max-rounds 3 --parents-per-round 1024 --no-cpu-baseline 2>gpurun_out/r03e/err_$name.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],1), d['config']['cloud_sha1'][:10], {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()})"
  grep "pais tile" gpurun_out/r03e/err_$name.txt | tail -2
}
run default X=1


run s1_24 PAIS_TILE_STRIP1=24
