#!/bin/bash
# dome seeds + R rounds: the prefetching one-pixel instantiations of the tile kernel (PAIS_TILE_PF=1, round 5) against round 4's (PAIS_TILE_PF=0)
out=gpurun_out/${1:-dome_pf}; mkdir -p $out; R=${2:-10}; B=${3:-4096}
run() { name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --scene dome --max-rounds $R --parents-per-round $B --steps 1 --warmup 1 > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{})
    print("%-14s value %9.1f ms/step %9.1f pso %9.1f frac %.4f sha %s" % (sys.argv[2], d['value'], d['ms_per_step'], k.get('pso_pass',0), d['roofline']['frac'], str(d['config'].get('cloud_sha1'))[:10]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
{
run pf0 PAIS_TILE_PF=0
run pf1_24 PAIS_TILE_PF=1
run pf1_16 PAIS_TILE_PF=1 PAIS_TILE_STRIP1=16
run pf1_32 PAIS_TILE_PF=1 PAIS_TILE_STRIP1=32
run pf1_12 PAIS_TILE_PF=1 PAIS_TILE_STRIP1=12
run pf0_b PAIS_TILE_PF=0
} > $out/summary.txt 2>&1
cat $out/summary.txt
