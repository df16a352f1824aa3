#!/bin/bash
# dome (configs[4]) seeds + R rounds: per-particle tiles of the tile kernel (round 5) against union tiles only (PAIS_TILE_PRIVATE=0 =
# round 4's kernel), strip lengths swept
out=gpurun_out/${1:-dome_adapt}; mkdir -p $out; R=${2:-10}; B=${3:-4096}
run() { name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --scene dome --max-rounds $R --parents-per-round $B --steps 1 --warmup 1 > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{})
    print("%-22s value %9.1f ms/step %9.1f pso %9.1f frac %.4f sha %s" % (sys.argv[2], d['value'], d['ms_per_step'], k.get('pso_pass',0), d['roofline']['frac'], str(d['config'].get('cloud_sha1'))[:10]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
  grep "pais tile" $out/$name.err | tail -2
}
{
run union_14_24 PAIS_TILE_PRIVATE=0
run priv_14_24 PAIS_TILE_PRIVATE=1
run priv_20_32 PAIS_TILE_STRIP2=20 PAIS_TILE_STRIP1=32
run priv_28_41 PAIS_TILE_STRIP2=28 PAIS_TILE_STRIP1=41
run priv_10_16 PAIS_TILE_STRIP2=10 PAIS_TILE_STRIP1=16
run priv_dbg PAIS_TILE_DEBUG=1
run union_14_24b PAIS_TILE_PRIVATE=0
} > $out/summary.txt 2>&1
cat $out/summary.txt
