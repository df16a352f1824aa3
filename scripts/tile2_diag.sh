#!/bin/bash
# round 6: where does the split tile kernel (pais_tile2.hpp) spend its time?  bash scripts/tile2_diag.sh on the GPU box
out=gpurun_out/tile2_diag; mkdir -p $out
scripts/ubench/inst_rate > $out/inst_rate.txt 2>&1
B="--scene dome --max-rounds 10 --parents-per-round 4096 --steps 1 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; env "$@" python bench.py $B > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{})
    print("%-14s value %9.1f ms/step %9.1f pso %9.1f frac %.4f busy_frac %.4f sha %s" % (sys.argv[2], d['value'], d['ms_per_step'], k.get('pso_pass',0), d['roofline']['frac'], d['roofline'].get('frac_over_busy_time') or 0, str(d['config'].get('cloud_sha1'))[:10]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
{
run split PAIS_X=1
run old PAIS_TILE_SPLIT=0
run bias0 PAIS_TILE_BIAS=0
run bias7 PAIS_TILE_BIAS=7
run strip12 PAIS_TILE_STRIP_SPLIT=12
run strip41 PAIS_TILE_STRIP_SPLIT=41
run nosync PAIS_LIB_PATH=pais_mvs_amd/csrc/variants/libpais_nosync.so
run onestream PAIS_PSO_STREAMS=1
run old_onestream PAIS_TILE_SPLIT=0 PAIS_PSO_STREAMS=1
run split2 PAIS_X=1
} > $out/summary.txt 2>&1
cat $out/summary.txt
