#!/bin/bash
# dome (configs[4]) A/B of tile-kernel workgroup shapes: the library (8 waves, one workgroup per CU) against
# variants/libpais_tile4x2.so (-DTILE_WAVES=4 -DTILE_WGS_PER_CU=2: two 4-wave workgroups share a CU's LDS), strip lengths swept
out=gpurun_out/${1:-dome_wg}; mkdir -p $out; R=${2:-10}; B=${3:-4096}
run() { name=$1; lib=$2; s2=$3; s1=$4
  if [ -n "$lib" ]; then export PAIS_LIB_PATH=$lib; else unset PAIS_LIB_PATH; fi
  PAIS_TILE_STRIP2=$s2 PAIS_TILE_STRIP1=$s1 python bench.py --no-cpu-baseline --scene dome --max-rounds $R --parents-per-round $B --steps 1 --warmup 1 > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{})
    print("%-22s value %9.1f ms/step %9.1f pso %9.1f frac %.4f sha %s" % (sys.argv[2], d['value'], d['ms_per_step'], k.get('pso_pass',0), d['roofline']['frac'], str(d['config'].get('cloud_sha1'))[:10]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
{
run base_14_24 "" 14 24
run w4x2_14_24 pais_mvs_amd/csrc/variants/libpais_tile4x2.so 14 24
run w4x2_10_16 pais_mvs_amd/csrc/variants/libpais_tile4x2.so 10 16
run w4x2_8_12 pais_mvs_amd/csrc/variants/libpais_tile4x2.so 8 12
run w4x2_6_8 pais_mvs_amd/csrc/variants/libpais_tile4x2.so 6 8
run base_14_24b "" 14 24
} > $out/summary.txt 2>&1
cat $out/summary.txt
