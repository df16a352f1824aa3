#!/bin/bash
# dome seeds + R rounds: the library against builds whose waves start a strip's walk staggered (-DTILE_STAGGER=k: wave w sleeps 64 k w cycles)
out=gpurun_out/${1:-dome_stag}; mkdir -p $out; R=${2:-10}; B=${3:-4096}
run() { name=$1; lib=$2
  if [ -n "$lib" ]; then export PAIS_LIB_PATH=$lib; else unset PAIS_LIB_PATH; fi
  python bench.py --no-cpu-baseline --scene dome --max-rounds $R --parents-per-round $B --steps 1 --warmup 1 > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{})
    print("%-14s value %9.1f ms/step %9.1f pso %9.1f frac %.4f sha %s" % (sys.argv[2], d['value'], d['ms_per_step'], k.get('pso_pass',0), d['roofline']['frac'], str(d['config'].get('cloud_sha1'))[:10]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
{
run base ""
for k in 2 6 15; do run stag$k pais_mvs_amd/csrc/variants/libpais_stag$k.so; done
run base_b ""
} > $out/summary.txt 2>&1
cat $out/summary.txt
