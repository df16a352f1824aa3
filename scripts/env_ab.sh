#!/bin/bash
# A/B of environment settings of the built library on one box: env_ab.sh <outdir> "<bench args>" NAME=ENV[,ENV...] ...   (alternating, 3 passes)
out=gpurun_out/$1; mkdir -p $out; bargs=$2; shift 2
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{})
    print("%-22s value %10.1f ms/step %9.2f pso %8.2f frac %.4f sha %s" % (sys.argv[2], d['value'], d['ms_per_step'], k.get('pso_pass',0), d['roofline']['frac'], str(d['config'].get('cloud_sha1'))[:10]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
{
run warm PAIS_X=1 -- --steps 4 --warmup 2
for i in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    IFS=',' read -ra E <<< "$envs"
    run ${name}_$i "${E[@]}" -- $bargs
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
