#!/bin/bash
# dome bench under several environments, one box, PASSES alternating passes: dome_env_ab.sh <outdir> <rounds> <passes> NAME=ENV[,ENV...] ...
out=gpurun_out/$1; mkdir -p $out; R=$2; P=$3; shift 3
B="--scene dome --max-rounds $R --parents-per-round 4096 --steps 1 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; timeout 600 env "$@" python bench.py $B > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{})
    print("%-16s value %9.1f ms/step %9.1f pso %9.1f frac %.4f busy_frac %.4f sha %s" % (sys.argv[2], d['value'], d['ms_per_step'], k.get('pso_pass',0), d['roofline']['frac'], d['roofline'].get('frac_over_busy_time') or 0, str(d['config'].get('cloud_sha1'))[:10]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
{
for i in $(seq 1 $P); do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    IFS=',' read -ra E <<< "$envs"
    run ${name}_$i "${E[@]}"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
