#!/bin/bash
# A/B of the built library against pais_mvs_amd/csrc/variants/libpais_prev.so on one box, dome scene (seeds + R rounds), alternating
out=gpurun_out/${1:-ab_dome}; mkdir -p $out; R=${2:-10}; B=${3:-4096}
V=${PREV_LIB:-pais_mvs_amd/csrc/variants/libpais_prev.so}
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline --scene dome --max-rounds $R --parents-per-round $B "$@" > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{})
    print("%-14s value %9.1f ms/step %9.1f pso %9.1f frac %.4f sha %s gold %s" % (sys.argv[2], d['value'], d['ms_per_step'], k.get('pso_pass',0), d['roofline']['frac'], str(d['config'].get('cloud_sha1'))[:10], d['config'].get('cloud_matches_oracle_golden')))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
{
for i in 1 2; do
run new$i PAIS_X=1 -- --steps 1 --warmup 1
run prev$i PAIS_LIB_PATH=$V -- --steps 1 --warmup 1
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
