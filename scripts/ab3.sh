#!/bin/bash
# the built library against variants/libpais_prev.so and further variants (names after the out dir), pawn bench, alternating
out=gpurun_out/${1:-ab3}; mkdir -p $out; shift
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-emulate "$@" > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{})
    print("%-22s value %10.1f ms/step %9.2f pso %8.2f frac %.4f sha %s" % (sys.argv[2], d['value'], d['ms_per_step'], k.get('pso_pass',0), d['roofline']['frac'], str(d['config'].get('cloud_sha1'))[:10]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
{
run warm PAIS_X=1 -- --steps 5 --warmup 2
for i in 1 2 3; do
run new$i PAIS_X=1 -- --steps 20 --warmup 3
run prev$i PAIS_LIB_PATH=pais_mvs_amd/csrc/variants/libpais_prev.so -- --steps 20 --warmup 3
for v in "$@"; do run ${v}_$i PAIS_LIB_PATH=pais_mvs_amd/csrc/variants/libpais_$v.so -- --steps 20 --warmup 3; done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
