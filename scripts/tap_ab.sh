#!/bin/bash
# tap representation A/B on one box: float2 (built library) vs double2 (variants/libpais_tapd.so) vs bytes (PAIS_TAP_FLOAT_MAX_MB=0)
out=gpurun_out/${1:-tap}; mkdir -p $out
V=pais_mvs_amd/csrc/variants/libpais_tapd.so
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); r=d['roofline']
    print("%-22s value %10.1f ms/step %9.2f pso %8.2f frac %.4f busy-frac %s sha %s" % (sys.argv[2], d['value'], d['ms_per_step'], k.get('pso_pass',0), r['frac'], ("%.3f" % r['frac_over_busy_time']) if r.get('frac_over_busy_time') else "-", str(d['config'].get('cloud_sha1'))[:10]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
{
for i in 1 2 3; do
python scripts/microbench_eval.py 2>&1 | tail -n 1 | sed 's/^/float2  /'
PAIS_LIB_PATH=$V python scripts/microbench_eval.py 2>&1 | tail -n 1 | sed 's/^/double2 /'
done
for i in 1 2 3; do
run pawn_f2_$i PAIS_X=1 -- --steps 20 --warmup 3
run pawn_d2_$i PAIS_LIB_PATH=$V -- --steps 20 --warmup 3
done
for i in 1 2; do
run ring_f2_$i PAIS_X=1 -- --scene ring --steps 1 --warmup 0 --max-rounds 300
run ring_d2_$i PAIS_LIB_PATH=$V -- --scene ring --steps 1 --warmup 0 --max-rounds 300
run ring_bytes_$i PAIS_TAP_FLOAT_MAX_MB=0 -- --scene ring --steps 1 --warmup 0 --max-rounds 300
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
