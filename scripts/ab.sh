#!/bin/bash
# A/B of the built library against pais_mvs_amd/csrc/variants/libpais_prev.so on one box: pawn bench, alternating
out=gpurun_out/${1:-ab}; mkdir -p $out
V=${PREV_LIB:-pais_mvs_amd/csrc/variants/libpais_prev.so}
if [ -n "$AB_TESTS" ]; then python -m pytest tests/test_gpu_parity.py tests/test_bench_parity.py -x -q -m gpu -k "$AB_TESTS" > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log; fi
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
run warm PAIS_X=1 -- --steps 5 --warmup 2
for i in 1 2 3; do
run pawn_new$i PAIS_X=1 -- --steps 20 --warmup 3
run pawn_prev$i PAIS_LIB_PATH=$V -- --steps 20 --warmup 3
done
} > $out/summary.txt 2>&1
cat $out/summary.txt; [ -f $out/tests.log ] && tail -n 4 $out/tests.log
