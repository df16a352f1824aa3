#!/bin/bash
# streamed rounds (PAIS_STREAM_ROUNDS / _ABOVE / _SPLIT): parity tests, then pawn / ring A/B.  Output under gpurun_out/$1
out=gpurun_out/${1:-stream}; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed or two_lanes" > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
run() { # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d.get('kernel_ms_per_step',{})
    print("%-28s value %10.1f ms/step %9.2f pso %8.2f enum %7.2f commit %7.2f streamed %d/%d sha %s" % (sys.argv[2], d['value'], d['ms_per_step'], k.get('pso_pass',0), k.get('host_enumerate',0), k.get('host_commit',0), d['config'].get('rounds_streamed_per_step',-1), d['config'].get('rounds_per_step',-1), str(d['config'].get('cloud_sha1'))[:10]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
{
run pawn_warm PAIS_STREAM_ROUNDS=0 -- --steps 5 --warmup 2
run pawn_off1 PAIS_STREAM_ROUNDS=0 -- --steps 20 --warmup 3
run pawn_auto1 PAIS_STREAM_ROUNDS=1 -- --steps 20 --warmup 3
run pawn_off2 PAIS_STREAM_ROUNDS=0 -- --steps 20 --warmup 3
run pawn_auto2 PAIS_STREAM_ROUNDS=1 -- --steps 20 --warmup 3
run ring_off1 PAIS_STREAM_ROUNDS=0 -- --scene ring --steps 1 --warmup 0 --max-rounds 300
run ring_auto1 PAIS_STREAM_ROUNDS=1 -- --scene ring --steps 1 --warmup 0 --max-rounds 300
run ring_all1 PAIS_STREAM_ROUNDS=2 -- --scene ring --steps 1 --warmup 0 --max-rounds 300
run ring_auto_a64 PAIS_STREAM_ABOVE=64 -- --scene ring --steps 1 --warmup 0 --max-rounds 300
run ring_auto_full PAIS_STREAM_ROUNDS=1 -- --scene ring --steps 1 --warmup 0
run ring_off_full PAIS_STREAM_ROUNDS=0 -- --scene ring --steps 1 --warmup 0
run dome_auto PAIS_STREAM_ROUNDS=1 -- --scene dome --steps 1 --warmup 0 --parents-per-round 4096 --max-rounds 12
run dome_off PAIS_STREAM_ROUNDS=0 -- --scene dome --steps 1 --warmup 0 --parents-per-round 4096 --max-rounds 12
} > $out/summary.txt 2>&1
cat $out/summary.txt; tail -n 3 $out/tests.log
