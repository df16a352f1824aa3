#!/bin/bash
# Slowdown profiling of the latency-bound PSO chain: the library against builds that execute ONE component of
# k_pso_iter twice (-DPAIS_EXP_DUP=1 move selections | 2 gBest scan + dispersion test | 3 cost evaluation | 4 uniforms | 5 ranks + lBest | 6 FDR x3 | 7 normal + homographies; same records).
# The slowdown of the pawn reconstruction is that component's share of the critical path.  On the GPU box:
#   bash scripts/dup_profile.sh <tag>       (variants built beforehand into pais_mvs_amd/csrc/variants/libpais_dup{1,2,3}.so)
out=gpurun_out/${1:-dup}; mkdir -p $out
run() { name=$1; lib=$2
  if [ -n "$lib" ]; then export PAIS_LIB_PATH=$lib; else unset PAIS_LIB_PATH; fi
  python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{})
    print("%-14s ms/step %8.2f  pso %7.2f  golden %s" % (sys.argv[2], d['ms_per_step'], k.get('pso_pass',0), d['config'].get('cloud_matches_oracle_golden')))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
{
run warm ""
for i in 1 2; do
run base_$i ""
for d in 1 2 3 4 5 6 7; do run dup${d}_$i pais_mvs_amd/csrc/variants/libpais_dup$d.so; done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
