#!/bin/bash
# Slowdown profile of the tile kernel's walk (dome seeds + R rounds): the library against builds that execute ONE component twice
# (-DTILE_EXP_DUP=1 homography reads | 2 byte taps | 3 a whole camera group | 4 the per-pixel tail | 5 WinPix loads; same records)
out=gpurun_out/${1:-dome_dup}; mkdir -p $out; R=${2:-10}; B=${3:-4096}
run() { name=$1; lib=$2
  if [ -n "$lib" ]; then export PAIS_LIB_PATH=$lib; else unset PAIS_LIB_PATH; fi
  python bench.py --no-cpu-baseline --scene dome --max-rounds $R --parents-per-round $B --steps 1 --warmup 1 > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{})
    print("%-10s ms/step %9.1f pso %9.1f sha %s" % (sys.argv[2], d['ms_per_step'], k.get('pso_pass',0), str(d['config'].get('cloud_sha1'))[:10]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
{
run base ""
for d in 1 2 3 4 5; do run tdup$d pais_mvs_amd/csrc/variants/libpais_tdup$d.so; done
run base_b ""
} > $out/summary.txt 2>&1
cat $out/summary.txt
