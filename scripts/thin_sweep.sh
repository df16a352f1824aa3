#!/bin/bash
# PAIS_THIN_FRONT (part of the schedule R(B): changes the cloud): pawn reconstruction time and work per setting
out=gpurun_out/${1:-thin}; mkdir -p $out
for t in 64 64 128 256 512 1024 2048 100000; do
  PAIS_THIN_FRONT=$t python bench.py --no-cpu-baseline --steps 12 --warmup 2 > $out/t$t.json 2> $out/t$t.err
  python - $out/t$t.json $t <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print("thin_front %7s  ms/step %7.2f  value %9.1f  refines %6d accepted %6d speculative %5d rounds %3d" % (sys.argv[2], d['ms_per_step'], d['value'], c['patches_per_step'], c['accepted_patches'], c['speculative_extra_refines_per_step'], c['rounds_per_step']))
PY
done > $out/summary.txt 2>&1
cat $out/summary.txt
