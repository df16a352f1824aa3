#!/bin/bash
# usage: scripts/variant_sweep.sh g4 g2 ...   (on the GPU box)
for v in "$@"; do
  export PAIS_LIB_PATH=$PWD/pais_mvs_amd/csrc/variants/libpais_hip_$v.so
  m=$(timeout 200 python scripts/microbench_eval.py 600000 2>&1 | tail -1)
  b=$(timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.0f patches/s step %.1f ms eval %.1f ms pass %.1f ms'%(j['value'], j['ms_per_step'], j['kernel_ms']['k_pso_eval']/2, j['kernel_ms']['pso_pass']/2))")
  echo "$v | $m | $b"
done
