run() { name=$1; shift
  env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],2), d['config']['cloud_matches_oracle_golden'], {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
}
run base X=1
run minper128 PAIS_PSO_MINPER=128
run minper256 PAIS_PSO_MINPER=256
run minper512 PAIS_PSO_MINPER=512
run fill06 PAIS_PART_FILL=0.6
run fill09 PAIS_PART_FILL=0.9
run split7k PAIS_SPLIT_ABOVE=7168
run split11k PAIS_SPLIT_ABOVE=11264
run base2 X=1
