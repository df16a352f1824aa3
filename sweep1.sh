mkdir -p gpurun_out/r03b
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03b/$name.json 2> gpurun_out/r03b/$name.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03b/$name.json").read().strip().splitlines()[-1])
print("$name", round(d["ms_per_step"],2), d["config"]["cloud_matches_oracle_golden"], {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()})
PY
}
run base X=1
run base2 X=1
run fill050 PAIS_PART_FILL=0.5
run fill075 PAIS_PART_FILL=0.75
run fill0375 PAIS_PART_FILL=0.375
run fill025 PAIS_PART_FILL=0.25
run split6k PAIS_SPLIT_ABOVE=6144
run split4k PAIS_SPLIT_ABOVE=4096
run split3k PAIS_SPLIT_ABOVE=3072
run split12k PAIS_SPLIT_ABOVE=12288
run parts1 PAIS_EVAL_PARTS=1
run parts2 PAIS_EVAL_PARTS=2
