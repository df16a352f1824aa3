mkdir -p gpurun_out/r03e
for mode in 0 1; do
  PAIS_TILE=$mode python bench.py --scene dome --steps 1 --warmup 1 --max-rounds 3 --parents-per-round 1024 --no-cpu-baseline > gpurun_out/r03e/dome_tile$mode.json 2> gpurun_out/r03e/dome_tile$mode.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03e/dome_tile$mode.json").read().strip().splitlines()[-1])
print("tile=$mode", round(d["value"],1), "patches/s", round(d["ms_per_step"],1), "ms", d["config"]["patches_per_step"], d["config"]["accepted_patches"], d["config"]["cloud_sha1"], {k:round(v,1) for k,v in d["kernel_ms_per_step"].items()})
PY
done
