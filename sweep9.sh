mkdir -p gpurun_out/r03j
python bench.py --steps 10 --warmup 3 > gpurun_out/r03j/bench.json 2> gpurun_out/r03j/bench.err; echo bench rc=$?
MB_JSON=gpurun_out/r03j/microbench_eval.json python scripts/microbench_eval.py 1200000 > gpurun_out/r03j/microbench.log 2>&1
bash scripts/make_profiles.sh r03 > gpurun_out/r03j/make_profiles.log 2>&1
bash scripts/kt.sh r03k > /dev/null 2>&1
SCENE=dome SEEDS=400 PPR=1024 MAXR=3 bash scripts/make_profiles.sh r03_dome --scene dome --max-rounds 3 --parents-per-round 1024 > gpurun_out/r03j/make_profiles_dome.log 2>&1
python bench.py --scene dome --steps 1 --warmup 1 --max-rounds 3 --parents-per-round 1024 --no-cpu-baseline > gpurun_out/r03j/dome.json 2> gpurun_out/r03j/dome.err
PAIS_TILE=0 python bench.py --scene dome --steps 1 --warmup 1 --max-rounds 3 --parents-per-round 1024 --no-cpu-baseline > gpurun_out/r03j/dome_notile.json 2> gpurun_out/r03j/dome_notile.err
python bench.py --scene ring --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03j/ring.json 2> gpurun_out/r03j/ring.err
python - <<PY
import json
for f in ("bench","dome","dome_notile","ring"):
    try:
        d=json.loads(open("gpurun_out/r03j/%s.json"%f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), round(d["ms_per_step"],2), d["config"].get("cloud_sha1","")[:10], d["config"].get("cloud_matches_oracle_golden"), round(d["roofline"]["frac"],4), {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()}, d["config"].get("predicted_speedup_at"))
    except Exception as e: print(f,"ERR",e)
PY
