#!/bin/bash
# round 6, final measurements on one box (the committed library): profiles/r06_*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/valu_per_eval.sh pawn ring dome > gpurun_out/valu_per_eval.log 2>&1; cp gpurun_out/valu_model.json profiles/valu_model.json
bash scripts/make_profiles.sh r06 > gpurun_out/make_profiles_r06.log 2>&1
python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
python bench.py --scene dome --max-rounds 3 --parents-per-round 1024 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r06_bench_dome.json 2> gpurun_out/r06_bench_dome.err
python bench.py --scene dome --max-rounds 40 --steps 1 --warmup 1 --no-cpu-baseline --emulate-world 8 --emulate-steps 1 > gpurun_out/r06_bench_dome_r40.json 2> gpurun_out/r06_bench_dome_r40.err
PAIS_TILE_SPLIT=0 python bench.py --scene dome --max-rounds 40 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r06_bench_dome_r40_k_pso_tile.json 2> gpurun_out/r06_bench_dome_r40_k_pso_tile.err
python bench.py --scene dome --max-rounds 400 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r06_bench_dome_r400.json 2> gpurun_out/r06_bench_dome_r400.err
python bench.py --scene ring --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r06_bench_ring.json 2> gpurun_out/r06_bench_ring.err
bash scripts/pmc_tile.sh > gpurun_out/r06_pmc_dome.txt 2>&1
python - <<'PY'
import json
for f in ("r06_bench","r06_bench_dome","r06_bench_dome_r40","r06_bench_dome_r40_k_pso_tile","r06_bench_dome_r400","r06_bench_ring"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-32s value %9.1f ms %10.1f frac %.4f busy %.4f fp64 %s valu %s bound %s sha %s gold %s emu %s" % (f, d["value"], d["ms_per_step"], r["frac"], r.get("frac_over_busy_time") or 0, r.get("fp64_flops_frac"), r.get("valu_issue_frac"), r.get("bound"), str(d["config"].get("cloud_sha1"))[:10], d["config"].get("cloud_matches_oracle_golden"), (d["config"].get("emulated_speedup_at") or {}).get("8")))
    except Exception as e: print(f, "FAILED", e)
PY
