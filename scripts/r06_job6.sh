bash scripts/valu_per_eval.sh pawn ring dome > gpurun_out/valu_per_eval.log 2>&1; cp gpurun_out/valu_model.json profiles/valu_model.json
python scripts/round_log.py default > gpurun_out/round_log_default.json 2> gpurun_out/round_log_default.err
PAIS_SPLIT_ABOVE=1 PAIS_RING_PER_CAM=0 python scripts/round_log.py ring_for_every_round > gpurun_out/round_log_ringall.json 2> gpurun_out/round_log_ringall.err
python bench.py > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err
tail -c 1500 gpurun_out/valu_per_eval.log; python - <<'PY'
import json
for f in ("gpurun_out/round_log_default.json","gpurun_out/round_log_ringall.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print({k:d[k] for k in d if k!="rounds"})
    except Exception as e: print(f, e)
d=json.loads(open("gpurun_out/r06_bench_a.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["bound"], d["roofline"]["frac"], d["roofline"]["fp64_flops_frac"], d["roofline"]["valu_issue_frac"], d["config"]["literal_arithmetic"], d["config"]["cloud_matches_oracle_golden"])
PY
