#!/bin/bash
# VALU instructions per cost evaluation of the dominant kernel of each bench scene (PMC SQ_INSTS_VALU over bench.py itself)
# -> gpurun_out/valu_model.json (copied to profiles/valu_model.json, which bench.py reads for roofline.valu_issue_frac).
#   bash scripts/valu_per_eval.sh [scenes...]      default: pawn ring dome
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/valu_model; rm -rf $out; mkdir -p $out
scenes=${@:-pawn ring dome}
for sc in $scenes; do
  extra=""
  [ $sc = ring ] && extra="--max-rounds 3"
  [ $sc = dome ] && extra="--max-rounds 10"
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -d $out/$sc -o p -- python bench.py --scene $sc --steps 1 --warmup 0 --no-emulate --no-cpu-baseline $extra > $out/$sc.json 2> $out/$sc.err
done
python - $out $scenes <<'PY'
import sqlite3, glob, json, sys, os
out, scenes = sys.argv[1], sys.argv[2:]
model = {}
for sc in scenes:
    try:
        d = json.loads(open("%s/%s.json" % (out, sc)).read().strip().splitlines()[-1])
        f = glob.glob("%s/%s/**/*.db" % (out, sc), recursive=True)
        cur = sqlite3.connect(f[0]).cursor()
        rows = list(cur.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name='SQ_INSTS_VALU' group by kernel_name"))
        r = d["roofline"]
        kind = "tile" if "k_pso_tile" in r["kernel"] else ("ring" if r["kernel"].startswith("k_pso_ring") else "eval2")
        pat = {"tile": "k_pso_tile", "ring": "k_pso_ring", "eval2": "k_pso_eval2"}[kind]
        insts = sum(v for k, v, n in rows if pat in k)
        calls = sum(n for k, v, n in rows if pat in k)
        steps = 2            # --steps 1 --warmup 0 + the instrumented roofline step
        evals = r["evals"] * steps
        model.setdefault(sc, {})[kind] = {"kernel": pat, "valu_insts_per_eval": insts / max(evals, 1), "valu_insts": insts, "evals": evals,
                                          "dispatches": calls, "cameras_per_eval": r.get("what_binds", {}).get("cameras_per_eval"),
                                          "source": "scripts/valu_per_eval.sh: rocprofv3 --pmc SQ_INSTS_VALU over bench.py --scene %s (2 steps)" % sc}
    except Exception as e:
        model.setdefault(sc, {})["error"] = str(e)
json.dump(model, open("gpurun_out/valu_model.json", "w"), indent=1)
print(json.dumps(model, indent=1))
PY
for sc in $scenes; do rm -rf $out/$sc; done
