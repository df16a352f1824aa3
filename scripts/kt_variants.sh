#!/bin/bash
# per-kernel average launch durations (rocprofv3 --kernel-trace) of the pawn bench for a list of library variants:
#   bash scripts/kt_variants.sh <tag> "" variants/libpais_dup2.so variants/libpais_dup12.so ...      ("" = the built library)
tag=${1:-ktv}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
for lib in "$@"; do
  name=$(basename "${lib:-base}" .so)
  if [ -n "$lib" ]; then export PAIS_LIB_PATH=pais_mvs_amd/csrc/$lib; else unset PAIS_LIB_PATH; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt_$name -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-emulate > $out/$name.log 2>&1
  python - $out/kt_$name $name <<'PY'
import sqlite3, glob, sys
f = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)
if not f:
    print(sys.argv[2], "no db"); sys.exit()
cur = sqlite3.connect(f[0]).cursor()
rows = cur.execute("select name, count(*), sum(duration) / 1e3, avg(duration) / 1e3 from kernels group by name").fetchall()
print("## %s" % sys.argv[2])
for n, c, t, a in sorted(rows, key=lambda r: -r[2]):
    if "k_pso" in n or "k_after" in n or "k_region" in n or "k_begin" in n:
        print("  %-40s calls %6d total_us %10.1f avg_us %8.2f" % (n.split("(")[0][:40], c, t, a))
PY
  grep '^{"metric"' $out/$name.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench ms/step %.2f golden %s' % (d['ms_per_step'], d['config']['cloud_matches_oracle_golden']))"
  rm -rf $out/kt_$name
done
