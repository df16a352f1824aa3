#!/bin/bash
# Throughput attribution of the tile kernel by REMOVAL (wrong results; only the first seed pass is looked at, whose work does not depend on the
# cost values): rocprofv3 --kernel-trace of `bench.py --scene dome --max-rounds 1`, average duration of the first 60 k_pso_tile launches per build
#   -DTILE_EXP_SKIP=1 four of five homography reads | 2 the byte taps | 3 the exp of the per-pixel tail
tag=${1:-dome_skip}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
for lib in ${LIBS:-"" variants/libpais_tskip1.so variants/libpais_tskip2.so variants/libpais_tskip3.so}; do
  name=$(basename "${lib:-base}" .so)
  if [ -n "$lib" ]; then export PAIS_LIB_PATH=pais_mvs_amd/csrc/$lib; else unset PAIS_LIB_PATH; fi
  timeout 240 rocprofv3 --kernel-trace -d $out/kt_$name -o kt -- python bench.py --scene dome --max-rounds 1 --steps 1 --warmup 0 --no-cpu-baseline > $out/$name.log 2>&1
  python - $out/kt_$name $name <<'PY'
import sqlite3, glob, sys
f = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)
if not f:
    print(sys.argv[2], "no db"); sys.exit()
cur = sqlite3.connect(f[0]).cursor()
rows = cur.execute("select name, duration from kernels where name like '%k_pso_tile%' order by start").fetchall()
first = [d for n, d in rows[:60]]
print("%-16s first %d tile launches (%s): avg %.1f us" % (sys.argv[2], len(first), rows[0][0].split("(")[0] if rows else "-", sum(first) / max(len(first), 1) / 1e3))
PY
  rm -rf $out/kt_$name
done
