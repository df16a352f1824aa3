cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r03f; rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace -d $out/kt -o kt -- python bench.py --scene dome --steps 1 --warmup 0 --max-rounds 3 --parents-per-round 1024 --no-cpu-baseline > $out/bench.log 2>&1
python - <<PY
import sqlite3,glob
db=sqlite3.connect(glob.glob("$out/kt/*.db")[0]); cur=db.cursor()
rows=[(r[0].split('(')[0].replace('void ',''),r[1],r[2],r[3],r[4]) for r in cur.execute("select name,start,end,grid_x,queue_id from kernels order by start")]
# find the middle of the tile launches
idx=[i for i,r in enumerate(rows) if r[0].startswith('k_pso_tile')]
m=idx[len(idx)//2]
t0=rows[m][1]
for r in rows[m-2:m+26]:
    print("%-28s q%-3d start %9.1f us  dur %8.1f us  grid %d"%(r[0][:28],r[4],(r[1]-t0)/1e3,(r[2]-r[1])/1e3,r[3]))
PY
rm -rf $out/kt
