#!/bin/bash
# engine clock and VALU issue utilisation of the saturated cost kernel: GRBM_GUI_ACTIVE / duration, SQ_INSTS_VALU x 4 / (SIMDs x cycles)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/pmc_clock}; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 -d $out/p -o p -- python scripts/microbench_eval.py 1200000 > $out/p.log 2>&1
python - << PY
import sqlite3,glob
f=glob.glob("$out/p/*.db")
cur=sqlite3.connect(f[0]).cursor()
c={}
for k,cn,v,n in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
    if "k_fitness" in k: c[cn]=(v,n)
d=[(n,cnt,t) for n,cnt,t in cur.execute("select name,count(*),sum(duration)/1e9 from kernels group by name") if "k_fitness" in n][0]
gui=c["GRBM_GUI_ACTIVE"][0]; valu=c["SQ_INSTS_VALU"][0]; dur=d[2]
print("k_fitness: %d launches, %.3f ms; GRBM_GUI_ACTIVE %.4g (sum over 8 XCDs) -> engine clock %.2f GHz; SQ_INSTS_VALU %.4g -> VALU issue utilisation %.1f %% (4 cycles per wave64 instruction, 1024 SIMDs); FP64 share of VALU instructions %.1f %%"
      % (d[1], dur*1e3, gui, gui/8/dur/1e9, valu, 100*valu*4/(1024*gui/8), 100*(c["SQ_INSTS_VALU_FMA_F64"][0]+c["SQ_INSTS_VALU_MUL_F64"][0]+c["SQ_INSTS_VALU_ADD_F64"][0])/valu))
PY
tail -n 1 $out/p.log
