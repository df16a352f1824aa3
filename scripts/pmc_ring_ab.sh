#!/bin/bash
# PMC passes over the pawn bench (k_pso_ring) with and without the swarm step's evaluation records (PAIS_PRE_SETUP): bash scripts/pmc_ring_ab.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_ring_ab; rm -rf $out; mkdir -p $out
for pre in 1 0; do
i=0
for set in \
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT" ; do
  i=$((i+1))
  PAIS_PRE_SETUP=$pre timeout 300 rocprofv3 --kernel-trace --pmc $set -d $out/p${pre}_$i -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-emulate > $out/p${pre}_$i.log 2>&1
  python - << PY
import sqlite3,glob
f=glob.glob("$out/p${pre}_$i/**/*.db", recursive=True)
if f:
    cur=sqlite3.connect(f[0]).cursor()
    print("## PAIS_PRE_SETUP=$pre pass $i")
    for k,c,v,n in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
        if any(s in k for s in ("k_pso_ring","k_pso_setup0")): print("%-24s %-36s %16.6g (%d)"%(k.split("(")[0][:24],c,v,n))
    for n,c,t in cur.execute("select name,count(*),sum(duration)/1e6 from kernels group by name"):
        if any(s in n for s in ("k_pso_ring","k_pso_setup0")): print("   dur_ms %-24s calls %d total %.3f"%(n.split("(")[0][:24],c,t))
PY
  rm -rf $out/p${pre}_$i
done
done > $out/summary.txt 2>&1
cat $out/summary.txt
