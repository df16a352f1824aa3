#!/bin/bash
# instruction-cache counters of the PSO kernels over one bench step (on the GPU box): bash scripts/pmc_icache.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/pmc_icache}; mkdir -p $out
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/p$i.log 2>&1
  python - << PY
import sqlite3,glob
f=glob.glob("$out/p$i/*.db")
if f:
    cur=sqlite3.connect(f[0]).cursor()
    print("## pass $i")
    try:
        for k,c,v,n in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
            if "k_pso" in k: print("%-40s %-30s %16.6g (%d)"%(k.split("(")[0][:40],c,v,n))
    except Exception as e: print("err",e)
PY
  rm -rf $out/p$i
done
