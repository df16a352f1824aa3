#!/bin/bash
# PMC passes over the saturated cost-evaluation kernel (k_fitness via scripts/microbench_eval.py).
# usage (on the GPU box): bash scripts/pmc_eval.sh <outdir> ; each pass is its own rocprofv3 run (kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/pmc_eval}
mkdir -p $out
i=0
for set in \
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_TRANS_F64" \
 "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
 "GRBM_GUI_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_ANY" \
 "FETCH_SIZE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o p -- python ${PMC_CMD:-scripts/microbench_eval.py 400000} > $out/p$i.log 2>&1
  python - << PY
import sqlite3,glob
f=glob.glob("$out/p$i/*.db")
if f:
    cur=sqlite3.connect(f[0]).cursor()
    print("## pass $i")
    try:
        for k,c,v,n in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
            if any(s in k for s in ("k_fitness","k_pso","k_after","k_begin")): print("%-24s %-36s %16.6g (%d)"%(k.split("(")[0][:24],c,v,n))
    except Exception as e: print("err",e)
    for n,c,t in cur.execute("select name,count(*),sum(duration)/1e6 from kernels group by name"):
        if any(s in n for s in ("k_fitness","k_pso","k_after","k_begin")): print("   dur_ms %-24s calls %d total %.3f"%(n.split("(")[0][:24],c,t))
PY
done
