# PMC passes over the tile kernel (dome, seeds + 2 rounds): bash scripts/pmc_tile.sh on the GPU box -> profiles/r03_pmc_tile.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_tile; rm -rf $out; mkdir -p $out
i=0
for set in \
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
 "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o p -- python bench.py --scene dome --steps 1 --warmup 0 --max-rounds 2 --parents-per-round 1024 --no-cpu-baseline > $out/p$i.log 2>&1
  python - << PY
import sqlite3,glob
f=glob.glob("$out/p$i/*.db")
if f:
    cur=sqlite3.connect(f[0]).cursor()
    print("## pass $i")
    try:
        for k,c,v,n in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
            if any(s in k for s in ("k_pso_tile","k_pso_eval2")): print("%-24s %-30s %16.6g (%d)"%(k.split("(")[0][:24],c,v,n))
    except Exception as e: print("err",e)
    for n,c,t in cur.execute("select name,count(*),sum(duration)/1e6 from kernels group by name"):
        if any(s in n for s in ("k_pso_tile","k_pso_eval2")): print("   dur_ms %-24s calls %d total %.3f"%(n.split("(")[0][:24],c,t))
else:
    print("no db", open("$out/p$i.log").read()[-500:])
PY
  rm -rf $out/p$i
done
