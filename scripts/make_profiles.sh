#!/bin/bash
# Runs ON THE GPU BOX (gpurun): produces the rocprofv3 evidence that is then copied into profiles/.
#   bash scripts/make_profiles.sh <tag>       e.g. r01d
# 1. kernel-trace + stats of the default bench command   -> gpurun_out/<tag>_kernel_stats.txt, <tag>_bench_under_rocprof.json
# 2. PMC passes (kernel-trace only, one rocprofv3 run per counter set) of `bench.py --steps 1 --warmup 0`
#                                                          -> gpurun_out/<tag>_pmc.txt, <tag>_pmc_traffic.json
tag=${1:-r01x}; shift
# further arguments go to bench.py (e.g. --scene dome --max-rounds 3 --parents-per-round 1024); SCENE / SEEDS / PPR / MAXR
# describe that workload in the traffic record (defaults = bench.py's defaults)
SCENE=${SCENE:-pawn}; SEEDS=${SEEDS:-200}; PPR=${PPR:-4096}; MAXR=${MAXR:-0}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --steps 2 --warmup 1 --cpu-seconds 4 --no-emulate "$@" > $out/bench.log 2>&1
grep '^{"metric"' $out/bench.log | tail -1 > gpurun_out/${tag}_bench_under_rocprof.json
python scripts/rocprof_summary.py $(ls $out/kt/*.db | head -1) gpurun_out/${tag}_kernel_stats.txt gpurun_out/${tag}_bench_under_rocprof.json "python bench.py --steps 2 --warmup 1 ($tag)" > /dev/null
# per-round table of one reconstruction (warm-up, timed, roofline step: scripts/rounds_table.py reads the middle one)
timeout 600 rocprofv3 --kernel-trace -d $out/kr -o kr -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-emulate "$@" > $out/rounds.log 2>&1
python scripts/rounds_table.py $(ls $out/kr/*.db | head -1) > gpurun_out/${tag}_rounds.txt 2>&1
rm -rf $out/kr
i=0
: > gpurun_out/${tag}_pmc.txt
echo "# rocprofv3 --pmc passes (separate runs, --kernel-trace only) of: python bench.py --steps 1 --warmup 0 --no-cpu-baseline" >> gpurun_out/${tag}_pmc.txt
echo "# (1 timed step + the instrumented roofline step = 2 reconstructions).  Sums over all dispatches of a kernel." >> gpurun_out/${tag}_pmc.txt
for set in \
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
 "FETCH_SIZE" \
 "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-emulate "$@" > $out/p$i.log 2>&1
  python - >> gpurun_out/${tag}_pmc.txt << PY
import sqlite3,glob,json
f=glob.glob("$out/p$i/*.db")
print("\n## pass $i: $set")
if f:
    cur=sqlite3.connect(f[0]).cursor()
    sums={}
    for k,c,v,n in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
        kn=k.split("(")[0].replace("void ","")
        print("%-18s %-28s %16.6g (%d dispatches)"%(kn[:18],c,v,n))
        sums[(kn,c)]=(v,n)
    for n_,c_,t_ in cur.execute("select name,count(*),sum(duration)/1e6 from kernels group by name"):
        print("   dur_ms %-22s calls %6d total %.3f"%(n_.split("(")[0].replace("void ","")[:22],c_,t_))
    if "$set"=="FETCH_SIZE":
        def per_launch(pred):
            tot=sum(v for (k,c),(v,n) in sums.items() if pred(k)); nl=sum(n for (k,c),(v,n) in sums.items() if pred(k))
            return (tot*1024.0/nl if nl else None), nl
        raw_all,nl_all=per_launch(lambda k: k.startswith("k_pso_iter") or k.startswith("k_pso_eval") or k.startswith("k_pso_ring"))
        raw_e2,nl_e2=per_launch(lambda k: k.startswith("k_pso_eval") or k.startswith("k_pso_tile") or k.startswith("k_pso_ring"))
        raw_ring,nl_ring=per_launch(lambda k: k.startswith("k_pso_ring"))
        raw_eo,nl_eo=per_launch(lambda k: k.startswith("k_pso_eval"))
        if nl_all:
            json.dump({"$SCENE": {"seeds": $SEEDS, "parents_per_round": $PPR, "max_rounds": $MAXR,
                       "eval2_hbm_read_bytes_per_launch": (2.0*raw_e2 if raw_e2 else None), "eval2_launches": nl_e2,
                       "ring_hbm_read_bytes_per_launch": (2.0*raw_ring if raw_ring else None), "ring_launches": nl_ring,
                       "eval2_only_hbm_read_bytes_per_launch": (2.0*raw_eo if raw_eo else None), "eval2_only_launches": nl_eo,
                       "eval_hbm_read_bytes_per_launch": 2.0*raw_all, "launches": nl_all, "tag": "$tag",
                       "note": "rocprofv3 --pmc FETCH_SIZE (KiB) summed over the dispatches of the kernel in 'bench.py --steps 1 --warmup 0 $*' / its launches, x2 (MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B; calibrated for wide coalesced reads only, so an upper bound for these 2- / 8-byte gathers)"}},
                      open("gpurun_out/${tag}_pmc_traffic.json","w"), indent=1)
else:
    print("no db (see log)")
PY
done
rm -rf $out/kt $out/p1 $out/p2 $out/p3 $out/p4   # raw dbs stay on the box (gpurun_out is capped at 64 MiB)
cat gpurun_out/${tag}_kernel_stats.txt | head -30
cat gpurun_out/${tag}_pmc.txt | grep -E "k_pso_iter|##" | head -60
cat gpurun_out/${tag}_pmc_traffic.json
