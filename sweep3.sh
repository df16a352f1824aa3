cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r03e; rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --scene dome --steps 1 --warmup 0 --max-rounds 3 --parents-per-round 1024 --no-cpu-baseline > $out/bench.log 2>&1
python scripts/rocprof_summary.py $(ls $out/kt/*.db | head -1) gpurun_out/r03e_dome_kernel_stats.txt /dev/null "dome tile" | head -30
rm -rf $out/kt
