#!/bin/bash
# kernel-trace summary of a bench command (on the GPU box): bash scripts/kt.sh <tag> [bench args]
tag=${1:-kt}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/bench.log 2>&1
grep '^{"metric"' $out/bench.log | tail -1 > gpurun_out/${tag}_bench_under_rocprof.json
python scripts/rocprof_summary.py $(ls $out/kt/*.db | head -1) gpurun_out/${tag}_kernel_stats.txt gpurun_out/${tag}_bench_under_rocprof.json "python bench.py --steps 2 --warmup 1 $* ($tag)"
python scripts/rounds_table.py $(ls $out/kt/*.db | head -1) > gpurun_out/${tag}_rounds.txt 2>&1
rm -rf $out/kt
