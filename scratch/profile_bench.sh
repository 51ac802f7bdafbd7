#!/bin/bash
# The headline bench line, its per-op event table, and the rocprofv3 kernel statistics of a run of the same command in
# which every launch belongs to a full-batch step (no CPU baseline, no extras).  usage: scratch/profile_bench.sh <tag> [bench args]
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout -k 10 500 python bench.py "$@" --detail $O/per_op_events.txt > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_bench.json 2> $O/prof.err
cd $R
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprof_kernel_stats.csv
rm -rf $O/prof
head -6 $O/rocprof_kernel_stats.csv | cut -c1-160
