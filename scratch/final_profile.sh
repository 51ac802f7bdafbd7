#!/bin/bash
# headline bench line, per-op table and the rocprofv3 kernel statistics of the same command: usage scratch/final_profile.sh <tag>
set -e
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
cd $R
timeout -k 10 400 python bench.py --detail gpurun_out/$tag/per_op_events.txt > gpurun_out/$tag/bench.json
tail -c 600 gpurun_out/$tag/bench.json; echo
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag/prof -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/$tag/prof_bench.json 2> $R/gpurun_out/$tag/prof.err
cd $R
db=$(find gpurun_out/$tag/prof -name "*.db" | head -1)
if [ -n "$db" ]; then python scratch/rocpd_stats.py $db gpurun_out/$tag/rocprof_kernel_stats.csv; else cp $(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1) gpurun_out/$tag/rocprof_kernel_stats.csv; fi
head -8 gpurun_out/$tag/rocprof_kernel_stats.csv
