#!/bin/bash
# In-situ tuning (candidates timed inside whole-program runs) of one conv family of one model; the table goes to
# gpurun_out/<tag>.json (not to the committed table).  usage: scratch/tune_insitu.sh <model> <batch> <only-prefix> <tag>
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
cp yolo_amd/tuned_gfx950.json gpurun_out/$4.json
timeout -k 10 900 python -u -m yolo_amd.tuning --model $1 --batch $2 --in-situ --only "$3" --reps 5 --out gpurun_out/$4.json > gpurun_out/$4.log 2>&1
tail -12 gpurun_out/$4.log
