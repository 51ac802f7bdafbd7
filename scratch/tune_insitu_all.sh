#!/bin/bash
# In-situ tuning of every conv layer of the six benchmark models, one after the other, each starting from the table the
# previous one left: gpurun_out/<tag>_table.json (the committed table is not touched).  usage: scratch/tune_insitu_all.sh <tag>
set -e
R=$GRAFT_REPO_ROOT
tag=$1
cd $R
mkdir -p gpurun_out
cp yolo_amd/tuned_gfx950.json gpurun_out/${tag}_table.json
export YAMD_TUNING_TABLE=$R/gpurun_out/${tag}_table.json
for mb in "v3 256" "v4 256" "v7pose 256" "v6face 256" "v8s 256" "v5s 256"; do
  set -- $mb
  echo "== $1 (batch $2)" >> gpurun_out/${tag}.log
  timeout -k 10 900 python -u -m yolo_amd.tuning --model $1 --batch $2 --in-situ --reps 5 --out gpurun_out/${tag}_table.json >> gpurun_out/${tag}.log 2>&1
  tail -2 gpurun_out/${tag}.log
done
