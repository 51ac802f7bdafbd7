#!/bin/bash
# re-measure the per-layer tilings of the benchmark models; the merged table lands in gpurun_out/tuned_gfx950.json
set -e
R=$GRAFT_REPO_ROOT
cd $R
cp yolo_amd/tuned_gfx950.json gpurun_out/tuned_gfx950.json
for spec in "v5s 256" "v8s 256" "v3 256" "v4 256" "v7pose 256" "v6face 256" "v7 128" "v6 128" "v5m 64" "v8m 64"; do
  set -- $spec
  echo "== $1 batch $2"
  YAMD_TUNING_TABLE=$R/gpurun_out/tuned_gfx950.json timeout -k 10 400 python -u -m yolo_amd.tuning --model $1 --batch $2 --reps 8 --out $R/gpurun_out/tuned_gfx950.json > gpurun_out/retune_$1.log 2>&1
  tail -1 gpurun_out/retune_$1.log
done
