#!/bin/bash
# re-measure the tilings of one layer family: usage tune_only.sh k3s2 "v5s 256" "v8s 256" ...; result in gpurun_out/tuned_gfx950.json
set -e
R=$GRAFT_REPO_ROOT
cd $R
only=$1; shift
cp yolo_amd/tuned_gfx950.json gpurun_out/tuned_gfx950.json
for spec in "$@"; do
  set -- $spec
  echo "== $1 batch $2"
  YAMD_TUNING_TABLE=$R/gpurun_out/tuned_gfx950.json timeout -k 10 400 python -u -m yolo_amd.tuning --model $1 --batch $2 --reps 8 --only $only --out $R/gpurun_out/tuned_gfx950.json > gpurun_out/tune_only_$1.log 2>&1
  tail -1 gpurun_out/tune_only_$1.log
done
