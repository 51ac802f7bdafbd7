#!/bin/bash
# re-measure the per-layer tilings of the benchmark models; the merged table lands in gpurun_out/tuned_gfx950.json
set -e
R=$GRAFT_REPO_ROOT
cd $R
# usage: retune.sh "v5s 256" "v3 256" ...   (starts from gpurun_out/tuned_gfx950.json when it exists, so calls chain)
[ -f gpurun_out/tuned_gfx950.json ] || cp yolo_amd/tuned_gfx950.json gpurun_out/tuned_gfx950.json
for spec in "$@"; do
  set -- $spec
  echo "== $1 batch $2"
  YAMD_TUNING_TABLE=$R/gpurun_out/tuned_gfx950.json timeout -k 10 400 python -u -m yolo_amd.tuning --model $1 --batch $2 --reps 8 --out $R/gpurun_out/tuned_gfx950.json > gpurun_out/retune_$1.log 2>&1
  tail -1 gpurun_out/retune_$1.log
done
# in-situ pass for the headline models: RETUNE_IN_SITU="v5s 256;v8s 256"
IFS=';' read -ra SPECS <<< "${RETUNE_IN_SITU:-}"
for spec in "${SPECS[@]}"; do
  set -- $spec
  echo "== $1 batch $2 (in situ)"
  YAMD_TUNING_TABLE=$R/gpurun_out/tuned_gfx950.json timeout -k 10 500 python -u -m yolo_amd.tuning --model $1 --batch $2 --reps 6 --in-situ --out $R/gpurun_out/tuned_gfx950.json > gpurun_out/retune_insitu_$1.log 2>&1
  tail -1 gpurun_out/retune_insitu_$1.log
done
