#!/bin/bash
# everything profiles/ holds for one state of the code: usage scratch/final_profiles.sh <tag>
set -e
tag=$1
bash scratch/profile_bench.sh $tag
bash scratch/layer_counters.sh v5s 256 ${tag}_v5s_layers
bash scratch/layer_counters.sh v3 256 ${tag}_v3_layers
bash scratch/layer_counters.sh v8s 256 ${tag}_v8s_layers
bash scratch/bench_all.sh gpurun_out/$tag/all_models_bench.txt
