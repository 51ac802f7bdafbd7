#!/bin/bash
# conv-stage time per image as a function of the batch (does a smaller working set help?)
set -e
for b in 16 32 64 128 256; do
  timeout -k 10 200 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --detail gpurun_out/sweep_$b.txt > gpurun_out/sweep_$b.json
  python - <<PY
import json
d = json.load(open("gpurun_out/sweep_$b.json"))
print("batch", $b, "ms_per_step", d["ms_per_step"], "us/img", 1000 * d["ms_per_step"] / $b, "stages", d.get("stages_ms"))
PY
done
