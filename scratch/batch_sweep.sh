#!/bin/bash
# conv stage time per image against the batch size: usage batch_sweep.sh model out.txt b1 b2 ...
m=$1; out=$2; shift 2; : > $out
for b in "$@"; do
  timeout -k 10 240 python bench.py --model $m --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /tmp/bs_$b.json 2> /tmp/bs_$b.err || { echo "$b FAILED" >> $out; tail -3 /tmp/bs_$b.err >> $out; exit 1; }
  python - $b <<'PY' >> $out
import json, sys
d = json.loads(open(f"/tmp/bs_{sys.argv[1]}.json").read().strip().splitlines()[-1])
b = d["config"]["images_per_rank"][0]
print("B=%-4d %9.1f images/s  %8.3f ms/step  conv %7.3f ms = %6.2f us/image (%6.1f TFLOP/s)  stages %s" % (
    b, d["value"], d["ms_per_step"], d["stages_ms"]["conv"], 1e3 * d["stages_ms"]["conv"] / b, d["roofline"]["achieved"], d["stages_ms"]))
PY
  tail -1 $out
done
