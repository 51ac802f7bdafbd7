#!/bin/bash
# bench.py for the models named on the command line, one line each: usage scratch/bench_some.sh out.txt v4 v7 ...
out=$1; shift; : > $out
for m in "$@"; do
  timeout -k 10 240 python bench.py --model $m --steps 8 --warmup 3 --no-cpu-baseline --no-extras > /tmp/b_$m.json 2> /tmp/b_$m.err || { echo "$m FAILED" >> $out; tail -3 /tmp/b_$m.err >> $out; exit 1; }
  python - $m <<'PY' >> $out
import json, sys
d = json.loads(open(f"/tmp/b_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("%-8s B=%-4d %9.1f images/s  %8.3f ms/step  conv %7.3f ms (%6.1f TFLOP/s)  stages %s" % (
    sys.argv[1], d["config"]["images_per_rank"][0], d["value"], d["ms_per_step"], d["stages_ms"]["conv"], d["roofline"]["achieved"], d["stages_ms"]))
PY
  tail -1 $out
done
